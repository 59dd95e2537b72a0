"""FlowUniPCMultistepScheduler: the multistep predictor-corrector of the 50-step sampler.

Mirrors the interface the reference's `CausalDiffusionInferencePipeline` drives
(wan/utils/fm_solvers_unipc.py: `__init__` :77-133, `set_timesteps` :160-227, `step` :655-739; used at
pipeline/causal_diffusion_inference.py:376-434, :517-525).  Design: every tensor expression of UniPC --
`convert_model_output` (:279-347), the UniP update (:350-484) and the UniC update (:486-626) -- is a LINEAR
combination of the current sample and the stored model outputs, with scalars that depend only on the sigma
table.  The host evaluates those scalars (float32 torch scalars in the reference's order of operations, then
collapsed per tensor in float64) and one `sf_lincomb_bf16` launch applies them; nothing but the latent-sized
bf16 tensors lives on the device and nothing syncs.

`step_plan()` returns the coefficients without touching a device, which is what the CPU tests check against the
oracle; `step()` is the device path (HIP only: `ops.lincomb` raises for CPU tensors).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import ops


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor


@dataclass(frozen=True)
class StepPlan:
    """Scalars of one `step()` call.  Tensors are named: 'sample' (the step's input), 'model' (the model output),
    'last' (the sample before the previous predictor), 'm0', 'm1', ... (converted outputs, newest first, BEFORE this
    step's output is pushed), 'mt' (this step's converted output), 'cur' (the sample after the corrector)."""
    step_index: int
    order: int                                   # order of the predictor that follows
    convert: Tuple[Tuple[str, float], ...]       # mt   = sum c * tensor   over ('sample', 'model')
    correct: Optional[Tuple[Tuple[str, float], ...]]   # cur = ... over ('last', 'm0', 'm1'.., 'mt'); None: cur = sample
    predict: Tuple[Tuple[str, float], ...]       # prev_sample = ... over ('cur', 'mt', 'm0', ...)  (old m0 = new m1)


class _Config:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class FlowUniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 shift: Optional[float] = 1.0, use_dynamic_shifting: bool = False, thresholding: bool = False,
                 dynamic_thresholding_ratio: float = 0.995, sample_max_value: float = 1.0, predict_x0: bool = True,
                 solver_type: str = "bh2", lower_order_final: bool = True, disable_corrector: Sequence[int] = (),
                 solver_p=None, timestep_spacing: str = "linspace", steps_offset: int = 0,
                 final_sigmas_type: Optional[str] = "zero"):
        if solver_type not in ("bh1", "bh2"):
            if solver_type in ("midpoint", "heun", "logrho"):
                solver_type = "bh2"             # fm_solvers_unipc.py:97-99
            else:
                raise NotImplementedError(f"{solver_type} is not implemented for {self.__class__}")
        if prediction_type != "flow_prediction":
            raise ValueError(f"prediction_type given as {prediction_type} must be `flow_prediction` for this scheduler")
        if use_dynamic_shifting or thresholding or solver_p is not None:
            raise NotImplementedError("use_dynamic_shifting / thresholding / solver_p are not linear-update paths and are "
                                      "not used by the causal sampler (causal_diffusion_inference.py:517-521)")
        if final_sigmas_type != "zero":
            raise NotImplementedError("final_sigmas_type must be 'zero' (the reference's 'sigma_min' branch reads an "
                                      "attribute that does not exist, fm_solvers_unipc.py:195-197)")
        self.config = _Config(num_train_timesteps=num_train_timesteps, solver_order=solver_order,
                              prediction_type=prediction_type, shift=shift, use_dynamic_shifting=use_dynamic_shifting,
                              thresholding=thresholding, dynamic_thresholding_ratio=dynamic_thresholding_ratio,
                              sample_max_value=sample_max_value, predict_x0=predict_x0, solver_type=solver_type,
                              lower_order_final=lower_order_final, disable_corrector=list(disable_corrector),
                              solver_p=solver_p, timestep_spacing=timestep_spacing, steps_offset=steps_offset,
                              final_sigmas_type=final_sigmas_type)
        self.predict_x0 = predict_x0
        self.num_inference_steps = None
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.sigmas = sigmas
        self.timesteps = sigmas * num_train_timesteps
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * solver_order
        self.timestep_list: List = [None] * solver_order
        self.lower_order_nums = 0
        self.disable_corrector = list(disable_corrector)
        self.solver_p = None
        self.last_sample = None
        self._step_index = None
        self._begin_index = None
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()

    # ------------------------------------------------------------------------------------------
    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def __len__(self):
        return self.config.num_train_timesteps

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, sigmas=None, mu=None,
                      shift: Optional[float] = None):
        """fm_solvers_unipc.py:160-227: linspace(sigma_max, sigma_min, n+1)[:-1], shifted; timesteps are the
        sigmas * num_train_timesteps TRUNCATED to int64 (that is what the model is conditioned on), the sigmas
        stay float32 with a final 0 appended."""
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        else:
            sigmas = np.asarray(sigmas, dtype=np.float64)
        if shift is None:
            shift = self.config.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.config.num_train_timesteps
        sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)
        self._timesteps_host = torch.from_numpy(timesteps).to(dtype=torch.int64)
        self.timesteps = self._timesteps_host.to(device=device) if device is not None else self._timesteps_host
        self.num_inference_steps = len(timesteps)
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self._step_index = None
        self._begin_index = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        """:628-640 (second match when a timestep is duplicated), on the host copy of the table."""
        sched = self._timesteps_host if schedule_timesteps is None else torch.as_tensor(schedule_timesteps).cpu()
        t = int(timestep.item()) if isinstance(timestep, torch.Tensor) else int(timestep)
        idx = (sched == t).nonzero()
        if len(idx) == 0:
            raise IndexError(f"timestep {t} is not in the schedule")
        return idx[1 if len(idx) > 1 else 0].item()

    def _init_step_index(self, timestep):
        self._step_index = self.index_for_timestep(timestep) if self.begin_index is None else self._begin_index

    # ------------------------------------------------------------------------------------------
    # scalar side
    def _lambda(self, sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _bh_terms(self, sig_t, sig_s0, prev_sigmas, order, corrector):
        """Shared scalar algebra of :405-452 / :554-600.  Returns (a_x, h_phi_1 factor, B_h factor, rks, rhos) as python
        floats, evaluated with float32 torch scalars in the reference's order."""
        alpha_t, sigma_t = 1 - sig_t, sig_t
        alpha_s0, sigma_s0 = 1 - sig_s0, sig_s0
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks = []
        for i in range(1, order):
            lambda_si = self._lambda(prev_sigmas[i - 1])
            rks.append(((lambda_si - lambda_s0) / h).item())
        rks_t = torch.tensor(rks + [1.0])
        hh = -h if self.predict_x0 else h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        factorial_i = 1
        B_h = hh if self.config.solver_type == "bh1" else torch.expm1(hh)
        R, b = [], []
        for i in range(1, order + 1):
            R.append(torch.pow(rks_t, i - 1))
            b.append((h_phi_k * factorial_i / B_h).item())
            factorial_i *= i + 1
            h_phi_k = h_phi_k / hh - 1 / factorial_i
        R = torch.stack(R)
        b = torch.tensor(b)
        if corrector:
            rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
        elif order == 1:
            rhos = torch.zeros(0)
        elif order == 2:
            rhos = torch.tensor([0.5])
        else:
            rhos = torch.linalg.solve(R[:-1, :-1], b[:-1])
        if self.predict_x0:
            a_x = (sigma_t / sigma_s0).item()
            lead = alpha_t
        else:
            a_x = (alpha_t / alpha_s0).item()
            lead = sigma_t
        return a_x, float(lead * h_phi_1), float(lead * B_h), rks, [float(r) for r in rhos]

    def step_plan(self, step_index: int, lower_order_nums: int, have_last: bool, prev_order: int) -> StepPlan:
        """The coefficients of `step()` at `step_index` given the multistep warm-up state."""
        cfg = self.config
        sig = self.sigmas
        s_cur = sig[step_index]
        if self.predict_x0:      # x0 = sample - sigma * v      (:318-321)
            convert = (("sample", 1.0), ("model", -float(s_cur)))
        else:                    # eps = sample - (1 - sigma) * v   (:332-335)
            convert = (("sample", 1.0), ("model", -float(1 - s_cur)))
        correct = None
        use_corrector = step_index > 0 and (step_index - 1) not in self.disable_corrector and have_last
        if use_corrector:
            order = prev_order
            prev = [sig[step_index - (i + 1)] for i in range(1, order)]
            a_x, c_phi, c_B, rks, rhos = self._bh_terms(s_cur, sig[step_index - 1], prev, order, corrector=True)
            # x_t = a_x x_last - c_phi m0 - c_B (sum_k rho_k (m_k - m0) / r_k + rho_last (mt - m0))
            c_m0 = -c_phi + c_B * (sum(rhos[k] / rks[k] for k in range(order - 1)) + rhos[-1])
            terms = [("last", a_x), ("m0", c_m0)]
            terms += [(f"m{k + 1}", -c_B * rhos[k] / rks[k]) for k in range(order - 1)]
            terms.append(("mt", -c_B * rhos[-1]))
            correct = tuple(terms)
        if cfg.lower_order_final:
            this_order = min(cfg.solver_order, self.num_inference_steps - step_index)
        else:
            this_order = cfg.solver_order
        this_order = min(this_order, lower_order_nums + 1)
        assert this_order > 0
        # predictor: the history is now (mt, old m0, old m1, ...)
        prev = [sig[step_index - i] for i in range(1, this_order)]
        a_x, c_phi, c_B, rks, rhos = self._bh_terms(sig[step_index + 1], s_cur, prev, this_order, corrector=False)
        c_mt = -c_phi + c_B * sum(rhos[k] / rks[k] for k in range(this_order - 1))
        terms = [("cur", a_x), ("mt", c_mt)]
        terms += [(f"m{k}", -c_B * rhos[k] / rks[k]) for k in range(this_order - 1)]
        return StepPlan(step_index, this_order, convert, correct, tuple(terms))

    # ------------------------------------------------------------------------------------------
    # tensor side
    def step(self, model_output: torch.Tensor, timestep: Union[int, torch.Tensor], sample: torch.Tensor,
             return_dict: bool = True, generator=None):
        """:655-739.  `timestep` is only used to find the first step's index: pass the HOST copy's entry
        (`scheduler.timesteps_host[i]`) or an int to stay free of device syncs."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.step_index is None:
            self._init_step_index(timestep)
        plan = self.step_plan(self._step_index, self.lower_order_nums, self.last_sample is not None,
                              getattr(self, "this_order", 1))
        hist = [m for m in reversed(self.model_outputs) if m is not None]      # m0, m1, ...
        named = {"sample": sample, "model": model_output, "last": self.last_sample}
        named.update({f"m{k}": m for k, m in enumerate(hist)})

        def apply(terms):
            return ops.lincomb([named[n] for n, _ in terms], [c for _, c in terms])

        named["mt"] = apply(plan.convert)
        named["cur"] = apply(plan.correct) if plan.correct is not None else sample
        for i in range(self.config.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
            self.timestep_list[i] = self.timestep_list[i + 1]
        self.model_outputs[-1] = named["mt"]
        self.timestep_list[-1] = timestep
        self.this_order = plan.order
        self.last_sample = named["cur"]
        prev_sample = apply(plan.predict)
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        if not return_dict:
            return (prev_sample,)
        return SchedulerOutput(prev_sample=prev_sample)

    @property
    def timesteps_host(self) -> torch.Tensor:
        return self._timesteps_host

    def scale_model_input(self, sample: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        return sample
