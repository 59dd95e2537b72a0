"""CPU oracle for the 50-step classifier-free-guidance causal sampler.

TEST INFRASTRUCTURE ONLY -- only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it, never the product package.

Restates, with torch CPU tensors as the array library:
  * `FlowUniPCMultistepScheduler` as the causal sampler configures it (wan/utils/fm_solvers_unipc.py:77-133
    constructor with shift=1; `set_timesteps(n, shift=s)` :160-227; `convert_model_output` :279-347;
    `multistep_uni_p_bh_update` :350-484; `multistep_uni_c_bh_update` :486-626; `step` :655-739), as a small state
    machine with explicit history (`UniPCState`), tensor expressions evaluated term by term in the sample's dtype
    exactly as the reference writes them (so a bf16 run rounds where the reference's does, a float32 run is the math);
  * the chunk loop of `CausalDiffusionInferencePipeline.inference` (pipeline/causal_diffusion_inference.py:175-457):
    two KV / cross-attention cache sets (prompt / negative prompt), per chunk a FRESH scheduler, per step two
    generator calls + the guidance blend (:423-424) + one scheduler step, then a timestep-0 pass over both caches.
    The fork's image / pose conditioning (CLIP features, `y`, the dwpose convolution stacks, :305-358) is outside
    the path; `add_condition` tokens are taken already embedded.

Parity status: PINNED.  `oracle/make_golden_unipc.py` imports the reference's scheduler and pipeline on CPU (the
diffusers mixins they inherit from are absent here and are replaced by a config-holding stand-in with no numerical
content, see that script) and stores its outputs under `tests/golden/unipc_*.npz`;
`tests/test_unipc_oracle_golden.py` checks this file against them.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

Tensor = torch.Tensor


def sampling_sigmas(num_steps: int, shift: float, num_train_timesteps: int = 1000):
    """(sigmas float32 [n+1] with a trailing 0, timesteps int64 [n]) of set_timesteps (:160-227)."""
    alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
    table = torch.from_numpy(1.0 - alphas).to(torch.float32)        # constructor, shift = 1 (:104-117)
    table = 1.0 * table / (1 + (1.0 - 1) * table)
    smax, smin = table[0].item(), table[-1].item()
    s = np.linspace(smax, smin, num_steps + 1).copy()[:-1]
    s = shift * s / (1 + (shift - 1) * s)
    timesteps = torch.from_numpy(s * num_train_timesteps).to(torch.int64)    # truncation, :205-210
    sigmas = torch.from_numpy(np.concatenate([s, [0]]).astype(np.float32))
    return sigmas, timesteps


@dataclass
class UniPCState:
    sigmas: Tensor                   # float32 [n+1]
    timesteps: Tensor                # int64 [n]
    solver_order: int = 2
    solver_type: str = "bh2"
    predict_x0: bool = True
    lower_order_final: bool = True
    disable_corrector: Sequence[int] = ()
    outputs: List[Tensor] = field(default_factory=list)     # converted model outputs, oldest first, <= solver_order
    lower_order_nums: int = 0
    last_sample: Optional[Tensor] = None
    this_order: int = 1
    step_index: Optional[int] = None


def new_state(num_steps: int, shift: float, **kw) -> UniPCState:
    sigmas, timesteps = sampling_sigmas(num_steps, shift)
    return UniPCState(sigmas=sigmas, timesteps=timesteps, **kw)


def _lam(sigma: Tensor) -> Tensor:
    return torch.log(1 - sigma) - torch.log(sigma)


def _bh(st: UniPCState, sigma_t: Tensor, sigma_s0: Tensor, prev_sigmas: Sequence[Tensor], order: int):
    """h, r_k, h*phi_1, B(h) and the (R, b) system shared by UniP and UniC (:405-452, :554-600)."""
    lam_s0 = _lam(sigma_s0)
    h = _lam(sigma_t) - lam_s0
    rks = [(_lam(s) - lam_s0) / h for s in prev_sigmas]
    rks_t = torch.tensor([float(r) for r in rks] + [1.0])
    hh = -h if st.predict_x0 else h
    h_phi_1 = torch.expm1(hh)
    h_phi_k = h_phi_1 / hh - 1
    B_h = hh if st.solver_type == "bh1" else torch.expm1(hh)
    R, b, fact = [], [], 1
    for i in range(1, order + 1):
        R.append(torch.pow(rks_t, i - 1))
        b.append(float(h_phi_k * fact / B_h))
        fact *= i + 1
        h_phi_k = h_phi_k / hh - 1 / fact
    return rks, h_phi_1, B_h, torch.stack(R), torch.tensor(b)


def _weighted(rhos: Tensor, D1s: List[Tensor]):
    if not D1s:
        return 0
    acc = None
    for r, d in zip(rhos, D1s):          # einsum("k,bkc...->bc...") as an explicit sum
        acc = r.to(d.dtype) * d if acc is None else acc + r.to(d.dtype) * d
    return acc


def unipc_step(st: UniPCState, model_output: Tensor, timestep, sample: Tensor) -> Tensor:
    """One `scheduler.step(model_output, t, sample)`; returns prev_sample and advances `st`."""
    if st.step_index is None:           # index_for_timestep (:628-640): second match if duplicated
        idx = (st.timesteps == int(timestep)).nonzero()
        st.step_index = idx[1 if len(idx) > 1 else 0].item()
    k = st.step_index
    sig = st.sigmas
    dt = sample.dtype
    # convert_model_output (:318-321 / :332-335)
    if st.predict_x0:
        m_t = sample - sig[k] * model_output
    else:
        m_t = sample - (1 - sig[k]) * model_output
    # corrector (:486-626) on the step that produced `sample`
    if k > 0 and (k - 1) not in st.disable_corrector and st.last_sample is not None:
        order = st.this_order
        m0 = st.outputs[-1]
        sigma_t, sigma_s0 = sig[k], sig[k - 1]
        prev = [sig[k - (i + 1)] for i in range(1, order)]
        rks, h_phi_1, B_h, R, b = _bh(st, sigma_t, sigma_s0, prev, order)
        D1s = [(st.outputs[-(i + 1)] - m0) / rks[i - 1] for i in range(1, order)]
        rhos_c = torch.tensor([0.5], dtype=dt) if order == 1 else torch.linalg.solve(R, b).to(dt)
        if st.predict_x0:
            x_t_ = sigma_t / sigma_s0 * st.last_sample - (1 - sigma_t) * h_phi_1 * m0
            lead = 1 - sigma_t
        else:
            x_t_ = (1 - sigma_t) / (1 - sigma_s0) * st.last_sample - sigma_t * h_phi_1 * m0
            lead = sigma_t
        sample = (x_t_ - lead * B_h * (_weighted(rhos_c[:-1], D1s) + rhos_c[-1] * (m_t - m0))).to(dt)
    # history shift (:704-708)
    st.outputs.append(m_t)
    if len(st.outputs) > st.solver_order:
        st.outputs.pop(0)
    n = len(st.timesteps)
    this_order = min(st.solver_order, n - k) if st.lower_order_final else st.solver_order
    st.this_order = min(this_order, st.lower_order_nums + 1)
    st.last_sample = sample
    # predictor (:350-484)
    order = st.this_order
    m0 = st.outputs[-1]
    sigma_t, sigma_s0 = sig[k + 1], sig[k]
    prev = [sig[k - i] for i in range(1, order)]
    rks, h_phi_1, B_h, R, b = _bh(st, sigma_t, sigma_s0, prev, order)
    D1s = [(st.outputs[-(i + 1)] - m0) / rks[i - 1] for i in range(1, order)]
    if order == 2:
        rhos_p = torch.tensor([0.5], dtype=dt)
    elif order > 2:
        rhos_p = torch.linalg.solve(R[:-1, :-1], b[:-1]).to(dt)
    else:
        rhos_p = None
    if st.predict_x0:
        x_t_ = sigma_t / sigma_s0 * sample - (1 - sigma_t) * h_phi_1 * m0
        lead = 1 - sigma_t
    else:
        x_t_ = (1 - sigma_t) / (1 - sigma_s0) * sample - sigma_t * h_phi_1 * m0
        lead = sigma_t
    prev_sample = (x_t_ - lead * B_h * (_weighted(rhos_p, D1s) if D1s else 0)).to(dt)
    if st.lower_order_nums < st.solver_order:
        st.lower_order_nums += 1
    st.step_index += 1
    return prev_sample


# --------------------------------------------------------------------------------------
# the chunk loop
# --------------------------------------------------------------------------------------
@dataclass
class CfgRolloutArgs:
    num_frame_per_block: int = 3
    independent_first_frame: bool = False
    timestep_shift: float = 5.0
    guidance_scale: float = 3.0
    sampling_steps: int = 50


def cfg_rollout(W, cfg, args: CfgRolloutArgs, noise: Tensor, prompt_embeds: Tensor, negative_embeds: Tensor,
                initial_latent: Optional[Tensor] = None, cache_tokens: Optional[int] = None,
                start_frame_index: int = 0, pose_emb: Optional[Tensor] = None) -> Tensor:
    """CausalDiffusionInferencePipeline.inference up to the latents (causal_diffusion_inference.py:175-457), without
    image conditioning; `pose_emb` [B, C_pose, F_total, h, w] is the already-embedded pose volume, sliced per chunk and
    flattened to tokens as :380-394 does.  Returns `output` [B, F_total, C, H, W] in noise.dtype."""
    from oracle import wan_oracle as wo
    B, Fn, C, H, Wd = noise.shape
    nf = args.num_frame_per_block
    if not args.independent_first_frame or initial_latent is not None:
        assert Fn % nf == 0
        num_blocks = Fn // nf
    else:
        assert (Fn - 1) % nf == 0
        num_blocks = (Fn - 1) // nf
    n_in = initial_latent.shape[1] if initial_latent is not None else 0
    fs = (H // cfg.patch_size[1]) * (Wd // cfg.patch_size[2])
    dtype = W["patch_embedding.weight"].dtype
    if cache_tokens is None:
        cache_tokens = (cfg.local_attn_size if cfg.local_attn_size != -1 else Fn + n_in) * fs
    caches = {tag: (wo.init_kv_cache(cfg, B, cache_tokens, dtype), wo.init_crossattn_cache(cfg, B, dtype))
              for tag in ("pos", "neg")}
    embeds = {"pos": prompt_embeds, "neg": negative_embeds}
    sched = wo.FlowMatchTables(args.timestep_shift)      # only feeds the wrapper's (unused here) x0 output
    rope = wo.rope_tables(cfg.head_dim)
    out = torch.zeros(B, Fn + n_in, C, H, Wd, dtype=noise.dtype)

    def gen(tag, xin, ts, start_frame, pose=None):
        kv, ca = caches[tag]
        flow, _ = wo.wrapper_forward(W, cfg, sched, xin.to(dtype), embeds[tag], ts, kv, ca, start_frame * fs, rope,
                                     add_condition=pose)
        return flow

    cur = start_frame_index       # RoPE / global position
    cstart = 0                    # position in `output` and in `noise`
    if initial_latent is not None:      # :239-297
        t0 = torch.zeros(B, 1, dtype=torch.int64)
        if args.independent_first_frame:
            assert (n_in - 1) % nf == 0
            n_in_blocks = (n_in - 1) // nf
            out[:, :1] = initial_latent[:, :1]
            gen("pos", initial_latent[:, :1], t0, cur)
            gen("neg", initial_latent[:, :1], t0, cur)
            cur += 1
            cstart += 1
        else:
            assert n_in % nf == 0
            n_in_blocks = n_in // nf
        for _ in range(n_in_blocks):
            ref = initial_latent[:, cstart:cstart + nf]
            out[:, cstart:cstart + nf] = ref
            gen("pos", ref, t0, cur)
            gen("neg", ref, t0, cur)
            cur += nf
            cstart += nf

    chunks = [nf] * num_blocks
    if args.independent_first_frame and initial_latent is None:
        chunks = [1] + chunks
    for f in chunks:
        latents = noise[:, cstart - n_in:cstart + f - n_in]
        pose = None
        if pose_emb is not None:      # 'b c f h w -> b (f h w) c'
            pose = pose_emb[:, :, cur:cur + f].permute(0, 2, 3, 4, 1).flatten(1, 3).to(dtype)
        st = new_state(args.sampling_steps, args.timestep_shift)          # :376, :517-525
        for t in st.timesteps:
            ts = t * torch.ones(B, f, dtype=torch.float32)
            cond = gen("pos", latents, ts, cur, pose)
            uncond = gen("neg", latents, ts, cur, pose)
            flow = uncond + args.guidance_scale * (cond - uncond)         # :423-424
            latents = unipc_step(st, flow, t, latents)
        out[:, cstart:cstart + f] = latents
        # :438-455; the conditional dicts still hold this chunk's `add_condition` (set at :396-397), so the pose
        # tokens take part in the cache-refresh pass too
        gen("pos", latents, ts * 0, cur, pose)
        gen("neg", latents, ts * 0, cur, pose)
        cur += f
        cstart += f
    return out
