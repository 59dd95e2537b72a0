"""CPU: the 50-step CFG sampler.  (1) oracle/unipc_oracle.py against the golden vectors the reference
itself produced (oracle/make_golden_unipc.py) -- this pins the oracle; (2) the product's host-side scheduler
(self_forcing_amd/unipc.py: sigma tables, step order logic, collapsed linear coefficients) against the same vectors,
with its one device primitive (`ops.lincomb`) replaced by a float32 stand-in for the duration of the test."""
import os

import numpy as np
import pytest
import torch

import self_forcing_amd as sfa
from oracle import unipc_oracle as uo
from oracle import wan_oracle as wo

GOLD = os.path.join(os.path.dirname(__file__), "golden")
LAT_H, LAT_W = 8, 12

STEP_CONFIGS = {
    "s50_shift5": (50, 5.0, {}),
    "s8_shift3_order3": (8, 3.0, {"solver_order": 3}),
    "s6_shift1_order1": (6, 1.0, {"solver_order": 1}),
    "s10_bh1": (10, 5.0, {"solver_type": "bh1"}),
    "s10_eps": (10, 5.0, {"predict_x0": False}),
    "s10_nocorr01": (10, 8.0, {"disable_corrector": [0, 1]}),
    "s12_nolof": (12, 5.0, {"lower_order_final": False, "solver_order": 2}),
}
ROLLOUTS = {
    "cfg_nfpb3": (3, False, 5.0, 3.0, 50),
    "cfg_iff": (3, True, 8.0, 5.0, 10),
    "cfg_ext": (3, False, 5.0, 6.0, 12),
    "cfg_start2": (1, False, 5.0, 4.0, 8),      # start_frame_index = 2
}
START_FRAME = {"cfg_start2": 2}


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def steps():
    return np.load(os.path.join(GOLD, "unipc_steps.npz"))


@pytest.mark.parametrize("name", list(STEP_CONFIGS))
def test_oracle_tables(steps, name):
    n, shift, _ = STEP_CONFIGS[name]
    sig, ts = uo.sampling_sigmas(n, shift)
    assert np.array_equal(ts.numpy(), steps[f"{name}_timesteps"])       # int64 truncation included
    assert np.allclose(sig.numpy(), steps[f"{name}_sigmas"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("name", list(STEP_CONFIGS))
def test_oracle_trajectory(steps, name):
    n, shift, kw = STEP_CONFIGS[name]
    st = uo.new_state(n, shift, **kw)
    x, v, traj = T(steps[f"{name}_x0"]), T(steps[f"{name}_v"]), steps[f"{name}_traj"]
    for i in range(traj.shape[0]):
        x = uo.unipc_step(st, v[i], st.timesteps[i], x)
        assert rel(x, T(traj[i])) < 2e-6, (name, i)


# -------------------------------------------------------------------------------- host scheduler of the product
@pytest.fixture()
def host_lincomb(monkeypatch):
    def lincomb(tensors, coefs, out=None):
        acc = None
        for t, c in zip(tensors, coefs):
            term = np.float32(c) * t.float()
            acc = term if acc is None else acc + term
        return acc
    monkeypatch.setattr(sfa.unipc.ops, "lincomb", lincomb)


@pytest.mark.parametrize("name", list(STEP_CONFIGS))
def test_host_scheduler_against_reference_trajectory(steps, name, host_lincomb):
    n, shift, kw = STEP_CONFIGS[name]
    sch = sfa.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False, **kw)
    sch.set_timesteps(n, device="cpu", shift=shift)
    assert np.array_equal(sch.timesteps.numpy(), steps[f"{name}_timesteps"])
    assert np.array_equal(sch.sigmas.numpy(), steps[f"{name}_sigmas"])
    x, v, traj = T(steps[f"{name}_x0"]), T(steps[f"{name}_v"]), steps[f"{name}_traj"]
    for i in range(traj.shape[0]):
        x = sch.step(v[i], sch.timesteps[i], x, return_dict=False)[0]
        # collapsed coefficients: same algebra, different association -> float32 round-off of a few ulp per term
        assert rel(x, T(traj[i])) < 2e-5, (name, i)
    assert sch.step_index == traj.shape[0]


def test_host_scheduler_rejects_what_it_does_not_implement():
    with pytest.raises(NotImplementedError):
        sfa.FlowUniPCMultistepScheduler(thresholding=True)
    with pytest.raises(NotImplementedError):
        sfa.FlowUniPCMultistepScheduler(use_dynamic_shifting=True)
    with pytest.raises(NotImplementedError):
        sfa.FlowUniPCMultistepScheduler(solver_type="nope")
    sch = sfa.FlowUniPCMultistepScheduler()
    with pytest.raises(ValueError):
        sch.step(torch.zeros(8), 999, torch.zeros(8))       # set_timesteps not called


def test_step_plan_final_step_returns_the_x0_prediction():
    """sigma -> 0 on the last step: h = inf, and the update degenerates to prev_sample = x0 prediction."""
    sch = sfa.FlowUniPCMultistepScheduler(shift=1)
    sch.set_timesteps(50, shift=5.0)
    plan = sch.step_plan(49, lower_order_nums=2, have_last=True, prev_order=2)
    assert plan.order == 1
    assert dict(plan.predict) == {"cur": 0.0, "mt": 1.0}
    assert all(np.isfinite(c) for _, c in plan.correct)


# -------------------------------------------------------------------------------- the chunk loop
@pytest.fixture(scope="module")
def weights():
    sd = sfa.synth_state_dict(sfa.WAN_REDUCED, seed=0)
    return {"f32": wo.prepare_weights(sd, torch.float32), "bf16": wo.prepare_weights(sd, torch.bfloat16)}


@pytest.mark.parametrize("tag", ["f32", "bf16"])
@pytest.mark.parametrize("name", list(ROLLOUTS))
def test_oracle_cfg_rollout(weights, name, tag):
    R = np.load(os.path.join(GOLD, "unipc_rollouts.npz"))
    nfpb, iff, shift, g, nsteps = ROLLOUTS[name]
    W = weights[tag]
    dt = W["patch_embedding.weight"].dtype
    s = sfa.WAN_REDUCED
    cfg = wo.OracleConfig(dim=s.dim, ffn_dim=s.ffn_dim, num_heads=s.num_heads, num_layers=s.num_layers,
                          text_dim=s.text_dim, local_attn_size=-1, sink_size=0)
    args = uo.CfgRolloutArgs(num_frame_per_block=nfpb, independent_first_frame=iff, timestep_shift=shift,
                             guidance_scale=g, sampling_steps=nsteps)
    initial = T(R[f"{name}_initial"]).to(dt) if f"{name}_initial" in R else None
    start = START_FRAME.get(name, 0)
    noise = T(R[f"{name}_noise"]).to(dt)
    tokens = (noise.shape[1] + (0 if initial is None else initial.shape[1]) + start) * (LAT_H // 2) * (LAT_W // 2)
    lat = uo.cfg_rollout(W, cfg, args, noise, T(R[f"{name}_pe"]).to(dt), T(R[f"{name}_ne"]).to(dt), initial,
                         cache_tokens=tokens, start_frame_index=start)
    # bf16: the reference's own bf16-vs-fp32 distance on these rollouts is 0.9-2.7e-2 (guidance amplifies the
    # rounding of 100 forwards per chunk); two bf16 implementations agree to about that
    tol = 5e-6 if tag == "f32" else 2.5e-2
    assert rel(lat.float(), T(R[f"{name}_lat_{tag}"])) < tol
