"""CPU: the oracle (oracle/wan_oracle.py) against the golden vectors produced by the
reference itself (oracle/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

import self_forcing_amd as sfa
from oracle import wan_oracle as wo

GOLD = os.path.join(os.path.dirname(__file__), "golden")
LAT_H, LAT_W = 8, 12
FS = (LAT_H // 2) * (LAT_W // 2)


def load(name):
    return np.load(os.path.join(GOLD, name))


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cfg_of(shape: sfa.WanShape, **kw) -> wo.OracleConfig:
    d = dict(dim=shape.dim, ffn_dim=shape.ffn_dim, num_heads=shape.num_heads, num_layers=shape.num_layers,
             text_dim=shape.text_dim, local_attn_size=shape.local_attn_size, sink_size=shape.sink_size)
    d.update(kw)
    return wo.OracleConfig(**d)


@pytest.fixture(scope="module")
def ops():
    return load("ops.npz")


@pytest.fixture(scope="module")
def mods():
    return load("modules_reduced.npz")


@pytest.fixture(scope="module")
def weights():
    sd = sfa.synth_state_dict(sfa.WAN_REDUCED, seed=0)
    return {"f32": wo.prepare_weights(sd, torch.float32), "bf16": wo.prepare_weights(sd, torch.bfloat16)}


# ------------------------------------------------------------------------------- per-op
def test_sinusoid(ops):
    out = wo.sinusoidal_embedding_1d(256, T(ops["sinus_t"], torch.float64))
    assert np.allclose(out.numpy(), ops["sinus_out"], rtol=0, atol=1e-12)


def test_rope_tables(ops):
    cos, sin, split = wo.rope_tables(128)
    assert split == (22, 21, 21)
    rows = ops["rope_rows"]
    assert np.allclose(cos[rows].numpy(), ops["rope_cos"], atol=1e-12)
    assert np.allclose(sin[rows].numpy(), ops["rope_sin"], atol=1e-12)


def test_rope_apply(ops):
    cos, sin, split = wo.rope_tables(128)
    grid = tuple(int(v) for v in ops["rope_grid"])
    x = T(ops["rope_x"])
    y64 = wo.causal_rope_apply(x.double(), grid, cos, sin, split, int(ops["rope_start_frame"]))
    assert np.allclose(y64.numpy(), ops["rope_out_f64"], atol=1e-12)
    ybf = wo.causal_rope_apply(x.bfloat16(), grid, cos, sin, split, int(ops["rope_start_frame"]))
    assert torch.equal(ybf.float(), T(ops["rope_out_bf16"]))


def test_rmsnorm_layernorm(ops):
    x, w = T(ops["rms_x"]), T(ops["rms_w"])
    assert torch.equal(wo.rms_norm(x.bfloat16(), w.bfloat16(), 1e-6).float(), T(ops["rms_out_bf16"]))
    assert rel(wo.rms_norm(x, w, 1e-6), T(ops["rms_out_f32"])) < 1e-6
    assert torch.equal(wo.layer_norm(x.bfloat16(), 1e-6).float(), T(ops["ln_out_bf16"]))
    assert rel(wo.layer_norm(x, 1e-6), T(ops["ln_out_f32"])) < 1e-6


@pytest.mark.parametrize("shift", [5, 8])
def test_scheduler_tables(ops, shift):
    s = wo.FlowMatchTables(float(shift))
    assert np.array_equal(s.sigmas.numpy(), ops[f"sched{shift}_sigmas"])
    assert np.array_equal(s.timesteps.numpy(), ops[f"sched{shift}_timesteps"])
    assert np.array_equal(s.warp([1000, 750, 500, 250]).numpy(), ops[f"sched{shift}_warped"])


def test_add_noise_and_x0(ops):
    s = wo.FlowMatchTables(5.0)
    x0, eps, t = T(ops["an_x0"]).bfloat16(), T(ops["an_eps"]).bfloat16(), T(ops["an_t"])
    assert torch.equal(s.add_noise(x0, eps, t).float(), T(ops["an_out_bf16"]))
    ti = torch.from_numpy(ops["x0_ti"])
    assert torch.equal(s.add_noise(x0, eps, ti).float(), T(ops["an_out_int_bf16"]))
    flow = T(ops["x0_flow"])
    assert torch.equal(wo.flow_to_x0(s, flow.bfloat16(), x0, t).float(), T(ops["x0_out_bf16"]))
    assert torch.equal(wo.flow_to_x0(s, flow, x0.float(), t), T(ops["x0_out_f32"]))
    assert torch.equal(wo.flow_to_x0(s, flow, x0.float(), ti), T(ops["x0_out_int_f32"]))


# --------------------------------------------------------------------------- per-module
TOL = {"f32": 2e-5, "bf16": 1.2e-2}


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_patch_embed_unpatchify(mods, weights, tag):
    W = weights[tag]
    cfg = cfg_of(sfa.WAN_REDUCED)
    dt = W["patch_embedding.weight"].dtype
    tok, grid = wo.patch_embed(W, cfg, T(mods["pe_x"]).to(dt))
    assert grid == (2, LAT_H // 2, LAT_W // 2)
    assert rel(tok.float(), T(mods[f"pe_out_{tag}"])) < TOL[tag]
    # unpatchify alone: feed the head output directly
    hx = T(mods["unp_x"]).to(dt)
    f, h, w = 2, LAT_H // 2, LAT_W // 2
    y = hx.reshape(1, f, h, w, 1, 2, 2, 16).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(1, 16, f, 2 * h, 2 * w)
    assert torch.equal(y[0].float(), T(mods[f"unp_out_{tag}"]))


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_self_attention_rolling_cache(mods, weights, tag):
    W = weights[tag]
    dt = W["patch_embedding.weight"].dtype
    cfg = cfg_of(sfa.WAN_REDUCED, local_attn_size=3, sink_size=1)
    rope = wo.rope_tables(128)
    kv = wo.init_kv_cache(cfg, 1, 3 * FS, dt)[0]
    grid = (1, LAT_H // 2, LAT_W // 2)
    for i, st in enumerate(mods["sa_starts"]):
        x = T(mods["sa_x"][i]).to(dt)
        y = wo.self_attention(W, "blocks.0.self_attn.", cfg, x, grid, rope, kv, int(st) * FS)
        assert rel(y.float(), T(mods[f"sa_y_{tag}"][i])) < TOL[tag], i
        assert int(kv["local_end_index"]) == int(mods["sa_local_end"][i])
        assert int(kv["global_end_index"]) == int(mods["sa_global_end"][i])
    assert rel(kv["k"].float(), T(mods[f"sa_k_final_{tag}"])) < TOL[tag]
    assert rel(kv["v"].float(), T(mods[f"sa_v_final_{tag}"])) < TOL[tag]


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_cross_attention(mods, weights, tag):
    W = weights[tag]
    dt = W["patch_embedding.weight"].dtype
    cfg = cfg_of(sfa.WAN_REDUCED)
    cache = {"k": None, "v": None, "is_init": False}
    x, ctx = T(mods["ca_x"]).to(dt), T(mods["ca_ctx"]).to(dt)
    y = wo.cross_attention(W, "blocks.1.cross_attn.", cfg, x, ctx, cache)
    assert rel(y.float(), T(mods[f"ca_y_{tag}"])) < TOL[tag]
    assert cache["is_init"]
    n = mods[f"ca_k_{tag}"].shape[1]
    assert rel(cache["k"][:, :n].float(), T(mods[f"ca_k_{tag}"])) < TOL[tag]
    assert rel(cache["v"][:, :n].float(), T(mods[f"ca_v_{tag}"])) < TOL[tag]
    # second call must use the cache, not the (zeroed) context
    y2 = wo.cross_attention(W, "blocks.1.cross_attn.", cfg, x, ctx * 0, cache)
    assert rel(y2.float(), T(mods[f"ca_y2_{tag}"])) < TOL[tag]


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_block(mods, weights, tag):
    W = weights[tag]
    dt = W["patch_embedding.weight"].dtype
    cfg = cfg_of(sfa.WAN_REDUCED)
    kv = wo.init_kv_cache(cfg, 1, 4 * FS, dt)[0]
    cc = {"k": None, "v": None, "is_init": False}
    y = wo.attention_block(W, 0, cfg, T(mods["blk_x"]).to(dt), T(mods["blk_e0"]).to(dt),
                           (2, LAT_H // 2, LAT_W // 2), wo.rope_tables(128), T(mods["ca_ctx"]).to(dt), kv, cc, 0)
    assert rel(y.float(), T(mods[f"blk_y_{tag}"])) < TOL[tag]


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_forward_inference(mods, weights, tag):
    W = weights[tag]
    dt = W["patch_embedding.weight"].dtype
    cfg = cfg_of(sfa.WAN_REDUCED)
    kv = wo.init_kv_cache(cfg, 1, 5 * FS, dt)
    ca = wo.init_crossattn_cache(cfg, 1, dt)
    pe = T(mods["fwd_pe"]).to(dt)
    y1 = wo.forward_inference(W, cfg, T(mods["fwd_x1"]).to(dt), T(mods["fwd_t1"]), pe, kv, ca, 0)
    y2 = wo.forward_inference(W, cfg, T(mods["fwd_x2"]).to(dt), torch.from_numpy(mods["fwd_t2"]), pe, kv, ca, 2 * FS)
    assert rel(y1.float(), T(mods[f"fwd_y1_{tag}"])) < TOL[tag]
    assert rel(y2.float(), T(mods[f"fwd_y2_{tag}"])) < TOL[tag]
    assert rel(kv[0]["k"].float(), T(mods[f"fwd_k0_{tag}"])) < TOL[tag]
    assert rel(kv[1]["v"].float(), T(mods[f"fwd_v1_{tag}"])) < TOL[tag]
    n = mods[f"fwd_ck1_{tag}"].shape[1]
    assert rel(ca[1]["k"][:, :n].float(), T(mods[f"fwd_ck1_{tag}"])) < TOL[tag]


# ----------------------------------------------------------------------------- rollouts
SCEN = {  # must match oracle/make_golden.py ROLLOUT_SCENARIOS
    "nfpb1": (1, False, 5.0, -1, 0), "nfpb3": (3, False, 5.0, -1, 0), "iff": (3, True, 8.0, -1, 0),
    "ext": (3, False, 5.0, -1, 0), "i2v": (3, True, 5.0, -1, 0), "roll": (1, False, 5.0, 3, 1),
}


@pytest.mark.parametrize("name", list(SCEN))
@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_rollout(weights, name, tag):
    R = load("rollouts_reduced.npz")
    W = weights[tag]
    dt = W["patch_embedding.weight"].dtype
    nfpb, iff, shift, las, sink = SCEN[name]
    cfg = cfg_of(sfa.WAN_REDUCED, local_attn_size=las, sink_size=sink)
    args = wo.RolloutArgs(num_frame_per_block=nfpb, independent_first_frame=iff, timestep_shift=shift)
    eps = [T(R[f"{name}_eps{j}"]) for j in range(int(R[f"{name}_neps"]))]
    initial = T(R[f"{name}_initial"]).to(dt) if f"{name}_initial" in R else None
    lat = wo.rollout(W, cfg, args, T(R[f"{name}_noise"]).to(dt), T(R[f"{name}_pe"]).to(dt), eps, initial)
    # bf16: two independent bf16 implementations of a 4-step rollout; the reference's own
    # bf16-vs-fp32 distance at this shape is 3.5e-3
    tol = 5e-5 if tag == "f32" else 1.5e-2
    assert rel(lat.float(), T(R[f"{name}_lat_{tag}"])) < tol


# ------------------------------------------------------------------------------- BASELINE configs[0] at full shape
def test_tconfig_full_shape_rollout_oracle_vs_reference():
    """The oracle's float32 rollout of the T config at the FULL Wan-1.3B shape (2 chunks x 5 forwards of 1560 tokens,
    ~1.5 minutes on 8 cores) against the latents the reference itself produced (oracle/make_golden.py --only tconfig)."""
    G = load("tconfig_1p3b.npz")
    g = torch.Generator().manual_seed(int(G["input_seed"]))
    bf = lambda shape: torch.randn(shape, generator=g).to(torch.bfloat16)  # noqa: E731
    noise = bf((1, 2, 16, 60, 104))
    pe = bf((1, 512, sfa.WAN_1_3B.text_dim))
    pe[:, 93:] = 0
    eps = [bf((1, 16, 60, 104)) for _ in range(6)]
    assert noise.double().sum().item() == float(G["noise_checksum"]), "torch CPU generator stream changed; regenerate the fixture"
    W = wo.prepare_weights(sfa.synth_state_dict(sfa.WAN_1_3B, seed=0), torch.float32)
    cfg = cfg_of(sfa.WAN_1_3B)
    args = wo.RolloutArgs(num_frame_per_block=1, independent_first_frame=True, timestep_shift=8.0)
    with torch.no_grad():
        lat = wo.rollout(W, cfg, args, noise.float(), pe.float(), [e.float() for e in eps], cache_tokens=2 * 1560)
    assert rel(lat, T(G["lat_f32"])) < 2e-5
