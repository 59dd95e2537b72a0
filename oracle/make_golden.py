"""Generate the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--full]

The reference's code is imported (never copied) through `oracle/ref_shim.py`; weights
come from the seeded recipe `self_forcing_amd.synth_state_dict`, so the fixtures hold
only inputs, seeds and the reference's outputs.  `--full` additionally runs one
1.3B-shape forward (about 2 minutes and 10 GB of RAM).

Fixture files (all numpy .npz, float32/float64/int64 arrays; bf16 tensors are stored
exactly as float32):
  ops.npz              per-op vectors (sinusoid, RoPE, RMSNorm, scheduler, x0, patches)
  modules_reduced.npz  self-attn cache sequences, cross-attn, block, full forward
  rollouts_reduced.npz five rollout scenarios, fp32 and bf16
  full_1p3b.npz        (--full) one full-shape single-frame forward
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_shim  # noqa: E402
import self_forcing_amd as sfa  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
LAT_H, LAT_W = 8, 12  # reduced latent -> 4 x 6 = 24 tokens / frame


def f32(t):
    return t.detach().to(torch.float32).cpu().numpy()


def f64(t):
    return t.detach().to(torch.float64).cpu().numpy()


def bf16_randn(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(torch.bfloat16)


def build_model(ns, shape: sfa.WanShape, sd, dtype, local_attn_size=-1, sink_size=0):
    m = ns.CausalWanModel(dim=shape.dim, ffn_dim=shape.ffn_dim, num_heads=shape.num_heads,
                          num_layers=shape.num_layers, text_dim=shape.text_dim,
                          freq_dim=shape.freq_dim, in_dim=shape.in_dim, out_dim=shape.out_dim,
                          local_attn_size=local_attn_size, sink_size=sink_size)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("pose_proj") for k in missing), missing
    m.to(dtype).eval()
    ns.set_attention_dtype("bf16" if dtype == torch.bfloat16 else "input")
    return m


def fresh_caches(shape: sfa.WanShape, batch, cache_tokens, dtype):
    kv = [{"k": torch.zeros(batch, cache_tokens, shape.num_heads, shape.head_dim, dtype=dtype),
           "v": torch.zeros(batch, cache_tokens, shape.num_heads, shape.head_dim, dtype=dtype),
           "global_end_index": torch.tensor([0], dtype=torch.long),
           "local_end_index": torch.tensor([0], dtype=torch.long)} for _ in range(shape.num_layers)]
    ca = [{"k": torch.zeros(batch, shape.text_len, shape.num_heads, shape.head_dim, dtype=dtype),
           "v": torch.zeros(batch, shape.text_len, shape.num_heads, shape.head_dim, dtype=dtype),
           "is_init": False} for _ in range(shape.num_layers)]
    return kv, ca


# --------------------------------------------------------------------------------------
def gen_ops(ns):
    out = {}
    g = torch.Generator().manual_seed(1234)
    wm, wcm = ns.model_mod, ns.causal_mod

    # sinusoidal_embedding_1d (model.py:15-25) incl. warped non-integer timesteps
    t = torch.tensor([0.0, 250.0, 625.0, 833.3333129882812, 937.5, 1000.0], dtype=torch.float32)
    out["sinus_t"] = f64(t)
    out["sinus_out"] = f64(wm.sinusoidal_embedding_1d(256, t))

    # rope tables as built in CausalWanModel.__init__ (causal_model.py:481-488)
    d = 128
    freqs = torch.cat([wm.rope_params(1024, d - 4 * (d // 6)), wm.rope_params(1024, 2 * (d // 6)),
                       wm.rope_params(1024, 2 * (d // 6))], dim=1)
    rows = torch.tensor(list(range(32)) + [100, 511, 1023])
    out["rope_rows"] = rows.numpy()
    out["rope_cos"] = f64(freqs.real[rows])
    out["rope_sin"] = f64(freqs.imag[rows])

    # causal_rope_apply with start_frame > 0 (causal_model.py:28-56)
    fgrid = (2, 4, 6)
    L = fgrid[0] * fgrid[1] * fgrid[2]
    x = bf16_randn((1, L, 4, d), g)
    grid = torch.tensor([list(fgrid)], dtype=torch.long)
    out["rope_x"] = f32(x)
    out["rope_grid"] = np.array(fgrid)
    out["rope_start_frame"] = np.array(3)
    out["rope_out_bf16"] = f32(wcm.causal_rope_apply(x, grid, freqs, start_frame=3))
    out["rope_out_f64"] = f64(wcm.causal_rope_apply(x.double(), grid, freqs, start_frame=3))

    # WanRMSNorm (model.py:70-86)
    rn = wm.WanRMSNorm(512, eps=1e-6)
    w = (1 + 0.1 * torch.randn(512, generator=g)).to(torch.bfloat16)
    xr = bf16_randn((5, 512), g, 2.0)
    rn.weight.data = w.clone()
    out["rms_x"] = f32(xr)
    out["rms_w"] = f32(w)
    out["rms_out_bf16"] = f32(rn(xr))
    rn.weight.data = w.float()
    out["rms_out_f32"] = f32(rn(xr.float()))

    # WanLayerNorm without / with affine (model.py:89-99)
    ln = wm.WanLayerNorm(512, eps=1e-6)
    out["ln_out_bf16"] = f32(ln(xr))
    out["ln_out_f32"] = f32(ln(xr.float()))

    # FlowMatchScheduler tables + warped step lists (scheduler.py:118-141,
    # causal_inference.py:27-31)
    for shift in (5.0, 8.0):
        s = ns.FlowMatchScheduler(shift=shift, sigma_min=0.0, extra_one_step=True)
        s.set_timesteps(1000, training=True)
        tag = str(int(shift))
        out[f"sched{tag}_sigmas"] = f32(s.sigmas)
        out[f"sched{tag}_timesteps"] = f32(s.timesteps)
        steps = torch.tensor([1000, 750, 500, 250], dtype=torch.long)
        table = torch.cat((s.timesteps.cpu(), torch.tensor([0], dtype=torch.float32)))
        out[f"sched{tag}_warped"] = f32(table[1000 - steps])

    # add_noise (scheduler.py:159-176) and flow -> x0 (wan_wrapper.py:204-228)
    s = ns.FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    s.set_timesteps(1000, training=True)
    x0 = bf16_randn((3, 16, 8, 12), g)
    eps = bf16_randn((3, 16, 8, 12), g)
    ts = torch.tensor([937.5, 833.3333129882812, 625.0], dtype=torch.float32)
    out["an_x0"] = f32(x0)
    out["an_eps"] = f32(eps)
    out["an_t"] = f32(ts)
    out["an_out_bf16"] = f32(s.add_noise(x0, eps, ts))
    w = ns.WanDiffusionWrapper.__new__(ns.WanDiffusionWrapper)
    torch.nn.Module.__init__(w)
    w.scheduler = s
    flow = bf16_randn((3, 16, 8, 12), g)
    out["x0_flow"] = f32(flow)
    out["x0_out_bf16"] = f32(w._convert_flow_pred_to_x0(flow, x0, ts))
    out["x0_out_f32"] = f32(w._convert_flow_pred_to_x0(flow.float(), x0.float(), ts))
    # integer (un-warped) timesteps take the nearest table entry
    ti = torch.tensor([750, 500, 250], dtype=torch.int64)
    out["x0_ti"] = ti.numpy()
    out["x0_out_int_f32"] = f32(w._convert_flow_pred_to_x0(flow.float(), x0.float(), ti))
    out["an_out_int_bf16"] = f32(s.add_noise(x0, eps, ti))

    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)
    print("ops.npz", len(out), "arrays")


# --------------------------------------------------------------------------------------
def gen_modules(ns):
    shape = sfa.WAN_REDUCED
    sd = sfa.synth_state_dict(shape, seed=0)
    out = {"weights_seed": np.array(0)}
    g = torch.Generator().manual_seed(4321)
    fs = (LAT_H // 2) * (LAT_W // 2)
    C = shape.dim

    for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        # ---- patch embedding + unpatchify through the reference modules
        m = build_model(ns, shape, sd, dtype)
        xin = bf16_randn((1, 16, 2, LAT_H, LAT_W), torch.Generator().manual_seed(7)).to(dtype)
        tok = m.patch_embedding(xin).flatten(2).transpose(1, 2)
        out[f"pe_x"] = f32(xin)
        out[f"pe_out_{tag}"] = f32(tok)
        hx = bf16_randn((2 * fs, 64), torch.Generator().manual_seed(8)).to(dtype)
        out["unp_x"] = f32(hx)
        out[f"unp_out_{tag}"] = f32(m.unpatchify([hx], torch.tensor([[2, LAT_H // 2, LAT_W // 2]]))[0])

        # ---- self attention over a rolling cache: local_attn_size=3, sink=1, one-frame
        # chunks; call pattern = new chunk, same chunk again (overwrite), next chunks
        # until eviction happens twice (causal_model.py:194-236)
        mr = build_model(ns, shape, sd, dtype, local_attn_size=3, sink_size=1)
        sa = mr.blocks[0].self_attn
        grid = torch.tensor([[1, LAT_H // 2, LAT_W // 2]], dtype=torch.long)
        kv = {"k": torch.zeros(1, 3 * fs, shape.num_heads, 128, dtype=dtype),
              "v": torch.zeros(1, 3 * fs, shape.num_heads, 128, dtype=dtype),
              "global_end_index": torch.tensor([0]), "local_end_index": torch.tensor([0])}
        gs = torch.Generator().manual_seed(99)
        starts = [0, 0, 1, 2, 3, 3, 4]
        xs = [bf16_randn((1, fs, C), gs) for _ in starts]
        out["sa_starts"] = np.array(starts)
        out["sa_x"] = np.stack([f32(u) for u in xs])
        ys, les, ges = [], [], []
        for st, u in zip(starts, xs):
            y = sa(u.to(dtype), None, grid, mr.freqs, None, kv, st * fs, None)
            ys.append(f32(y))
            les.append(int(kv["local_end_index"]))
            ges.append(int(kv["global_end_index"]))
        out[f"sa_y_{tag}"] = np.stack(ys)
        out["sa_local_end"] = np.array(les)
        out["sa_global_end"] = np.array(ges)
        out[f"sa_k_final_{tag}"] = f32(kv["k"])
        out[f"sa_v_final_{tag}"] = f32(kv["v"])

        # ---- cross attention with cache (model.py:159-194)
        ca = m.blocks[1].cross_attn
        gx = torch.Generator().manual_seed(17)
        xq = bf16_randn((1, 2 * fs, C), gx)
        ctx = bf16_randn((1, 512, C), gx)
        cache = {"k": None, "v": None, "is_init": False}
        out["ca_x"] = f32(xq)
        out["ca_ctx"] = f32(ctx)
        out[f"ca_y_{tag}"] = f32(ca(xq.to(dtype), ctx.to(dtype), None, crossattn_cache=cache))
        out[f"ca_k_{tag}"] = f32(cache["k"][:, :64])
        out[f"ca_v_{tag}"] = f32(cache["v"][:, :64])
        out[f"ca_y2_{tag}"] = f32(ca(xq.to(dtype), ctx.to(dtype) * 0, None, crossattn_cache=cache))

        # ---- one full block with per-frame modulation (causal_model.py:284-336)
        blk = m.blocks[0]
        gb = torch.Generator().manual_seed(23)
        xb = bf16_randn((1, 2 * fs, C), gb)
        e0 = bf16_randn((1, 2, 6, C), gb, 0.3)
        kv = {"k": torch.zeros(1, 4 * fs, shape.num_heads, 128, dtype=dtype),
              "v": torch.zeros(1, 4 * fs, shape.num_heads, 128, dtype=dtype),
              "global_end_index": torch.tensor([0]), "local_end_index": torch.tensor([0])}
        cc = {"k": None, "v": None, "is_init": False}
        grid2 = torch.tensor([[2, LAT_H // 2, LAT_W // 2]], dtype=torch.long)
        yb = blk(xb.to(dtype), e0.to(dtype), torch.tensor([2 * fs]), grid2, m.freqs, ctx.to(dtype), None,
                 None, kv_cache=kv, crossattn_cache=cc, current_start=0)
        out["blk_x"] = f32(xb)
        out["blk_e0"] = f32(e0)
        out[f"blk_y_{tag}"] = f32(yb)

        # ---- full _forward_inference: two calls (second one appends to the cache)
        kvs, cas = fresh_caches(shape, 1, 5 * fs, dtype)
        gf = torch.Generator().manual_seed(31)
        pe = bf16_randn((1, 512, shape.text_dim), gf)
        pe[:, 77:] = 0  # zero padding rows as WanTextEncoder produces (wan_wrapper.py:50-51)
        x1 = bf16_randn((1, 16, 2, LAT_H, LAT_W), gf)
        x2 = bf16_randn((1, 16, 3, LAT_H, LAT_W), gf)
        t1 = torch.tensor([[937.5, 833.3333129882812]], dtype=torch.float32)
        t2 = torch.tensor([[500, 500, 250]], dtype=torch.int64)
        y1 = m(x1.to(dtype), t=t1, context=pe.to(dtype), seq_len=32760, kv_cache=kvs, crossattn_cache=cas,
               current_start=0, cache_start=None)
        y2 = m(x2.to(dtype), t=t2, context=pe.to(dtype), seq_len=32760, kv_cache=kvs, crossattn_cache=cas,
               current_start=2 * fs, cache_start=None)
        out["fwd_pe"] = f32(pe)
        out["fwd_x1"] = f32(x1)
        out["fwd_x2"] = f32(x2)
        out["fwd_t1"] = f32(t1)
        out["fwd_t2"] = t2.numpy()
        out[f"fwd_y1_{tag}"] = f32(y1)
        out[f"fwd_y2_{tag}"] = f32(y2)
        out[f"fwd_k0_{tag}"] = f32(kvs[0]["k"])
        out[f"fwd_v1_{tag}"] = f32(kvs[1]["v"])
        out[f"fwd_ck1_{tag}"] = f32(cas[1]["k"][:, :64])

    np.savez_compressed(os.path.join(GOLD, "modules_reduced.npz"), **out)
    print("modules_reduced.npz", len(out), "arrays")


# --------------------------------------------------------------------------------------
class _IdentityVAE:
    def decode_to_pixel(self, latents, use_cache=False):
        return latents


ROLLOUT_SCENARIOS = {
    # name: (nfpb, independent_first_frame, shift, n_noise_frames, n_initial, local_attn, sink)
    "nfpb1": (1, False, 5.0, 3, 0, -1, 0),
    "nfpb3": (3, False, 5.0, 6, 0, -1, 0),
    "iff": (3, True, 8.0, 4, 0, -1, 0),
    "ext": (3, False, 5.0, 3, 3, -1, 0),
    "i2v": (3, True, 5.0, 3, 1, -1, 0),
    "roll": (1, False, 5.0, 6, 0, 3, 1),
}


def run_reference_rollout(ns, shape, sd, dtype, name, noise, pe, eps_list, initial):
    nfpb, iff, shift, nfr, nin, las, sink = ROLLOUT_SCENARIOS[name]
    fs = (LAT_H // 2) * (LAT_W // 2)
    model = build_model(ns, shape, sd, dtype, local_attn_size=las, sink_size=sink)
    wrapper = ref_shim.build_wrapper(ns, model, shift)
    args = types.SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                                 independent_first_frame=iff, num_frame_per_block=nfpb, context_noise=0,
                                 model_kwargs={})
    sink_out = io.StringIO()
    with contextlib.redirect_stdout(sink_out):
        pipe = ns.CausalInferencePipeline(args, device="cpu", generator=wrapper,
                                          text_encoder=lambda text_prompts: {"prompt_embeds": pe.to(dtype)},
                                          vae=_IdentityVAE())
    # the reference hard-codes the 1.3B/480p constants (causal_inference.py:33-34, :288-293)
    pipe.num_transformer_blocks = shape.num_layers
    pipe.frame_seq_length = fs
    cache_tokens = (las if las != -1 else nfr + nin) * fs
    pipe.kv_cache1, pipe.crossattn_cache = fresh_caches(shape, noise.shape[0], cache_tokens, dtype)
    queue = list(eps_list)
    orig = torch.randn_like

    def injected(t, *a, **kw):
        return queue.pop(0).to(t.dtype).reshape(t.shape)

    torch.randn_like = injected
    try:
        with contextlib.redirect_stdout(sink_out), torch.no_grad():
            _, lat = pipe.inference(noise.to(dtype), ["p"] * noise.shape[0],
                                    initial_latent=None if initial is None else initial.to(dtype),
                                    return_latents=True)
    finally:
        torch.randn_like = orig
    assert not queue, f"{len(queue)} unused eps tensors"
    return lat, pipe.kv_cache1


def gen_rollouts(ns):
    shape = sfa.WAN_REDUCED
    sd = sfa.synth_state_dict(shape, seed=0)
    out = {"weights_seed": np.array(0)}
    for si, name in enumerate(ROLLOUT_SCENARIOS):
        nfpb, iff, shift, nfr, nin, las, sink = ROLLOUT_SCENARIOS[name]
        g = torch.Generator().manual_seed(1000 + si)
        B = 1
        noise = bf16_randn((B, nfr, 16, LAT_H, LAT_W), g)
        pe = bf16_randn((B, 512, shape.text_dim), g)
        pe[:, 40 + 10 * si:] = 0
        initial = bf16_randn((B, nin, 16, LAT_H, LAT_W), g) if nin else None
        chunks = [nfpb] * ((nfr - 1) // nfpb if (iff and not nin) else nfr // nfpb)
        if iff and not nin:
            chunks = [1] + chunks
        eps = [bf16_randn((B * f, 16, LAT_H, LAT_W), g) for f in chunks for _ in range(3)]
        out[f"{name}_noise"] = f32(noise)
        out[f"{name}_pe"] = f32(pe)
        if initial is not None:
            out[f"{name}_initial"] = f32(initial)
        for j, e in enumerate(eps):
            out[f"{name}_eps{j}"] = f32(e)
        out[f"{name}_neps"] = np.array(len(eps))
        for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            lat, kv = run_reference_rollout(ns, shape, sd, dtype, name, noise, pe, eps, initial)
            out[f"{name}_lat_{tag}"] = f32(lat)
            out[f"{name}_local_end"] = np.array(int(kv[0]["local_end_index"]))
            out[f"{name}_global_end"] = np.array(int(kv[0]["global_end_index"]))
            print(name, tag, "latents rms %.4f" % lat.float().pow(2).mean().sqrt().item())
    np.savez_compressed(os.path.join(GOLD, "rollouts_reduced.npz"), **out)
    print("rollouts_reduced.npz", len(out), "arrays")


# --------------------------------------------------------------------------------------
def gen_full(ns):
    """One full-shape (Wan-1.3B, 60x104 latent) single-frame forward through the
    reference wrapper: bf16 as shipped, and the fp32 math variant."""
    shape = sfa.WAN_1_3B
    sd = sfa.synth_state_dict(shape, seed=0)
    out = {"weights_seed": np.array(0), "input_seed": np.array(77)}
    g = torch.Generator().manual_seed(77)
    noisy = bf16_randn((1, 1, 16, 60, 104), g)
    pe = bf16_randn((1, 512, shape.text_dim), g)
    pe[:, 120:] = 0
    ts = torch.tensor([[937.5]], dtype=torch.float32)
    out["noisy"] = f32(noisy)
    out["pe_checksum"] = np.array(pe.double().sum().item())
    out["pe_abs_checksum"] = np.array(pe.double().abs().sum().item())
    for tag, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        model = build_model(ns, shape, sd, dtype)
        wrapper = ref_shim.build_wrapper(ns, model, 5.0)
        kv, ca = fresh_caches(shape, 1, 1560, dtype)
        with torch.no_grad():
            flow, x0 = wrapper(noisy.to(dtype), {"prompt_embeds": pe.to(dtype)}, ts, kv_cache=kv,
                               crossattn_cache=ca, current_start=0)
        out[f"flow_{tag}"] = f32(flow)
        out[f"x0_{tag}"] = f32(x0)
        out[f"k0_head0_{tag}"] = f32(kv[0]["k"][0, :, 0])
        out[f"k29_head5_{tag}"] = f32(kv[29]["k"][0, :, 5])
        print("full", tag, "flow rms %.4f" % flow.float().pow(2).mean().sqrt().item())
        del model, wrapper
    np.savez_compressed(os.path.join(GOLD, "full_1p3b.npz"), **out)
    print("full_1p3b.npz", len(out), "arrays")


def gen_tconfig(ns):
    """BASELINE configs[0] ("T"): the reference's CausalInferencePipeline at the FULL Wan-1.3B shape,
    configs/tiny_test.yaml + the few-step keys (steps [1000, 750, 500, 250] warped, independent_first_frame,
    1 frame per block, wrapper-default timestep_shift 8.0), noise [1, 2, 16, 60, 104] => chunks [1, 1] = 2 x (4 + 1)
    forwards of 1560 tokens; one prompt.  Re-noise tensors injected (pre-drawn), bf16 as shipped and the fp32 math
    variant.  Stored: inputs' seed, the latents of both variants, every 8th KV row of one head of the first / last layer."""
    import time
    shape = sfa.WAN_1_3B
    sd = sfa.synth_state_dict(shape, seed=0)
    H, W = 60, 104
    fs = (H // 2) * (W // 2)
    g = torch.Generator().manual_seed(4242)
    noise = bf16_randn((1, 2, 16, H, W), g)
    pe = bf16_randn((1, 512, shape.text_dim), g)
    pe[:, 93:] = 0
    eps = [bf16_randn((1, 16, H, W), g) for _ in range(6)]
    out = {"weights_seed": np.array(0), "input_seed": np.array(4242), "noise_checksum": np.array(noise.double().sum().item()),
           "pe_checksum": np.array(pe.double().sum().item()), "eps_checksum": np.array(sum(e.double().sum().item() for e in eps))}
    args = types.SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                                 independent_first_frame=True, num_frame_per_block=1, context_noise=0, model_kwargs={})
    for tag, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        t0 = time.time()
        model = build_model(ns, shape, sd, dtype)
        wrapper = ref_shim.build_wrapper(ns, model, 8.0)
        sink_out = io.StringIO()
        with contextlib.redirect_stdout(sink_out):
            pipe = ns.CausalInferencePipeline(args, device="cpu", generator=wrapper,
                                              text_encoder=lambda text_prompts: {"prompt_embeds": pe.to(dtype)},
                                              vae=_IdentityVAE())
        pipe.kv_cache1, pipe.crossattn_cache = fresh_caches(shape, 1, 2 * fs, dtype)
        queue = list(eps)
        orig = torch.randn_like
        torch.randn_like = lambda t, *a, **kw: queue.pop(0).to(t.dtype).reshape(t.shape)
        try:
            with contextlib.redirect_stdout(sink_out), torch.no_grad():
                _, lat = pipe.inference(noise.to(dtype), ["p"], return_latents=True)
        finally:
            torch.randn_like = orig
        assert not queue
        out[f"lat_{tag}"] = f32(lat)
        out[f"k0_head3_{tag}"] = f32(pipe.kv_cache1[0]["k"][0, ::8, 3])      # every 8th cache row
        out[f"v29_head7_{tag}"] = f32(pipe.kv_cache1[29]["v"][0, ::8, 7])
        print("tconfig", tag, "latents rms %.4f" % lat.float().pow(2).mean().sqrt().item(), "%.0f s" % (time.time() - t0))
        del model, wrapper, pipe
    d = np.linalg.norm(out["lat_bf16"] - out["lat_f32"]) / np.linalg.norm(out["lat_f32"])
    print("tconfig: reference bf16 vs fp32 rel err %.4f" % d)
    np.savez_compressed(os.path.join(GOLD, "tconfig_1p3b.npz"), **out)
    print("tconfig_1p3b.npz", len(out), "arrays")


def gen_s1(ns):
    """BASELINE configs[1] ("S1"), first two chunks at the FULL Wan-1.3B shape: self_forcing_dmd settings
    (steps [1000, 750, 500, 250] warped with shift 5.0, 3 frames per block, context_noise 0), noise
    [1, 6, 16, 60, 104] => 2 chunks x (4 + 1) forwards of 4680 tokens against caches of 4680 / 9360 tokens -- the shapes
    at which the rollout's large-tile kernels run.  fp32 math variant of the reference (and its bf16 run for the
    distance between the two).  Stored: latent frames 0, 2, 3 and 5 in full, per-frame sums of all six."""
    import time
    shape = sfa.WAN_1_3B
    sd = sfa.synth_state_dict(shape, seed=0)
    H, W = 60, 104
    fs = (H // 2) * (W // 2)
    g = torch.Generator().manual_seed(5151)
    noise = bf16_randn((1, 6, 16, H, W), g)
    pe = bf16_randn((1, 512, shape.text_dim), g)
    pe[:, 141:] = 0
    eps = [bf16_randn((3, 16, H, W), g) for _ in range(6)]
    out = {"weights_seed": np.array(0), "input_seed": np.array(5151), "noise_checksum": np.array(noise.double().sum().item()),
           "pe_checksum": np.array(pe.double().sum().item()), "eps_checksum": np.array(sum(e.double().sum().item() for e in eps))}
    args = types.SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                                 independent_first_frame=False, num_frame_per_block=3, context_noise=0, model_kwargs={})
    lats = {}
    for tag, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        t0 = time.time()
        model = build_model(ns, shape, sd, dtype)
        wrapper = ref_shim.build_wrapper(ns, model, 5.0)
        sink_out = io.StringIO()
        with contextlib.redirect_stdout(sink_out):
            pipe = ns.CausalInferencePipeline(args, device="cpu", generator=wrapper,
                                              text_encoder=lambda text_prompts: {"prompt_embeds": pe.to(dtype)},
                                              vae=_IdentityVAE())
        pipe.kv_cache1, pipe.crossattn_cache = fresh_caches(shape, 1, 6 * fs, dtype)
        queue = list(eps)
        orig = torch.randn_like
        torch.randn_like = lambda t, *a, **kw: queue.pop(0).to(t.dtype).reshape(t.shape)
        try:
            with contextlib.redirect_stdout(sink_out), torch.no_grad():
                _, lat = pipe.inference(noise.to(dtype), ["p"], return_latents=True)
        finally:
            torch.randn_like = orig
        assert not queue
        lats[tag] = lat.float()
        print("s1", tag, "latents rms %.4f" % lat.float().pow(2).mean().sqrt().item(), "%.0f s" % (time.time() - t0), flush=True)
        del model, wrapper, pipe
    lf = lats["f32"]
    out["frames"] = np.array([0, 2, 3, 5])
    out["lat_f32_frames"] = f32(lf[:, [0, 2, 3, 5]])
    out["lat_f32_frame_sums"] = lf.double().sum(dim=(0, 2, 3, 4)).numpy()
    out["lat_f32_frame_abs_sums"] = lf.double().abs().sum(dim=(0, 2, 3, 4)).numpy()
    out["ref_bf16_vs_f32"] = np.array(((lats["bf16"] - lf).norm() / lf.norm()).item())
    print("s1: reference bf16 vs fp32 rel err %.4f" % float(out["ref_bf16_vs_f32"]))
    np.savez_compressed(os.path.join(GOLD, "s1_2chunks_1p3b.npz"), **out)
    print("s1_2chunks_1p3b.npz", len(out), "arrays")


def gen_s1_full(ns):
    """BASELINE configs[1] ("S1") IN FULL at the Wan-1.3B shape: 21 latent frames = 7 chunks x (4 + 1) forwards of 4680
    tokens against caches of 4680 ... 32760 tokens -- every cache length the benchmark times, incl. the ones where 57 %
    of its FLOPs run.  Same settings as gen_s1; fp32 math variant of the reference plus its own bf16 run (the noise
    floor the tolerance is stated against).  Stored: seven latent frames in full (one per chunk), per-frame sums and
    absolute sums of all 21, the reference's per-frame bf16-vs-fp32 distance, and every 16th row of one head of the
    first / last layer's K / V cache (rows of all seven chunks)."""
    import time
    shape = sfa.WAN_1_3B
    sd = sfa.synth_state_dict(shape, seed=0)
    H, W, F = 60, 104, 21
    fs = (H // 2) * (W // 2)
    g = torch.Generator().manual_seed(6161)
    noise = bf16_randn((1, F, 16, H, W), g)
    pe = bf16_randn((1, 512, shape.text_dim), g)
    pe[:, 117:] = 0
    eps = [bf16_randn((3, 16, H, W), g) for _ in range(21)]
    out = {"weights_seed": np.array(0), "input_seed": np.array(6161), "noise_checksum": np.array(noise.double().sum().item()),
           "pe_checksum": np.array(pe.double().sum().item()), "eps_checksum": np.array(sum(e.double().sum().item() for e in eps))}
    args = types.SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                                 independent_first_frame=False, num_frame_per_block=3, context_noise=0, model_kwargs={})
    lats = {}
    for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        t0 = time.time()
        model = build_model(ns, shape, sd, dtype)
        wrapper = ref_shim.build_wrapper(ns, model, 5.0)
        sink_out = io.StringIO()
        with contextlib.redirect_stdout(sink_out):
            pipe = ns.CausalInferencePipeline(args, device="cpu", generator=wrapper,
                                              text_encoder=lambda text_prompts: {"prompt_embeds": pe.to(dtype)},
                                              vae=_IdentityVAE())
        pipe.kv_cache1, pipe.crossattn_cache = fresh_caches(shape, 1, F * fs, dtype)
        queue = list(eps)
        orig = torch.randn_like
        torch.randn_like = lambda t, *a, **kw: queue.pop(0).to(t.dtype).reshape(t.shape)
        try:
            with contextlib.redirect_stdout(sink_out), torch.no_grad():
                _, lat = pipe.inference(noise.to(dtype), ["p"], return_latents=True)
        finally:
            torch.randn_like = orig
        assert not queue
        lats[tag] = lat.float()
        if tag == "f32":
            out["k0_head5_f32"] = f32(pipe.kv_cache1[0]["k"][0, ::16, 5])
            out["v29_head11_f32"] = f32(pipe.kv_cache1[29]["v"][0, ::16, 11])
        print("s1full", tag, "latents rms %.4f" % lat.float().pow(2).mean().sqrt().item(), "%.0f s" % (time.time() - t0), flush=True)
        del model, wrapper, pipe
    lf, lb = lats["f32"], lats["bf16"]
    frames = [0, 4, 8, 11, 14, 17, 20]
    out["frames"] = np.array(frames)
    out["lat_f32_frames"] = f32(lf[:, frames])
    out["lat_f32_frame_sums"] = lf.double().sum(dim=(0, 2, 3, 4)).numpy()
    out["lat_f32_frame_abs_sums"] = lf.double().abs().sum(dim=(0, 2, 3, 4)).numpy()
    out["lat_f32_frame_norms"] = lf.double().pow(2).sum(dim=(0, 2, 3, 4)).sqrt().numpy()
    out["ref_bf16_vs_f32_per_frame"] = ((lb - lf).double().pow(2).sum(dim=(0, 2, 3, 4)).sqrt()
                                        / lf.double().pow(2).sum(dim=(0, 2, 3, 4)).sqrt()).numpy()
    out["ref_bf16_vs_f32"] = np.array(((lb - lf).norm() / lf.norm()).item())
    print("s1full: reference bf16 vs fp32 rel err %.4f; per frame" % float(out["ref_bf16_vs_f32"]),
          np.array2string(out["ref_bf16_vs_f32_per_frame"], precision=4))
    np.savez_compressed(os.path.join(GOLD, "s1_full_clip_1p3b.npz"), **out)
    print("s1_full_clip_1p3b.npz", len(out), "arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    ns = ref_shim.load()
    todo = a.only.split(",") if a.only else ["ops", "modules", "rollouts"] + (["full"] if a.full else [])
    with torch.no_grad():
        if "ops" in todo:
            gen_ops(ns)
        if "modules" in todo:
            gen_modules(ns)
        if "rollouts" in todo:
            gen_rollouts(ns)
        if "full" in todo:
            gen_full(ns)
        if "tconfig" in todo:
            gen_tconfig(ns)
        if "s1" in todo:
            gen_s1(ns)
        if "s1full" in todo:
            gen_s1_full(ns)


if __name__ == "__main__":
    main()
