"""GPU parity tests of the whole path: `WanDiffusionWrapper.forward` and
`CausalInferencePipeline.inference` (HIP, through the C-ABI) against the golden vectors the
reference itself produced (tests/golden/, oracle/make_golden.py) and against the CPU oracle.

Tolerance contract: bf16-MFMA / fp32-accumulate build vs the fp32 math reference
<= 2e-2 relative Frobenius per forward and per rollout (the reference's own bf16 path sits at
1.3e-2 per full-size forward and 3.5e-3 per reduced rollout from the same fp32 math)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import self_forcing_amd as sfa
from oracle import wan_oracle as wo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
LAT_H, LAT_W = 8, 12
FS = (LAT_H // 2) * (LAT_W // 2)
TOL = 2e-2


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


@pytest.fixture(scope="module")
def sd_reduced():
    return sfa.synth_state_dict(sfa.WAN_REDUCED, seed=0)


def make_pipe(sd, nfpb, iff, shift, las=-1, sink=0, pe=None):
    args = SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                           independent_first_frame=iff, num_frame_per_block=nfpb, context_noise=0)
    gen = sfa.WanDiffusionWrapper(shape=sfa.WAN_REDUCED, state_dict=sd, timestep_shift=shift, is_causal=True,
                                  local_attn_size=las, sink_size=sink, device=DEV)
    return sfa.CausalInferencePipeline(args, DEV, generator=gen, text_encoder=sfa.FixedTextEncoder(pe), vae=sfa.IdentityVAE())


def test_forward_two_calls_vs_reference_golden(sd_reduced):
    mods = np.load(os.path.join(GOLD, "modules_reduced.npz"))
    pipe = make_pipe(sd_reduced, 1, False, 5.0)
    pipe.frame_seq_length = FS
    pipe._initialize_kv_cache(1, torch.bfloat16, DEV, cache_tokens=5 * FS)
    pipe._initialize_crossattn_cache(1, torch.bfloat16, DEV)
    gen = pipe.generator
    cond = {"prompt_embeds": T(mods["fwd_pe"]).bfloat16().to(DEV)}
    x1 = T(mods["fwd_x1"]).bfloat16().permute(0, 2, 1, 3, 4).contiguous().to(DEV)   # -> [B, F, C, H, W]
    x2 = T(mods["fwd_x2"]).bfloat16().permute(0, 2, 1, 3, 4).contiguous().to(DEV)
    t1, t2 = T(mods["fwd_t1"]).to(DEV), torch.from_numpy(mods["fwd_t2"]).to(DEV)
    f1, x01 = gen(x1, cond, t1, pipe.kv_cache1, pipe.crossattn_cache, 0)
    f2, x02 = gen(x2, cond, t2, pipe.kv_cache1, pipe.crossattn_cache, 2 * FS)
    torch.cuda.synchronize()
    y1 = T(mods["fwd_y1_f32"]).permute(0, 2, 1, 3, 4)
    y2 = T(mods["fwd_y2_f32"]).permute(0, 2, 1, 3, 4)
    assert rel(f1, y1) < TOL and rel(f2, y2) < TOL
    assert rel(pipe.kv_cache1[0]["k"], T(mods["fwd_k0_f32"])) < TOL
    assert rel(pipe.kv_cache1[1]["v"], T(mods["fwd_v1_f32"])) < TOL
    n = mods["fwd_ck1_f32"].shape[1]
    assert rel(pipe.crossattn_cache[1]["k"][:, :n], T(mods["fwd_ck1_f32"])) < TOL
    assert all(c["is_init"] for c in pipe.crossattn_cache)
    assert int(pipe.kv_cache1[1]["global_end_index"]) == 5 * FS and int(pipe.kv_cache1[1]["local_end_index"]) == 5 * FS
    # x0 = xt - sigma * flow with the flow the kernel itself produced (fp64, bit exact)
    sched = wo.FlowMatchTables(5.0)
    ref_x0 = wo.flow_to_x0(sched, f2[0].cpu(), x2[0].cpu(), t2[0].cpu())
    assert torch.equal(x02[0].cpu(), ref_x0)


SCEN = {  # name: (nfpb, independent_first_frame, shift, local_attn, sink) -- oracle/make_golden.py
    "nfpb1": (1, False, 5.0, -1, 0), "nfpb3": (3, False, 5.0, -1, 0), "iff": (3, True, 8.0, -1, 0),
    "ext": (3, False, 5.0, -1, 0), "i2v": (3, True, 5.0, -1, 0), "roll": (1, False, 5.0, 3, 1),
}


@pytest.mark.parametrize("name", list(SCEN))
def test_rollout_vs_reference_golden(sd_reduced, name):
    R = np.load(os.path.join(GOLD, "rollouts_reduced.npz"))
    nfpb, iff, shift, las, sink = SCEN[name]
    pe = T(R[f"{name}_pe"]).bfloat16().to(DEV)
    pipe = make_pipe(sd_reduced, nfpb, iff, shift, las, sink, pe)
    eps = [T(R[f"{name}_eps{j}"]).bfloat16() for j in range(int(R[f"{name}_neps"]))]
    queue = list(eps)
    pipe.noise_source = lambda t: queue.pop(0).reshape(t.shape)
    initial = T(R[f"{name}_initial"]).bfloat16().to(DEV) if f"{name}_initial" in R else None
    noise = T(R[f"{name}_noise"]).bfloat16().to(DEV)
    video, lat = pipe.inference(noise, ["p"], initial_latent=initial, return_latents=True)
    torch.cuda.synchronize()
    assert not queue
    assert rel(lat, T(R[f"{name}_lat_f32"])) < TOL          # vs the reference run in fp32
    assert rel(lat, T(R[f"{name}_lat_bf16"])) < TOL         # vs the reference as shipped (bf16)
    assert int(pipe.kv_cache1[0]["local_end_index"]) == int(R[f"{name}_local_end"])
    assert int(pipe.kv_cache1[0]["global_end_index"]) == int(R[f"{name}_global_end"])
    assert torch.equal(video, (lat * 0.5 + 0.5).clamp(0, 1))
    if initial is not None:   # initial frames are returned verbatim
        assert torch.equal(lat[:, :initial.shape[1]], initial)

    # a second call on the same pipeline takes the reset branch and reproduces the first bit-exactly
    queue.extend(eps)
    _, lat2 = pipe.inference(noise, ["p"], initial_latent=initial, return_latents=True)
    assert torch.equal(lat, lat2)


def test_rollout_batch2_matches_per_sample(sd_reduced):
    """Batch > 1: each sample must equal its own batch-1 rollout (independent caches)."""
    g = torch.Generator().manual_seed(77)
    B, F = 2, 3
    noise = torch.randn(B, F, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16)
    pe = torch.randn(B, 512, sfa.WAN_REDUCED.text_dim, generator=g).to(torch.bfloat16)
    eps = [torch.randn(B, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(9)]
    outs = []
    for sel in (slice(0, 2), slice(0, 1), slice(1, 2)):
        pipe = make_pipe(sd_reduced, 1, False, 5.0, pe=pe[sel].to(DEV))
        q = [e[sel] for e in eps]
        pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
        outs.append(pipe.inference(noise[sel].to(DEV), ["p"] * (sel.stop - sel.start), return_latents=True)[1])
    assert rel(outs[0][0:1], outs[1]) < 1e-6 and rel(outs[0][1:2], outs[2]) < 1e-6


def test_global_cache_overflow_raises(sd_reduced):
    """Running one frame past the capacity in global mode is an error (the reference fails with a
    slice-shape RuntimeError, causal_model.py:228)."""
    pe = torch.zeros(1, 512, sfa.WAN_REDUCED.text_dim, dtype=torch.bfloat16, device=DEV)
    pipe = make_pipe(sd_reduced, 1, False, 5.0, pe=pe)
    pipe.frame_seq_length = FS
    pipe._initialize_kv_cache(1, torch.bfloat16, DEV, cache_tokens=2 * FS)
    pipe._initialize_crossattn_cache(1, torch.bfloat16, DEV)
    x = torch.zeros(1, 1, 16, LAT_H, LAT_W, dtype=torch.bfloat16, device=DEV)
    t = torch.zeros(1, 1, dtype=torch.int64, device=DEV)
    cond = {"prompt_embeds": pe}
    for fr in range(2):
        pipe.generator(x, cond, t, pipe.kv_cache1, pipe.crossattn_cache, fr * FS)
    with pytest.raises(RuntimeError, match="overflow"):
        pipe.generator(x, cond, t, pipe.kv_cache1, pipe.crossattn_cache, 2 * FS)


def test_foreign_cache_dicts_and_rebound_indices(sd_reduced):
    """Caches built the reference's way (plain per-layer index tensors, reset by REBINDING them,
    causal_inference.py:128-132) must work: the wrapper falls back to reading the device indices."""
    shape = sfa.WAN_REDUCED
    gen = sfa.WanDiffusionWrapper(shape=shape, state_dict=sd_reduced, timestep_shift=5.0, is_causal=True, device=DEV)
    kv = [{"k": torch.zeros(1, 3 * FS, shape.num_heads, 128, dtype=torch.bfloat16, device=DEV),
           "v": torch.zeros(1, 3 * FS, shape.num_heads, 128, dtype=torch.bfloat16, device=DEV),
           "global_end_index": torch.tensor([0], dtype=torch.long, device=DEV),
           "local_end_index": torch.tensor([0], dtype=torch.long, device=DEV)} for _ in range(shape.num_layers)]
    ca = [{"k": torch.zeros(1, 512, shape.num_heads, 128, dtype=torch.bfloat16, device=DEV),
           "v": torch.zeros(1, 512, shape.num_heads, 128, dtype=torch.bfloat16, device=DEV), "is_init": False}
          for _ in range(shape.num_layers)]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 1, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16).to(DEV)
    cond = {"prompt_embeds": torch.randn(1, 512, shape.text_dim, generator=g).to(torch.bfloat16).to(DEV)}
    t = torch.full((1, 1), 500, dtype=torch.int64, device=DEV)
    a0, _ = gen(x, cond, t, kv, ca, 0)
    a1, _ = gen(x, cond, t, kv, ca, FS)
    assert int(kv[1]["local_end_index"]) == 2 * FS
    for d in kv:   # reset exactly as the reference does
        d["global_end_index"] = torch.tensor([0], dtype=torch.long, device=DEV)
        d["local_end_index"] = torch.tensor([0], dtype=torch.long, device=DEV)
    b0, _ = gen(x, cond, t, kv, ca, 0)
    assert torch.equal(a0, b0)


def test_full_1p3b_forward_vs_reference_golden():
    """Full Wan-1.3B shape, one 60x104 latent frame (1560 tokens): flow / x0 / cached keys against
    the reference's own outputs (bf16 as shipped and fp32 math)."""
    path = os.path.join(GOLD, "full_1p3b.npz")
    Gd = np.load(path)
    shape = sfa.WAN_1_3B
    sd = sfa.synth_state_dict(shape, seed=int(Gd["weights_seed"]))
    g = torch.Generator().manual_seed(int(Gd["input_seed"]))
    noisy = torch.randn(1, 1, 16, 60, 104, generator=g).to(torch.bfloat16)
    pe = torch.randn(1, 512, shape.text_dim, generator=g).to(torch.bfloat16)
    pe[:, 120:] = 0
    assert torch.equal(noisy.float(), T(Gd["noisy"])), "torch CPU generator stream changed; regenerate the fixture"
    assert abs(pe.double().sum().item() - float(Gd["pe_checksum"])) < 1e-6
    args = SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                           independent_first_frame=False, num_frame_per_block=1, context_noise=0)
    gen = sfa.WanDiffusionWrapper(shape=shape, state_dict=sd, timestep_shift=5.0, is_causal=True, device=DEV)
    del sd
    pipe = sfa.CausalInferencePipeline(args, DEV, generator=gen, text_encoder=sfa.FixedTextEncoder(pe.to(DEV)), vae=sfa.IdentityVAE())
    pipe.frame_seq_length = 1560
    pipe._initialize_kv_cache(1, torch.bfloat16, DEV, cache_tokens=1560)
    pipe._initialize_crossattn_cache(1, torch.bfloat16, DEV)
    ts = torch.tensor([[937.5]], dtype=torch.float32, device=DEV)
    flow, x0 = gen(noisy.to(DEV), {"prompt_embeds": pe.to(DEV)}, ts, pipe.kv_cache1, pipe.crossattn_cache, 0)
    torch.cuda.synchronize()
    e_flow = rel(flow, T(Gd["flow_f32"]))
    e_x0 = rel(x0, T(Gd["x0_f32"]))
    e_ref = rel(T(Gd["flow_bf16"]), T(Gd["flow_f32"]))
    print(f"full-shape forward: flow err {e_flow:.3e} x0 err {e_x0:.3e} (reference bf16 vs fp32: {e_ref:.3e})")
    assert e_flow < TOL and e_x0 < TOL
    assert rel(pipe.kv_cache1[0]["k"][0, :, 0], T(Gd["k0_head0_f32"])) < TOL
    assert rel(pipe.kv_cache1[29]["k"][0, :, 5], T(Gd["k29_head5_f32"])) < TOL


def test_cache_only_pass_leaves_identical_caches(sd_reduced):
    """The context pass may skip what nothing reads (sf_forward_args.cache_only): the KV caches
    after it must equal those of a full pass bit for bit."""
    shape = sfa.WAN_REDUCED
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 2, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16).to(DEV)
    pe = torch.randn(1, 512, shape.text_dim, generator=g).to(torch.bfloat16).to(DEV)
    t = torch.zeros(1, 2, dtype=torch.int64, device=DEV)
    caches = []
    for cache_only in (False, True):
        pipe = make_pipe(sd_reduced, 1, False, 5.0, pe=pe)
        pipe.frame_seq_length = FS
        pipe._initialize_kv_cache(1, torch.bfloat16, DEV, cache_tokens=2 * FS)
        pipe._initialize_crossattn_cache(1, torch.bfloat16, DEV)
        out = pipe.generator(x, {"prompt_embeds": pe}, t, pipe.kv_cache1, pipe.crossattn_cache, 0, cache_only=cache_only)
        assert (out == (None, None)) == cache_only
        caches.append(pipe.kv_cache1)
    for a, b in zip(*caches):
        assert torch.equal(a["k"], b["k"]) and torch.equal(a["v"], b["v"])
        assert int(a["local_end_index"]) == int(b["local_end_index"]) == 2 * FS


def test_two_concurrent_streams_match_sequential(sd_reduced):
    """RolloutPool: two rollouts in flight on two HIP streams (shared weights, own caches and
    workspaces) give the same latents as running them one after the other."""
    shape = sfa.WAN_REDUCED
    g = torch.Generator().manual_seed(13)
    jobs = []
    for j in range(4):
        noise = torch.randn(1, 3, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16)
        eps = [torch.randn(1, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(9)]
        jobs.append((f"prompt {j}", noise, eps))
    args = SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                           independent_first_frame=False, num_frame_per_block=1, context_noise=0)
    gen = sfa.WanDiffusionWrapper(shape=shape, state_dict=sd_reduced, timestep_shift=5.0, is_causal=True, device=DEV)
    enc = sfa.SyntheticTextEncoder(shape.text_len, shape.text_dim, device=DEV)

    def roll(pipe, job):
        prompt, noise, eps = job
        q = list(eps)
        pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
        return pipe.inference(noise.to(DEV), [prompt], return_latents=True)[1].clone()

    seq = sfa.RolloutPool(args, DEV, gen, lambda: enc, sfa.IdentityVAE, streams=1).run(jobs, roll)
    par = sfa.RolloutPool(args, DEV, gen, lambda: enc, sfa.IdentityVAE, streams=2).run(jobs, roll)
    for a, b in zip(seq, par):
        assert torch.equal(a, b)


def test_streaming_matches_batch_inference(sd_reduced):
    """`stream()` (chunk-at-a-time, last context pass skipped as demo.py:396 does) yields exactly the
    latents `inference()` returns."""
    g = torch.Generator().manual_seed(31)
    noise = torch.randn(1, 6, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16).to(DEV)
    pe = torch.randn(1, 512, sfa.WAN_REDUCED.text_dim, generator=g).to(torch.bfloat16).to(DEV)
    eps = [torch.randn(3, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(6)]
    pipe = make_pipe(sd_reduced, 3, False, 5.0, pe=pe)
    q = list(eps)
    pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
    _, lat = pipe.inference(noise, ["p"], return_latents=True)
    q.extend(eps)
    chunks = list(pipe.stream(noise, ["p"]))
    assert [c[0] for c in chunks] == [0, 1] and all(c[1].shape[1] == 3 for c in chunks)
    assert torch.equal(torch.cat([c[1] for c in chunks], dim=1), lat)
    assert torch.equal(chunks[1][2], (chunks[1][1] * 0.5 + 0.5).clamp(0, 1))
    # the skipped pass would only have rewritten the last chunk's K/V: indices still cover all tokens
    assert int(pipe.kv_cache1[0]["global_end_index"]) == 6 * FS


def test_streaming_overlapped_decode_matches_serial(sd_reduced):
    """stream(overlap_decode=True): the VAE decode of chunk k runs on a second HIP stream under the denoising
    of chunk k+1 and is yielded one chunk later -- same latents, same pixels, bit for bit."""
    from self_forcing_amd import vae_weights as vw
    g = torch.Generator().manual_seed(41)
    noise = torch.randn(1, 6, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16).to(DEV)
    pe = torch.randn(1, 512, sfa.WAN_REDUCED.text_dim, generator=g).to(torch.bfloat16).to(DEV)
    eps = [torch.randn(2, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(9)]
    pipe = make_pipe(sd_reduced, 2, False, 5.0, pe=pe)
    pipe.vae = sfa.WanVAEWrapper(vw.synth_vae_state_dict(vw.VAE_REDUCED, seed=0), device=DEV, shape=vw.VAE_REDUCED)
    q = list(eps)
    pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
    serial = [(i, x.clone(), p.clone()) for i, x, p in pipe.stream(noise, ["p"])]
    q.extend(eps)
    over = [(i, x.clone(), p.clone()) for i, x, p in pipe.stream(noise, ["p"], overlap_decode=True)]
    assert [c[0] for c in over] == [0, 1, 2]
    assert serial[0][2].shape == (1, 5, 3, 8 * LAT_H, 8 * LAT_W) and serial[1][2].shape[1] == 8
    for a, b in zip(serial, over):
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_add_condition_pose_tokens_vs_oracle():
    """x += pose_proj(add_condition) after the patch embedding (the intent of causal_model.py:786-819).
    Parity for this branch is pinned by the oracle restatement only: the reference's own inference
    branch raises on it (see oracle/wan_oracle.py)."""
    shape = sfa.WAN_REDUCED
    sd = sfa.synth_state_dict(shape, seed=0, pose=True)
    assert torch.equal(sd["head.head.weight"], sfa.synth_state_dict(shape, seed=0)["head.head.weight"])
    g = torch.Generator().manual_seed(55)
    pe = torch.randn(1, 512, shape.text_dim, generator=g).to(torch.bfloat16)
    x = torch.randn(1, 2, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16)
    cond = (0.5 * torch.randn(1, 2 * FS, 5120, generator=g)).to(torch.bfloat16)
    t = torch.tensor([[833.3333129882812, 625.0]])
    gen = sfa.WanDiffusionWrapper(shape=shape, state_dict=sd, timestep_shift=5.0, is_causal=True, device=DEV)
    args = SimpleNamespace(denoising_step_list=[1000], warp_denoising_step=False, independent_first_frame=False,
                           num_frame_per_block=1, context_noise=0)
    pipe = sfa.CausalInferencePipeline(args, DEV, generator=gen, text_encoder=sfa.FixedTextEncoder(pe.to(DEV)), vae=sfa.IdentityVAE())
    pipe.frame_seq_length = FS
    outs = []
    for c in (cond, None):
        pipe._initialize_kv_cache(1, torch.bfloat16, DEV, cache_tokens=2 * FS)
        pipe._initialize_crossattn_cache(1, torch.bfloat16, DEV)
        cd = {"prompt_embeds": pe.to(DEV)}
        if c is not None:
            cd["add_condition"] = c.to(DEV)          # the reference passes it through the conditional dict
        outs.append(gen(x.to(DEV), cd, t.to(DEV), pipe.kv_cache1, pipe.crossattn_cache, 0)[0])
    Wf = wo.prepare_weights(sd, torch.float32)
    cfg = wo.OracleConfig(dim=shape.dim, ffn_dim=shape.ffn_dim, num_heads=shape.num_heads, num_layers=shape.num_layers,
                          text_dim=shape.text_dim)
    for out, c in zip(outs, (cond, None)):
        kv, ca = wo.init_kv_cache(cfg, 1, 2 * FS, torch.float32), wo.init_crossattn_cache(cfg, 1, torch.float32)
        ref = wo.wrapper_forward(Wf, cfg, wo.FlowMatchTables(5.0), x.float(), pe.float(), t, kv, ca, 0,
                                 add_condition=None if c is None else c.float())[0]
        assert rel(out, ref) < TOL
    assert rel(outs[0], outs[1]) > 0.05              # the condition really changes the result
    with pytest.raises(ValueError, match="spatial dim"):
        gen(x.to(DEV), {"prompt_embeds": pe.to(DEV), "add_condition": cond[:, :-1].to(DEV)}, t.to(DEV), pipe.kv_cache1,
            pipe.crossattn_cache, 0)


def test_rollout_bit_reproducible_beside_vae_convolutions(sd_reduced):
    """The rollout must not depend on what runs on another HIP stream.  It once did: with the VAE's convolution
    (MFMA + LDS-DMA waves) sharing the CUs, compiler-formed packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32
    with op_sel) in the q/k RoPE kernel intermittently returned wrong values in lanes 48-63 -- 6 to 15 of 15 rollouts
    differed, in up to a third of the overlapped streaming runs.  The library is built with -fno-slp-vectorize
    since (csrc/Makefile); this test is the regression check for it."""
    from self_forcing_amd.vae import repack_conv
    g = torch.Generator().manual_seed(41)
    noise = torch.randn(1, 6, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16).to(DEV)
    pe = torch.randn(1, 512, sfa.WAN_REDUCED.text_dim, generator=g).to(torch.bfloat16).to(DEV)
    eps = [torch.randn(2, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(9)]
    pipe = make_pipe(sd_reduced, 2, False, 5.0, pe=pe)
    q = []
    pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
    xc = torch.randn(6, 64, 96, 64, generator=g).to(torch.bfloat16).to(DEV)
    wc = repack_conv((torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).to(torch.bfloat16)).to(DEV)
    bc = torch.randn(64, generator=g).to(torch.bfloat16).to(DEV)
    side = torch.cuda.Stream(device=DEV)

    def rollout():
        q.extend(eps)
        return pipe.inference(noise, ["p"], return_latents=True)[1].clone()

    ref = rollout()
    torch.cuda.synchronize()
    for _ in range(12):
        with torch.cuda.stream(side):
            for _ in range(200):
                sfa.ops.conv_igemm(xc, wc, bc, (3, 3, 3), 4)
        lat = rollout()
        torch.cuda.synchronize()
        assert torch.equal(lat, ref)


def test_add_condition_identity_projection_on_dim_5120_models():
    """`pose_proj = nn.Identity()` when dim == 5120 (causal_model.py:500-501; BASELINE configs[4] is that model): the pose
    tokens are added to the patch embedding as they are.  Two layers of the 14B geometry (40 heads of 128; ffn cut to
    2048 so the fp32 CPU oracle stays quick) against the oracle -- oracle-only like the rest of this branch (parity
    unpinned: the reference snapshot raises on it)."""
    shape = sfa.WanShape(dim=5120, ffn_dim=2048, num_heads=40, num_layers=2)
    sd = sfa.synth_state_dict(shape, seed=3, pose=True)
    assert "pose_proj.weight" not in sd                      # nothing to load for the identity
    g = torch.Generator().manual_seed(56)
    pe = torch.randn(1, 512, shape.text_dim, generator=g).to(torch.bfloat16)
    pe[:, 90:] = 0
    x = torch.randn(1, 2, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16)
    cond = (0.5 * torch.randn(1, 2 * FS, 5120, generator=g)).to(torch.bfloat16)
    t = torch.tensor([[833.3333129882812, 625.0]])
    gen = sfa.WanDiffusionWrapper(shape=shape, state_dict=sd, timestep_shift=5.0, is_causal=True, device=DEV)
    assert gen.model.accepts_pose and not gen.model.has_pose_proj
    args = SimpleNamespace(denoising_step_list=[1000], warp_denoising_step=False, independent_first_frame=False,
                           num_frame_per_block=1, context_noise=0)
    pipe = sfa.CausalInferencePipeline(args, DEV, generator=gen, text_encoder=sfa.FixedTextEncoder(pe.to(DEV)), vae=sfa.IdentityVAE())
    pipe.frame_seq_length = FS
    outs = []
    for c in (cond, None):
        pipe._initialize_kv_cache(1, torch.bfloat16, DEV, cache_tokens=2 * FS)
        pipe._initialize_crossattn_cache(1, torch.bfloat16, DEV)
        cd = {"prompt_embeds": pe.to(DEV)}
        if c is not None:
            cd["add_condition"] = c.to(DEV)
        outs.append(gen(x.to(DEV), cd, t.to(DEV), pipe.kv_cache1, pipe.crossattn_cache, 0)[0])
    Wf = wo.prepare_weights(sd, torch.float32)
    cfg = wo.OracleConfig(dim=shape.dim, ffn_dim=shape.ffn_dim, num_heads=shape.num_heads, num_layers=shape.num_layers,
                          text_dim=shape.text_dim)
    for out, c in zip(outs, (cond, None)):
        kv, ca = wo.init_kv_cache(cfg, 1, 2 * FS, torch.float32), wo.init_crossattn_cache(cfg, 1, torch.float32)
        ref = wo.wrapper_forward(Wf, cfg, wo.FlowMatchTables(5.0), x.float(), pe.float(), t, kv, ca, 0,
                                 add_condition=None if c is None else c.float())[0]
        assert rel(out, ref) < TOL
    assert rel(outs[0], outs[1]) > 0.05
    with pytest.raises(ValueError, match="spatial dim"):
        gen(x.to(DEV), {"prompt_embeds": pe.to(DEV), "add_condition": cond[:, :-1].to(DEV)}, t.to(DEV), pipe.kv_cache1,
            pipe.crossattn_cache, 0)
    # a dim-1536 model WITHOUT pose_proj weights cannot take pose tokens: refused, not guessed
    small = sfa.WanDiffusionWrapper(shape=sfa.WAN_REDUCED, state_dict=sfa.synth_state_dict(sfa.WAN_REDUCED, seed=0), timestep_shift=5.0,
                                    is_causal=True, device=DEV)
    pe2 = torch.randn(1, 512, sfa.WAN_REDUCED.text_dim, generator=g).to(torch.bfloat16).to(DEV)
    p2 = sfa.CausalInferencePipeline(args, DEV, generator=small, text_encoder=sfa.FixedTextEncoder(pe2), vae=sfa.IdentityVAE())
    p2.frame_seq_length = FS
    p2._initialize_kv_cache(1, torch.bfloat16, DEV, cache_tokens=2 * FS)
    p2._initialize_crossattn_cache(1, torch.bfloat16, DEV)
    with pytest.raises(ValueError, match="needs pose_proj weights"):
        small(x.to(DEV), {"prompt_embeds": pe2, "add_condition": cond.to(DEV)}, t.to(DEV), p2.kv_cache1, p2.crossattn_cache, 0)


@pytest.mark.parametrize("nfpb,frames,las,sink,batch", [(2, 6, -1, 0, 1), (1, 7, 3, 1, 1), (3, 6, -1, 0, 2), (1, 5, 2, 0, 2)])
def test_pairing_the_context_pass_with_the_next_chunks_first_pass_is_bit_identical(sd_reduced, nfpb, frames, las, sink, batch):
    """`sf_dit_forward_pair`: a chunk's context pass and the next chunk's first denoising pass as ONE call (twice the rows per
    GEMM) against the reference's order, one call per pass: the same latents, the same K / V in every layer's cache and the
    same cache indices, bit for bit -- global cache, rolling window with eviction in the paired pass, batch 1 and 2."""
    g = torch.Generator().manual_seed(91 + frames)
    noise = torch.randn(batch, frames, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16).to(DEV)
    pe = torch.randn(batch, 512, sfa.WAN_REDUCED.text_dim, generator=g).to(torch.bfloat16).to(DEV)
    eps = [torch.randn(batch * nfpb, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(3 * (frames // nfpb))]
    res = []
    for paired in (True, False):
        pipe = make_pipe(sd_reduced, nfpb, False, 5.0, las=las, sink=sink, pe=pe)
        pipe.pair_context_with_next = paired
        assert batch * nfpb * FS <= pipe.pair_max_rows
        calls = {"pair": 0, "single": 0}
        gen = pipe.generator
        fwd, fwd_pair = gen.forward, gen.forward_pair
        gen.forward = lambda *a, _f=fwd, **k: (calls.__setitem__("single", calls["single"] + 1), _f(*a, **k))[1]
        gen.forward_pair = lambda *a, _f=fwd_pair, **k: (calls.__setitem__("pair", calls["pair"] + 1), _f(*a, **k))[1]
        q = list(eps)
        pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
        lat = pipe.inference(noise, ["p"] * batch, return_latents=True)[1].clone()
        torch.cuda.synchronize()
        n_chunks = frames // nfpb
        assert calls["pair"] == (n_chunks - 1 if paired else 0) and calls["single"] + 2 * calls["pair"] == 5 * n_chunks
        res.append((lat, [kv["k"].clone() for kv in pipe.kv_cache1], [kv["v"].clone() for kv in pipe.kv_cache1],
                    int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[-1]["local_end_index"])))
    a, b = res
    assert torch.equal(a[0], b[0]) and a[3:] == b[3:]
    for ka, kb, va, vb in zip(a[1], b[1], a[2], b[2]):
        assert torch.equal(ka, kb) and torch.equal(va, vb)


def test_streaming_with_paired_passes_yields_the_same_chunks(sd_reduced):
    """`stream()` (chunk-at-a-time, last context pass skipped as demo.py does) with the context pass of chunk k paired with
    chunk k + 1's first pass: the same latents per chunk as with one call per pass, and as the batch `inference()`."""
    g = torch.Generator().manual_seed(97)
    noise = torch.randn(1, 6, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16).to(DEV)
    pe = torch.randn(1, 512, sfa.WAN_REDUCED.text_dim, generator=g).to(torch.bfloat16).to(DEV)
    eps = [torch.randn(2, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(9)]
    outs = {}
    for paired in (True, False):
        pipe = make_pipe(sd_reduced, 2, False, 5.0, pe=pe)
        pipe.pair_context_with_next = paired
        q = list(eps)
        pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
        outs[paired] = [x.clone() for _, x, _ in pipe.stream(noise, ["p"])]
        assert not q                                            # every re-noise tensor was consumed, in order
    assert len(outs[True]) == 3 and all(torch.equal(a, b) for a, b in zip(outs[True], outs[False]))
    pipe = make_pipe(sd_reduced, 2, False, 5.0, pe=pe)
    q = list(eps)
    pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
    lat = pipe.inference(noise, ["p"], return_latents=True)[1]
    assert torch.equal(torch.cat(outs[True], dim=1), lat)


def test_host_pacing_bounds_the_queue_and_changes_nothing(sd_reduced):
    """`WanDiffusionWrapper.max_inflight_forwards` (host-side pacing: the calling thread polls the oldest pass's event
    between sleeps instead of spinning for launch-queue room): never more than that many passes' events outstanding, and
    the latents are bit-identical with the pacing off."""
    g = torch.Generator().manual_seed(61)
    noise = torch.randn(1, 4, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16).to(DEV)
    pe = torch.randn(1, 512, sfa.WAN_REDUCED.text_dim, generator=g).to(torch.bfloat16).to(DEV)
    eps = [torch.randn(2, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(6)]
    outs = []
    for limit in (2, 0, 1):
        pipe = make_pipe(sd_reduced, 2, False, 5.0, pe=pe)
        pipe.generator.max_inflight_forwards = limit
        q = list(eps)
        pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
        seen = []
        orig = pipe.generator._pace

        def pace(device, _orig=orig, _gen=pipe.generator):
            _orig(device)
            seen.append(len(_gen._inflight))
        pipe.generator._pace = pace
        outs.append(pipe.inference(noise, ["p"], return_latents=True)[1].clone())
        assert len(seen) == 9 and max(seen) <= max(limit, 0)       # 2 chunks x (4 + 1) passes, one context pass paired with the next chunk's first
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def _conv_corunner(g):
    """The co-runner that triggered the packed-fp32 fault: a 3x3x3 convolution whose gathered
    k-loop is 54 slices long; returns a function that enqueues `n` of them on the current stream."""
    from self_forcing_amd.vae import repack_conv
    xc = torch.randn(6, 64, 96, 64, generator=g).to(torch.bfloat16).to(DEV)
    wc = repack_conv((torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).to(torch.bfloat16)).to(DEV)
    bc = torch.randn(64, generator=g).to(torch.bfloat16).to(DEV)

    def run(n):
        for _ in range(n):
            sfa.ops.conv_igemm(xc, wc, bc, (3, 3, 3), 4)
    return run


def test_torch_renoise_kernels_bit_reproducible_beside_vae_convolutions():
    """The kernels of torch's that a rollout still launches (the library does not own them and they are built with the
    SLP vectoriser ON): `torch.randn_like` -- the reference's global-RNG re-noise, causal_inference.py:208, which must
    stay torch's -- and the once-per-rollout `ones * t` broadcast.  Large tensors (thousands of waves, so that many share
    a CU with the convolution's waves), fixed seed: alone vs beside the 54-slice convolution, bit for bit."""
    g = torch.Generator().manual_seed(43)
    conv = _conv_corunner(g)
    side = torch.cuda.Stream(device=DEV)
    like = torch.empty(8, 16, 480, 832, dtype=torch.bfloat16, device=DEV)        # 49 M elements
    steps = torch.tensor([1000.0, 937.5, 833.3333, 625.0], device=DEV)
    ones = torch.ones([64, 4096], dtype=torch.int64, device=DEV)

    def torch_kernels():
        torch.manual_seed(777)
        a = torch.randn_like(like)
        b = torch.randn(3, 1 << 22, device=DEV)                                  # fp32 normal: the other template
        c = (ones.unsqueeze(0) * steps.reshape(-1, 1, 1)).contiguous()
        return a, b, c

    ref = torch_kernels()
    torch.cuda.synchronize()
    for _ in range(10):
        with torch.cuda.stream(side):
            conv(150)
        got = torch_kernels()
        torch.cuda.synchronize()
        for r, x in zip(ref, got):
            assert torch.equal(r, x)


def test_rollout_with_torch_renoise_bit_reproducible_beside_vae_convolutions(sd_reduced):
    """The DEFAULT path (noise_source = None: re-noise drawn by torch.randn_like from the global generator) under a fixed
    seed, alone vs beside the convolution co-runner: the same latents bit for bit.  (The regression test above injects
    pre-drawn noise and so never ran torch's generator kernel beside a convolution.)"""
    g = torch.Generator().manual_seed(47)
    noise = torch.randn(1, 6, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16).to(DEV)
    pe = torch.randn(1, 512, sfa.WAN_REDUCED.text_dim, generator=g).to(torch.bfloat16).to(DEV)
    pipe = make_pipe(sd_reduced, 2, False, 5.0, pe=pe)
    assert pipe.noise_source is None
    conv = _conv_corunner(g)
    side = torch.cuda.Stream(device=DEV)

    def rollout():
        torch.manual_seed(4242)
        return pipe.inference(noise, ["p"], return_latents=True)[1].clone()

    ref = rollout()
    torch.cuda.synchronize()
    torch.manual_seed(4243)
    assert not torch.equal(pipe.inference(noise, ["p"], return_latents=True)[1], ref)   # the seed really drives the re-noise
    for _ in range(10):
        with torch.cuda.stream(side):
            conv(200)
        lat = rollout()
        torch.cuda.synchronize()
        assert torch.equal(lat, ref)


def test_two_streams_rollout_plus_decode_match_sequential(sd_reduced):
    """RolloutPool with the real VAE in every pipeline: while one stream decodes its clip the other one is still
    denoising (the configuration of bench.py's rollout + decode rate).  Latents and pixels must equal the one-stream
    run bit for bit, for every job, over several rounds."""
    from self_forcing_amd import vae_weights as vw
    shape = sfa.WAN_REDUCED
    g = torch.Generator().manual_seed(17)
    jobs = []
    for j in range(6):
        noise = torch.randn(1, 4, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16)
        eps = [torch.randn(2, 16, LAT_H, LAT_W, generator=g).to(torch.bfloat16) for _ in range(6)]
        jobs.append((f"prompt {j}", noise, eps))
    args = SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                           independent_first_frame=False, num_frame_per_block=2, context_noise=0)
    gen = sfa.WanDiffusionWrapper(shape=shape, state_dict=sd_reduced, timestep_shift=5.0, is_causal=True, device=DEV)
    enc = sfa.SyntheticTextEncoder(shape.text_len, shape.text_dim, device=DEV)
    vsd = vw.synth_vae_state_dict(vw.VAE_REDUCED, seed=0)
    make_vae = lambda: sfa.WanVAEWrapper(vsd, device=DEV, shape=vw.VAE_REDUCED)  # noqa: E731

    def roll(pipe, job):
        prompt, noise, eps = job
        q = list(eps)
        pipe.noise_source = lambda t: q.pop(0).reshape(t.shape)
        video, lat = pipe.inference(noise.to(DEV), [prompt], return_latents=True)
        return lat.clone(), video.clone()

    seq = sfa.RolloutPool(args, DEV, gen, lambda: enc, make_vae, streams=1).run(jobs, roll)
    pool = sfa.RolloutPool(args, DEV, gen, lambda: enc, make_vae, streams=2)
    for _ in range(3):
        par = pool.run(jobs, roll)
        for (la, va), (lb, vb) in zip(seq, par):
            assert torch.equal(la, lb) and torch.equal(va, vb)
