"""Generate the golden fixtures of the 50-step CFG sampler by running the REFERENCE on CPU (build container only).

TEST INFRASTRUCTURE.  Imports `wan/utils/fm_solvers_unipc.py` and `pipeline/causal_diffusion_inference.py` through
`ref_shim.load_sampler()` (see there for what stands in for the absent diffusers mixins) and stores

    tests/golden/unipc_steps.npz     FlowUniPCMultistepScheduler alone, driven the way the pipeline drives it
                                     (constructor shift=1, set_timesteps(n, shift=s)): per configuration the
                                     timesteps, the sigmas and prev_sample after EVERY step, on a seeded float32
                                     sample [2, 3, 4, 8] with seeded "model outputs"
    tests/golden/unipc_rollouts.npz  CausalDiffusionInferencePipeline.inference at the reduced DiT shape
                                     (seeded weights of self_forcing_amd.synth_state_dict, two text conditions),
                                     latents in float32 and in the reference's own bf16

    tests/golden/unipc_full_1p3b.npz (--full)  the same pipeline at the FULL Wan-1.3B shape: one 1-frame chunk, 4 steps

Usage: python oracle/make_golden_unipc.py [--full]
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402
from oracle.make_golden import LAT_H, LAT_W, _IdentityVAE, bf16_randn, build_model, f32, fresh_caches  # noqa: E402
import self_forcing_amd as sfa  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

STEP_CONFIGS = {
    # name: (num_steps, shift, constructor kwargs)
    "s50_shift5": (50, 5.0, {}),
    "s8_shift3_order3": (8, 3.0, {"solver_order": 3}),
    "s6_shift1_order1": (6, 1.0, {"solver_order": 1}),
    "s10_bh1": (10, 5.0, {"solver_type": "bh1"}),
    "s10_eps": (10, 5.0, {"predict_x0": False}),
    "s10_nocorr01": (10, 8.0, {"disable_corrector": [0, 1]}),
    "s12_nolof": (12, 5.0, {"lower_order_final": False, "solver_order": 2}),
}


def gen_steps(ns):
    out = {}
    for ci, (name, (n, shift, kw)) in enumerate(STEP_CONFIGS.items()):
        g = torch.Generator().manual_seed(300 + ci)
        sch = ns.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False, **kw)
        sch.set_timesteps(n, device="cpu", shift=shift)
        x = torch.randn(2, 3, 4, 8, generator=g)
        vs = torch.randn(n, 2, 3, 4, 8, generator=g)
        out[f"{name}_x0"] = f32(x)
        out[f"{name}_v"] = f32(vs)
        out[f"{name}_timesteps"] = sch.timesteps.numpy().copy()
        out[f"{name}_sigmas"] = sch.sigmas.numpy().copy()
        traj = []
        # the final step (sigma -> 0, h = inf) is finite only for the configuration the sampler uses (bh2, x0
        # prediction, lower_order_final); the reference returns NaN there otherwise, and those are cut one step short
        finite_end = kw.get("lower_order_final", True) and kw.get("solver_type", "bh2") == "bh2" and kw.get("predict_x0", True)
        last = n - 1 if finite_end else n - 2
        for i, t in enumerate(sch.timesteps[:last + 1]):
            x = sch.step(vs[i], t, x, return_dict=False)[0]
            traj.append(f32(x))
        out[f"{name}_traj"] = np.stack(traj)
        print(name, "timesteps", sch.timesteps[:4].tolist(), "...", "final rms %.4f" % float(x.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(OUT, "unipc_steps.npz"), **out)
    print("unipc_steps.npz", len(out), "arrays")


ROLLOUTS = {
    # name: (nfpb, independent_first_frame, shift, guidance, sampling_steps, n_noise_frames, n_initial)
    "cfg_nfpb3": (3, False, 5.0, 3.0, 50, 6, 0),
    "cfg_iff": (3, True, 8.0, 5.0, 10, 4, 0),
    "cfg_ext": (3, False, 5.0, 6.0, 12, 3, 3),
    # start_frame_index = 2 (long-video window, causal_diffusion_inference.py:183, :234): RoPE positions and the
    # cache write position start two frames in, the first two frames' cache rows stay zero and ARE attended to
    "cfg_start2": (1, False, 5.0, 4.0, 8, 2, 0),
}
START_FRAME = {"cfg_start2": 2}


def run_reference(ns, shape, sd, dtype, name, noise, pe, ne, initial):
    nfpb, iff, shift, g, steps, nfr, nin = ROLLOUTS[name]
    fs = (LAT_H // 2) * (LAT_W // 2)
    model = build_model(ns, shape, sd, dtype)
    wrapper = ref_shim.build_wrapper(ns, model, shift)
    args = types.SimpleNamespace(num_train_timestep=1000, timestep_shift=shift, independent_first_frame=iff,
                                 num_frame_per_block=nfpb, negative_prompt="NEG", guidance_scale=g, model_kwargs={})

    def text_encoder(text_prompts):
        return {"prompt_embeds": (ne if text_prompts[0] == "NEG" else pe).to(dtype)}

    sink = io.StringIO()
    with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
        pipe = ns.CausalDiffusionInferencePipeline(args, device="cpu", generator=wrapper, text_encoder=text_encoder,
                                                   vae=_IdentityVAE(), image_encoder=object())
    # the reference hard-codes the 1.3B / 480p constants (causal_diffusion_inference.py:69-72, :464-487)
    pipe.num_transformer_blocks = shape.num_layers
    pipe.frame_seq_length = fs
    pipe.sampling_steps = steps
    tokens = (nfr + nin + START_FRAME.get(name, 0)) * fs
    pipe.kv_cache_pos, pipe.crossattn_cache_pos = fresh_caches(shape, noise.shape[0], tokens, dtype)
    pipe.kv_cache_neg, pipe.crossattn_cache_neg = fresh_caches(shape, noise.shape[0], tokens, dtype)
    with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink), torch.no_grad():
        _, lat = pipe.inference(noise.to(dtype), ["p"] * noise.shape[0], None, None, None,
                                initial_latent=None if initial is None else initial.to(dtype), return_latents=True,
                                start_frame_index=START_FRAME.get(name, 0))
    return lat, pipe.kv_cache_pos, pipe.kv_cache_neg


def gen_rollouts(ns):
    shape = sfa.WAN_REDUCED
    sd = sfa.synth_state_dict(shape, seed=0)
    out = {"weights_seed": np.array(0)}
    for si, name in enumerate(ROLLOUTS):
        nfpb, iff, shift, gd, steps, nfr, nin = ROLLOUTS[name]
        g = torch.Generator().manual_seed(2000 + si)
        noise = bf16_randn((1, nfr, 16, LAT_H, LAT_W), g)
        pe = bf16_randn((1, 512, shape.text_dim), g)
        pe[:, 60 + 10 * si:] = 0
        ne = bf16_randn((1, 512, shape.text_dim), g)
        ne[:, 20 + 5 * si:] = 0
        initial = bf16_randn((1, nin, 16, LAT_H, LAT_W), g) if nin else None
        out[f"{name}_noise"] = f32(noise)
        out[f"{name}_pe"] = f32(pe)
        out[f"{name}_ne"] = f32(ne)
        if initial is not None:
            out[f"{name}_initial"] = f32(initial)
        for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            lat, kvp, kvn = run_reference(ns, shape, sd, dtype, name, noise, pe, ne, initial)
            out[f"{name}_lat_{tag}"] = f32(lat)
            out[f"{name}_local_end"] = np.array(int(kvp[0]["local_end_index"]))
            out[f"{name}_global_end"] = np.array(int(kvn[0]["global_end_index"]))
            print(name, tag, "latents rms %.4f" % lat.float().pow(2).mean().sqrt().item())
        d = np.linalg.norm(out[f"{name}_lat_bf16"] - out[f"{name}_lat_f32"]) / np.linalg.norm(out[f"{name}_lat_f32"])
        print(name, "reference bf16 vs fp32 rel err %.4f" % d)
    np.savez_compressed(os.path.join(OUT, "unipc_rollouts.npz"), **out)
    print("unipc_rollouts.npz", len(out), "arrays")


def gen_full_shape(ns):
    """The CFG sampler at the FULL Wan-1.3B shape: one 1-frame chunk (1560 tokens), 4 UniPC steps, guidance 3, shift 5:
    2 x 4 + 2 forwards of the reference's CausalDiffusionInferencePipeline, fp32 math variant and bf16 as shipped."""
    import time
    shape = sfa.WAN_1_3B
    sd = sfa.synth_state_dict(shape, seed=0)
    H, W = 60, 104
    fs = (H // 2) * (W // 2)
    g = torch.Generator().manual_seed(6161)
    noise = bf16_randn((1, 1, 16, H, W), g)
    pe = bf16_randn((1, 512, shape.text_dim), g)
    pe[:, 88:] = 0
    ne = bf16_randn((1, 512, shape.text_dim), g)
    ne[:, 12:] = 0
    out = {"weights_seed": np.array(0), "input_seed": np.array(6161), "noise_checksum": np.array(noise.double().sum().item()),
           "pe_checksum": np.array(pe.double().sum().item()), "ne_checksum": np.array(ne.double().sum().item())}
    args = types.SimpleNamespace(num_train_timestep=1000, timestep_shift=5.0, independent_first_frame=False,
                                 num_frame_per_block=1, negative_prompt="NEG", guidance_scale=3.0, model_kwargs={})
    for tag, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        t0 = time.time()
        model = build_model(ns, shape, sd, dtype)
        wrapper = ref_shim.build_wrapper(ns, model, 5.0)
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            pipe = ns.CausalDiffusionInferencePipeline(
                args, device="cpu", generator=wrapper, vae=_IdentityVAE(), image_encoder=object(),
                text_encoder=lambda text_prompts: {"prompt_embeds": (ne if text_prompts[0] == "NEG" else pe).to(dtype)})
        pipe.sampling_steps = 4
        pipe.kv_cache_pos, pipe.crossattn_cache_pos = fresh_caches(shape, 1, fs, dtype)
        pipe.kv_cache_neg, pipe.crossattn_cache_neg = fresh_caches(shape, 1, fs, dtype)
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink), torch.no_grad():
            _, lat = pipe.inference(noise.to(dtype), ["p"], None, None, None, return_latents=True)
        out[f"lat_{tag}"] = f32(lat)
        print("cfg full shape", tag, "latents rms %.4f" % lat.float().pow(2).mean().sqrt().item(), "%.0f s" % (time.time() - t0), flush=True)
        del model, wrapper, pipe
    d = np.linalg.norm(out["lat_bf16"] - out["lat_f32"]) / np.linalg.norm(out["lat_f32"])
    out["ref_bf16_vs_f32"] = np.array(d)
    del out["lat_bf16"]
    print("cfg full shape: reference bf16 vs fp32 rel err %.4f" % d)
    np.savez_compressed(os.path.join(OUT, "unipc_full_1p3b.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ns = ref_shim.load_sampler()
    with torch.no_grad():
        if "--full" in sys.argv:          # ~4 minutes of CPU time
            gen_full_shape(ns)
            return
        gen_steps(ns)
        gen_rollouts(ns)


if __name__ == "__main__":
    main()
