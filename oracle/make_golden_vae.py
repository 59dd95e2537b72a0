"""Generate the VAE-decode golden fixtures by running the REFERENCE on CPU (build container only).

TEST INFRASTRUCTURE.  Imports `wan/modules/vae.py` from /root/reference through `ref_shim`, builds
`WanVAE_` with the decoder shape under test, loads the seeded weights of
`self_forcing_amd.vae_weights.synth_vae_state_dict` (strict=False: the encoder keeps its own init and is
never run), and records `decode` / `cached_decode` outputs for seeded latents:

    tests/golden/vae_reduced.npz   dim 32 decoder, latent [1,3,16,6,8]  -> [1,9,3,48,64]
    tests/golden/vae_full.npz      dim 96 decoder (the real widths), latent [1,2,16,4,6] -> [1,5,3,32,48]

Each file holds the latent, the fp32 output of `decode`, the output of two `cached_decode` calls
(frames [0:1] then [1:]) and the reference's own bf16 output (its distance from the fp32 one is the
noise floor the GPU tolerance is compared with).  Weights are NOT stored: both sides regenerate them
from the seed.  Usage: python oracle/make_golden_vae.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_shim  # noqa: E402
from self_forcing_amd import vae_weights as vw  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def build_reference(shape: vw.VaeShape, sd, dtype):
    ref_shim.load()
    import wan.modules.vae as rv
    m = rv.WanVAE_(dim=shape.dim, z_dim=shape.z_dim, dim_mult=list(shape.dim_mult),
                   num_res_blocks=shape.num_res_blocks, attn_scales=[],
                   temperal_downsample=list(shape.temperal_upsample[::-1]), dropout=0.0)
    missing, unexpected = m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("encoder.") or k.startswith("conv1.") for k in missing), missing
    return m.eval().requires_grad_(False).to(dtype)


def run(shape: vw.VaeShape, name: str, latent_shape, seed: int):
    sd = vw.synth_vae_state_dict(shape, seed=seed)
    g = torch.Generator().manual_seed(1000 + seed)
    latent = torch.randn(latent_shape, generator=g).to(torch.bfloat16)          # [B,F,C,H,W]
    mean, std = torch.tensor(vw.LATENT_MEAN), torch.tensor(vw.LATENT_STD)
    out = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        m = build_reference(shape, sd, dtype)
        z = latent.to(dtype).permute(0, 2, 1, 3, 4)
        scale = [mean.to(dtype), 1.0 / std.to(dtype)]
        with torch.no_grad():
            y = m.decode(z, scale).float().clamp_(-1, 1)                          # [B,3,T,H,W]
            m.clear_cache()
            y0 = m.cached_decode(z[:, :, :1], scale).float().clamp_(-1, 1)
            y1 = m.cached_decode(z[:, :, 1:], scale).float().clamp_(-1, 1)
            m.clear_cache()
        out[f"pixels_{tag}"] = y.permute(0, 2, 1, 3, 4).contiguous().numpy()
        out[f"stream_{tag}"] = torch.cat([y0, y1], 2).permute(0, 2, 1, 3, 4).contiguous().numpy()
    noise = np.linalg.norm(out["pixels_bf16"] - out["pixels_f32"]) / np.linalg.norm(out["pixels_f32"])
    print(f"{name}: pixels {out['pixels_f32'].shape}, rms {out['pixels_f32'].std():.3f}, "
          f"clamped {np.mean(np.abs(out['pixels_f32']) >= 1):.3f}, reference bf16-vs-fp32 rel err {noise:.4f}, "
          f"stream==decode {np.array_equal(out['pixels_f32'], out['stream_f32'])}")
    np.savez_compressed(os.path.join(OUT, f"vae_{name}.npz"), latent=latent.float().numpy(), seed=np.int64(seed),
                        shape_dim=np.int64(shape.dim), ref_bf16_rel_err=np.float64(noise), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    run(vw.VAE_REDUCED, "reduced", (1, 3, 16, 6, 8), seed=0)
    run(vw.WAN_VAE, "full", (1, 2, 16, 4, 6), seed=1)


if __name__ == "__main__":
    main()
