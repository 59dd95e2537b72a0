"""Generate the umT5-encoder golden fixtures by running the REFERENCE on CPU (build container only).

TEST INFRASTRUCTURE.  Imports `wan/modules/t5.py` through `ref_shim`, builds `T5Encoder` (shared_pos=False as
`umt5_xxl`), loads the seeded weights of `self_forcing_amd.t5_weights.synth_t5_state_dict`, runs it on seeded
token ids with ragged lengths and applies the zero padding of `WanTextEncoder.forward`:

    tests/golden/t5_reduced.npz   2 layers, dim 512, 8 heads of 64, ffn 1024, vocab 512; ids [3, 160]
                                  (lengths 160, 37, 1; relative distances up to 159 reach the last bucket)

Stores ids, mask, the fp32 output, and the reference's own bf16 output.  Usage: python oracle/make_golden_t5.py
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
from self_forcing_amd import t5_weights as tw  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def build_reference(shape: tw.T5Shape, sd, dtype):
    ref_shim.load()
    import wan.modules.t5 as rt
    m = rt.T5Encoder(vocab=shape.vocab_size, dim=shape.dim, dim_attn=shape.dim_attn, dim_ffn=shape.dim_ffn,
                     num_heads=shape.num_heads, num_layers=shape.num_layers, num_buckets=shape.num_buckets,
                     shared_pos=False, dropout=0.1)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    return m.eval().requires_grad_(False).to(dtype)


def main():
    os.makedirs(OUT, exist_ok=True)
    shape, seed = tw.T5_REDUCED, 0
    sd = tw.synth_t5_state_dict(shape, seed=seed)
    g = torch.Generator().manual_seed(7)
    L, lens = 160, [160, 37, 1]
    ids = torch.randint(1, shape.vocab_size, (len(lens), L), generator=g)
    mask = torch.zeros(len(lens), L, dtype=torch.long)
    for i, n in enumerate(lens):
        mask[i, :n] = 1
        ids[i, n:] = 0
    out = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        m = build_reference(shape, sd, dtype)
        with torch.no_grad():
            ctx = m(ids, mask)
        for u, n in zip(ctx, mask.gt(0).sum(dim=1).long()):
            u[n:] = 0.0
        out[f"context_{tag}"] = ctx.float().numpy()
    noise = np.linalg.norm(out["context_bf16"] - out["context_f32"]) / np.linalg.norm(out["context_f32"])
    print(f"t5_reduced: context {out['context_f32'].shape}, rms {out['context_f32'].std():.3f}, reference bf16-vs-fp32 rel err {noise:.4f}")
    np.savez_compressed(os.path.join(OUT, "t5_reduced.npz"), ids=ids.numpy(), mask=mask.numpy(), seed=np.int64(seed),
                        ref_bf16_rel_err=np.float64(noise), **out)
    xxl_geometry()


def xxl_geometry():
    """umT5-XXL LAYER geometry (dim 4096, 64 heads of 64, gated-GELU ffn 10240) with 2 layers and a 512-row vocabulary:
    every kernel of the encoder at its real channel counts.  ids [2, 192] with lengths 192 and 45.  Stored: every 4th
    row x every 4th channel of the fp32 output, its full-tensor sum / abs-sum per prompt, and the reference's own
    bf16-vs-fp32 distance."""
    shape, seed = tw.T5Shape(vocab_size=512, num_layers=2), 3
    sd = tw.synth_t5_state_dict(shape, seed=seed)
    g = torch.Generator().manual_seed(11)
    L, lens = 192, [192, 45]
    ids = torch.randint(1, shape.vocab_size, (len(lens), L), generator=g)
    mask = torch.zeros(len(lens), L, dtype=torch.long)
    for i, n in enumerate(lens):
        mask[i, :n] = 1
        ids[i, n:] = 0
    res = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        m = build_reference(shape, sd, dtype)
        with torch.no_grad():
            ctx = m(ids, mask)
        for u, n in zip(ctx, mask.gt(0).sum(dim=1).long()):
            u[n:] = 0.0
        res[tag] = ctx.float()
    f = res["f32"]
    noise = ((res["bf16"] - f).norm() / f.norm()).item()
    print(f"t5_xxl_geometry: context {tuple(f.shape)}, rms {f.std():.3f}, reference bf16-vs-fp32 rel err {noise:.4f}")
    np.savez_compressed(os.path.join(OUT, "t5_xxl_geometry.npz"), ids=ids.numpy(), mask=mask.numpy(), seed=np.int64(seed),
                        context_f32_sub=f[:, ::4, ::4].numpy(), sums=f.double().sum(dim=(1, 2)).numpy(),
                        abs_sums=f.double().abs().sum(dim=(1, 2)).numpy(), ref_bf16_rel_err=np.float64(noise))


if __name__ == "__main__":
    main()
