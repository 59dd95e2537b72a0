#!/usr/bin/env python
"""Kernel micro-benchmarks at the S1 shapes (for rocprofv3 --pmc passes and A/B timing):
self-attention N=4680 x Lk, the four GEMM shapes of one DiT block, and (--what vae) the VAE decode of
one 21-frame 60x104 latent clip."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from self_forcing_amd import ops  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="attn,gemm")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--n", type=int, default=4680)
    ap.add_argument("--lk", default="4680,18720,32760")
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--vae-frames", type=int, default=21)
    ap.add_argument("--vae-fpc", default="4", help="latent frames per sf_vae_decode_frames call (comma list)")
    ap.add_argument("--batch", type=int, default=1, help="samples per attention launch")
    ap.add_argument("--structures", default="auto", help="GEMM tilings to time, e.g. auto,t128,pp256,pp224,pp192,pp128")
    ap.add_argument("--rounds", type=int, default=1, help="interleaved timing rounds per (shape, structure); the best is printed")
    a = ap.parse_args()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    C = a.heads * 128
    if "attn" in a.what:
        q = torch.randn(a.batch, a.n, a.heads, 128, generator=g).to(torch.bfloat16).to(dev)
        for lk in [int(x) for x in a.lk.split(",")]:
            k = torch.randn(a.batch, lk, a.heads, 128, generator=g).to(torch.bfloat16).to(dev)
            v = torch.randn(a.batch, lk, a.heads, 128, generator=g).to(torch.bfloat16).to(dev)
            ms = timeit(lambda: ops.attention(q, k, v), a.iters)
            fl = 4.0 * C * a.n * lk * a.batch
            print(f"attention B={a.batch} N={a.n} Lk={lk} H={a.heads}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
    if "t5" in a.what:
        bench_t5(max(2, a.iters // 2))
    if "vae" in a.what:
        bench_vae(a.vae_frames, max(1, a.iters // 5), [int(x) for x in a.vae_fpc.split(",")])
    if "conv" in a.what:
        bench_conv(a.iters, a.rounds)
    if "gemm" in a.what:
        for (N, K, epi) in [(3 * C, C, "bias"), (C, C, "resid"), (8960 if C == 1536 else 13824, C, "gelu"),
                            (C, 8960 if C == 1536 else 13824, "resid")]:
            x = torch.randn(a.n, K, generator=g).to(torch.bfloat16).to(dev)
            w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).to(dev)
            b = torch.zeros(N, dtype=torch.bfloat16, device=dev)
            r = torch.zeros(a.n, N, dtype=torch.bfloat16, device=dev)
            o = torch.empty(a.n, N, dtype=torch.bfloat16, device=dev)
            kw = {"resid": r} if epi == "resid" else {}
            sts = a.structures.split(",")
            best = {st: 1e9 for st in sts}
            for _ in range(a.rounds):           # interleaved rounds in one process (clock drift hits all structures alike)
                for st in sts:
                    best[st] = min(best[st], timeit(lambda: ops.gemm(x, w, b, epilogue=epi, out=o, structure=st, **kw), a.iters))
            for st in sts:
                ms = best[st]
                print(f"gemm M={a.n} N={N} K={K} {epi:6s} {st:6s}: {ms * 1e3:8.1f} us  {2.0 * a.n * N * K / ms / 1e9:7.1f} TFLOP/s", flush=True)


def bench_t5(iters):
    """umT5-XXL encoder shape (24 layers, dim 4096, ffn 10240, 64 heads) on one 512-token prompt; the vocabulary is
    cut to 4096 rows (the embedding lookup is not what is timed) so that the random weights are drawn in a minute."""
    import time
    import self_forcing_amd as sfa
    from self_forcing_amd import t5_weights as tw
    shape = tw.T5Shape(vocab_size=4096)
    t0 = time.time()
    enc = sfa.WanTextEncoder(tw.synth_t5_state_dict(shape, seed=0), device="cuda:0", shape=shape)
    print(f"weights drawn and uploaded in {time.time() - t0:.0f} s ({enc.text_encoder.param_bytes() / 1e9:.1f} GB)", flush=True)
    ids = torch.randint(1, 4096, (1, 512), generator=torch.Generator().manual_seed(1))
    mask = torch.zeros(1, 512, dtype=torch.long)
    mask[:, :77] = 1
    ms = timeit(lambda: enc.encode_ids(ids, mask), iters)
    fl = 24 * (4 * 2.0 * 512 * 4096 * 4096 + 3 * 2.0 * 512 * 4096 * 10240 + 4.0 * 512 * 512 * 4096)
    print(f"umT5-XXL encode, 1 prompt x 512 tokens: {ms:8.2f} ms  {fl / ms / 1e9:7.1f} TFLOP/s ({fl / 1e12:.2f} TFLOP)", flush=True)


def bench_conv(iters, rounds):
    """The VAE decoder's big 3 x 3 x 3 convolutions at their real sizes (T = 4 output frames), both kernels."""
    from self_forcing_amd.vae import repack_conv
    g = torch.Generator().manual_seed(0)
    for (cin, cout, T, H, W) in [(96, 96, 4, 480, 832), (192, 192, 4, 240, 416), (384, 384, 2, 120, 208), (192, 96, 4, 480, 832)]:
        x = torch.randn(T + 2, H, W, cin, generator=g).to(torch.bfloat16).cuda()
        w = repack_conv((torch.randn(cout, cin, 3, 3, 3, generator=g) * (27 * cin) ** -0.5).to(torch.bfloat16)).cuda()
        b = torch.zeros(cout, dtype=torch.bfloat16, device="cuda")
        fl = 2.0 * T * H * W * cout * 27 * cin
        best = {"igemm": 1e9, "halo": 1e9}
        for rnd in range(max(1, rounds)):
            for st in (("igemm", "halo") if rnd % 2 == 0 else ("halo", "igemm")):
                best[st] = min(best[st], timeit(lambda: ops.conv_igemm(x, w, b, (3, 3, 3), T, structure=st), max(2, iters // 2)))
        for st, ms in best.items():
            print(f"conv3x3x3 Cin={cin} Cout={cout} T={T} {H}x{W} {st:5s}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


def bench_vae(frames, iters, fpcs=(4,)):
    import self_forcing_amd as sfa
    from self_forcing_amd import vae_weights as vw
    dev = "cuda:0"
    sd = vw.synth_vae_state_dict(vw.WAN_VAE, seed=0)
    lat = torch.randn(1, frames, 16, 60, 104, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(dev)
    fl = vw.vae_decode_flops(vw.WAN_VAE, 60, 104, frames)
    nf = 1 + 4 * (frames - 1)
    for fpc in fpcs:
        vae = sfa.WanVAEWrapper(sd, device=dev, frames_per_call=fpc)
        ms = timeit(lambda: vae.decode_to_pixel(lat), iters)
        print(f"vae decode {frames} latent frames -> {nf} frames 480x832, {fpc} latent frames per call: {ms:8.1f} ms  {nf / ms * 1e3:7.1f} frames/s  "
              f"{fl / ms / 1e9:7.1f} TFLOP/s ({fl / 1e12:.1f} TFLOP)", flush=True)
        del vae
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
