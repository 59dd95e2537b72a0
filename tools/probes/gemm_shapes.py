"""Separates a GEMM structure's per-tile FIXED cost (workgroup start, prologue, epilogue) from its k-loop rate: the same
(M, N) at two K values, interleaved rounds in one process, best of each.
    python tools/probes/gemm_shapes.py [structures, default t128,pp256,pp224,pp192,pp128]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from self_forcing_amd import ops  # noqa: E402


def t(fn, it=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / it


structures = (sys.argv[1] if len(sys.argv) > 1 else "t128,pp256,pp224,pp192,pp128").split(",")
g = torch.Generator().manual_seed(0)
K1, K2 = 1536, 6144
for (M, N, epi) in [(4680, 8960, "gelu"), (4680, 8960, "bias"), (4680, 4608, "bias"), (4680, 1536, "resid"), (4096, 4096, "bias"),
                    (9360, 4608, "bias"), (9360, 1536, "resid")]:
    xs = {K: torch.randn(M, K, generator=g).to(torch.bfloat16).cuda() for K in (K1, K2)}
    ws = {K: (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda() for K in (K1, K2)}
    b = torch.zeros(N, dtype=torch.bfloat16, device="cuda")
    r = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    kw = {"resid": r} if epi == "resid" else {}
    best = {(st, K): 1e9 for st in structures for K in (K1, K2)}
    for rnd in range(4):
        order = structures if rnd % 2 == 0 else structures[::-1]
        for st in order:
            for K in (K1, K2):
                best[(st, K)] = min(best[(st, K)], t(lambda: ops.gemm(xs[K], ws[K], b, epilogue=epi, out=o, structure=st, **kw)))
    for st in structures:
        t1, t2 = best[(st, K1)] * 1e3, best[(st, K2)] * 1e3
        per_k = (t2 - t1) / ((K2 - K1) / 64)                 # us per 64-deep k-tile (whole launch)
        fixed = t1 - per_k * (K1 / 64)
        loop_tf = 2.0 * M * N * 64 / per_k / 1e6
        print(f"M={M} N={N} {epi:5s} {st:6s}: K={K1}: {t1:7.1f} us {2.0 * M * N * K1 / t1 / 1e6:7.1f} TF/s | K={K2}: {t2:7.1f} us "
              f"{2.0 * M * N * K2 / t2 / 1e6:7.1f} TF/s | k-loop {loop_tf:7.1f} TF/s, fixed {fixed:6.1f} us", flush=True)
