"""Library reference point for the four GEMM shapes of one DiT layer: torch.matmul (-> hipBLASLt) beside
sf_gemm_bf16 in ONE process, same operands, each timed as 20 launches replayed from a HIP graph (device time only:
eager launches of a 30 us kernel are host-bound through either path), interleaved rounds, best of each. Not part of
the product path: shows what the vendor library reaches on the same box at the same clock.
    python tools/probes/blaslt_ref.py [M, default 4680]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from self_forcing_amd import ops  # noqa: E402

REPS = 20


def graph_of(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    return g


def replay_us(g):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


M = int(sys.argv[1]) if len(sys.argv) > 1 else 4680
gen = torch.Generator().manual_seed(0)
for (N, K) in [(4608, 1536), (8960, 1536), (1536, 8960), (1536, 1536)]:
    a = torch.randn(M, K, generator=gen).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=gen) * 0.02).to(torch.bfloat16).cuda()
    b = torch.zeros(N, dtype=torch.bfloat16, device="cuda")
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    o2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    wt = w.t()
    g_lib = graph_of(lambda: torch.matmul(a, wt, out=o1))                       # no bias: the library's plain GEMM
    g_sf = graph_of(lambda: ops.gemm(a, w, b, epilogue="bias", out=o2))
    best = {"hipblaslt": 1e9, "sf_gemm": 1e9}
    for rnd in range(6):
        for name, g in ((("hipblaslt", g_lib), ("sf_gemm", g_sf)) if rnd % 2 == 0 else (("sf_gemm", g_sf), ("hipblaslt", g_lib))):
            best[name] = min(best[name], replay_us(g))
    fl = 2.0 * M * N * K
    err = (o1.float() - o2.float()).abs().max().item()
    print(f"M={M} N={N} K={K}: hipblaslt {best['hipblaslt']:7.1f} us {fl / best['hipblaslt'] / 1e6:7.1f} TF/s | "
          f"sf_gemm_bf16 {best['sf_gemm']:7.1f} us {fl / best['sf_gemm'] / 1e6:7.1f} TF/s | max |diff| {err:.3g}", flush=True)
