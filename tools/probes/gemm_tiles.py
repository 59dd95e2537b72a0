"""Which row tile (256 / 224 / 192 / 128 rows x 256 columns, ping-pong structure) is fastest on the rollout's GEMM shapes, at one
and two prompts per call: every structure + the automatic choice, each timed as 20 launches replayed from a HIP graph (device
time only), interleaved rounds in one process, best of each.  Used to calibrate the cost model in sf_gemm_bf16.
    python tools/probes/gemm_tiles.py [M ...]    (default 4680 9360)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from self_forcing_amd import ops  # noqa: E402

REPS = 20
STRUCTURES = ["auto", "pp256", "pp224", "pp192", "pp128", "t128"]


def graph_of(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
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


gen = torch.Generator().manual_seed(0)
for M in [int(x) for x in sys.argv[1:]] or [4680, 9360]:
    for (N, K, epi) in [(4608, 1536, "bias"), (1536, 1536, "resid"), (8960, 1536, "gelu"), (1536, 8960, "resid")]:
        a = torch.randn(M, K, generator=gen).to(torch.bfloat16).cuda()
        w = (torch.randn(N, K, generator=gen) * 0.02).to(torch.bfloat16).cuda()
        b = torch.zeros(N, dtype=torch.bfloat16, device="cuda")
        r = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        kw = {"resid": r} if epi == "resid" else {}
        graphs = {st: graph_of(lambda st=st: ops.gemm(a, w, b, epilogue=epi, out=o, structure=st, **kw)) for st in STRUCTURES}
        best = {st: 1e9 for st in STRUCTURES}
        for rnd in range(5):
            for st in (STRUCTURES if rnd % 2 == 0 else STRUCTURES[::-1]):
                best[st] = min(best[st], replay_us(graphs[st]))
        fl = 2.0 * M * N * K
        fastest = min((st for st in STRUCTURES if st != "auto"), key=lambda st: best[st])
        print(f"M={M} N={N} K={K} {epi:5s}: " + "  ".join(f"{st} {best[st]:6.1f} us ({fl / best[st] / 1e6:5.0f})" for st in STRUCTURES)
              + f"  | fastest {fastest}" + ("" if abs(best["auto"] - best[fastest]) < 0.02 * best[fastest] else "  <-- auto differs"), flush=True)
