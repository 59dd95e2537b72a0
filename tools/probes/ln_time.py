"""Device time of the LayerNorm + modulate / affine kernels at the 1.3B and 14B token counts (HIP-graph replay)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from self_forcing_amd import ops  # noqa: E402


def replay(fn, n=50):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (M, C, groups) in [(4680, 1536, 3), (9360, 1536, 6), (10800, 5120, 3), (4680, 2048, 3), (4680, 3072, 3)]:
    x = torch.randn(M, C).to(torch.bfloat16).cuda()
    mod = torch.randn(6, C).to(torch.bfloat16).cuda()
    e0 = torch.randn(groups, 6, C).to(torch.bfloat16).cuda()
    w, b = torch.randn(C).to(torch.bfloat16).cuda(), torch.randn(C).to(torch.bfloat16).cuda()
    us = replay(lambda: ops.layernorm_modulate(x, mod[0], mod[1], e0[:, 0], e0[:, 1], M // groups))
    us2 = replay(lambda: ops.layernorm_affine(x, w, b))
    by = 2.0 * M * C * 2
    print(f"M={M} C={C}: modulate {us:6.2f} us {by / us / 1e6:5.2f} TB/s | affine {us2:6.2f} us {by / us2 / 1e6:5.2f} TB/s", flush=True)

# the time projection (M = frames per chunk = 3, K = C, N = 6 C): weights streamed once
for (C,) in [(1536,), (5120,)]:
    xs = torch.randn(3, C).to(torch.bfloat16).cuda()
    w6 = (torch.randn(6 * C, C) * 0.02).to(torch.bfloat16).cuda()
    b6 = torch.zeros(6 * C, dtype=torch.bfloat16, device="cuda")
    us = replay(lambda: ops.small_linear(xs, w6, b6, act_in="silu"))
    print(f"small_linear M=3 K={C} N={6 * C}: {us:6.2f} us {6.0 * C * C * 2 / us / 1e6:5.2f} TB/s of weights", flush=True)
