"""Timing of sf_gemm_bf16 over a few (M, N, K, epilogue) to separate per-tile fixed cost from the k-loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from self_forcing_amd import ops
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
g = torch.Generator().manual_seed(0)
for (M, N, K, epi) in [(4680, 8960, 1536, "bias"), (4680, 8960, 1536, "gelu"), (4680, 8960, 3072, "bias"), (4608, 8960, 1536, "bias"),
                       (4608, 8192, 1536, "bias"), (4680, 4608, 1536, "bias"), (4680, 4608, 3072, "bias")]:
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda(); w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
    b = torch.zeros(N, dtype=torch.bfloat16, device="cuda"); o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ms = t(lambda: ops.gemm(x, w, b, epilogue=epi, out=o))
    print(f"M={M} N={N} K={K} {epi:5s}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
