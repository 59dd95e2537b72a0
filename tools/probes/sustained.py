"""Does a kernel's rate depend on how long the chip has been under load?  The ffn.0 GEMM (20 launches replayed from a HIP
graph) timed cold, then every ~2 s while the same graph keeps the GPU busy for `seconds` (default 40)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from self_forcing_amd import ops  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
M, N, K = 4680, 8960, 1536
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
b = torch.zeros(N, dtype=torch.bfloat16, device="cuda")
o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
fn = lambda: ops.gemm(a, w, b, epilogue="gelu", out=o)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fn()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for _ in range(20):
        fn()


def measure():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); graph.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20


graph.replay(); torch.cuda.synchronize()
time.sleep(2.0)
print(f"idle 2 s, then: {measure():.1f} us  {measure():.1f} us  {measure():.1f} us", flush=True)
t0 = time.perf_counter()
nxt = 0.0
while time.perf_counter() - t0 < seconds:
    for _ in range(50):
        graph.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if el >= nxt:
        us = measure()
        print(f"after {el:5.1f} s of load: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s", flush=True)
        nxt += 2.0
time.sleep(5.0)
print(f"idle 5 s, then: {measure():.1f} us  {measure():.1f} us", flush=True)
