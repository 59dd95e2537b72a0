import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from self_forcing_amd import ops
def replay(fn, n=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(30): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * n)
gen = torch.Generator().manual_seed(0)
for B in (1, 2):
    q = torch.randn(B, 4680, 12, 128, generator=gen).to(torch.bfloat16).cuda()
    for lk in (128, 256, 512, 1024, 2048):
        k = torch.randn(B, lk, 12, 128, generator=gen).to(torch.bfloat16).cuda()
        v = torch.randn(B, lk, 12, 128, generator=gen).to(torch.bfloat16).cuda()
        row = []
        for st in ("auto", "w8", "w4", "r64"):
            try:
                us = replay(lambda: ops.attention(q, k, v, structure=st))
                row.append(f"{st} {us:6.1f} us ({4.0 * B * 4680 * lk * 1536 / us / 1e6:5.0f})")
            except Exception as e:
                row.append(f"{st} n/a")
        print(f"B={B} Lk={lk}: " + "  ".join(row), flush=True)
