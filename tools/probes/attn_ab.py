"""A/B timing of several builds of the attention kernel in ONE process (interleaved rounds: clock drift and box
spread hit all variants alike), with a numerics check of every variant against the first.

    tools/probes/build_attn_variants.sh base "" new "--align 6" ...
    python tools/probes/attn_ab.py base new ... [--lk 4680,18720,32760] [--batch 1] [--rounds 5]
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--lk", default="4680,18720,32760")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--heads", type=int, default=12)
ap.add_argument("--n", type=int, default=4680)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()

libs = {}
for n in a.names:
    h = C.CDLL(os.path.join(ROOT, "tools", "probes", "abl", f"libattn_{n}.so"))
    h.sf_attention_ex.restype = C.c_int
    h.sf_attention_ex.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_int64] * 6 + [C.c_int, C.c_void_p]
    h.sf_last_error.restype = C.c_char_p
    libs[n] = h

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
B, H, N = a.batch, a.heads, a.n
q = torch.randn(B, N, H, 128, generator=g).to(torch.bfloat16).to(dev)
lks = [int(x) for x in a.lk.split(",")]
k = torch.randn(B, max(lks), H, 128, generator=g).to(torch.bfloat16).to(dev)
v = torch.randn(B, max(lks), H, 128, generator=g).to(torch.bfloat16).to(dev)
outs = {n: torch.empty_like(q) for n in a.names}
st = torch.cuda.current_stream().cuda_stream


def launch(n, lk):
    kk, vv, o = k[:, :lk], v[:, :lk], outs[n]
    rc = libs[n].sf_attention_ex(q.data_ptr(), kk.data_ptr(), vv.data_ptr(), o.data_ptr(), B, H, N, lk, q.stride(1), q.stride(0),
                                 kk.stride(1), kk.stride(0), o.stride(1), o.stride(0), 1, st)
    assert rc == 0, libs[n].sf_last_error()


for lk in lks:
    for n in a.names:
        launch(n, lk)
    torch.cuda.synchronize()
    ref = outs[a.names[0]].float()
    diffs = {n: ((outs[n].float() - ref).norm() / ref.norm()).item() for n in a.names}
    best = {n: 1e9 for n in a.names}
    for _ in range(a.rounds):
        for n in a.names:
            for _ in range(5):
                launch(n, lk)          # keep the clocks up between variants (no host sync in between)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                launch(n, lk)
            e1.record()
            e1.synchronize()
            best[n] = min(best[n], e0.elapsed_time(e1) / a.iters)
    fl = 4.0 * H * 128 * N * lk * B
    print(f"B={B} Lk={lk}: " + "  ".join(f"{n} {best[n] * 1e3:7.1f} us ({fl / best[n] / 1e9:6.1f} TF, diff {diffs[n]:.1e})" for n in a.names), flush=True)
