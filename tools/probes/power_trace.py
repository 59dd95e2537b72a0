"""Shader clock and socket power while kernels run: samples `rocm-smi` (or amd-smi) every ~0.1 s in a child process while
this process replays (a) the ffn.0 GEMM, (b) self-attention at Lk = 32760, (c) a LayerNorm, each for `seconds`.
Evidence for the clock the chip holds under an MFMA load."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from self_forcing_amd import ops  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:   # noqa: BLE001
            samples.append((time.perf_counter(), f"ERR {e}"))
            return
        sclk = re.search(r"sclk clock level[^\n]*\((\d+)Mhz\)", out)
        pw = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([0-9.]+)", out)
        samples.append((time.perf_counter(), int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None, out if not sclk and len(samples) < 1 else ""))


def graph_of(fn, n):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    return g


gen = torch.Generator().manual_seed(0)
a = torch.randn(4680, 1536, generator=gen).to(torch.bfloat16).cuda()
w = (torch.randn(8960, 1536, generator=gen) * 0.02).to(torch.bfloat16).cuda()
b = torch.zeros(8960, dtype=torch.bfloat16, device="cuda")
o = torch.empty(4680, 8960, dtype=torch.bfloat16, device="cuda")
q = torch.randn(1, 4680, 12, 128, generator=gen).to(torch.bfloat16).cuda()
k = torch.randn(1, 32760, 12, 128, generator=gen).to(torch.bfloat16).cuda()
v = torch.randn(1, 32760, 12, 128, generator=gen).to(torch.bfloat16).cuda()
mod = torch.randn(6, 1536).to(torch.bfloat16).cuda()
e0 = torch.randn(3, 6, 1536).to(torch.bfloat16).cuda()
work = [("ffn.0 GEMM (pp224)", graph_of(lambda: ops.gemm(a, w, b, epilogue="gelu", out=o), 20), 20 * 2.0 * 4680 * 8960 * 1536),
        ("self-attention Lk=32760", graph_of(lambda: ops.attention(q, k, v), 4), 4 * 4.0 * 1536 * 4680 * 32760),
        ("layernorm + modulate", graph_of(lambda: ops.layernorm_modulate(a, mod[0], mod[1], e0[:, 0], e0[:, 1], 1560), 50), 0.0)]
th = threading.Thread(target=sampler)
th.start()
time.sleep(1.5)
marks = [("idle", time.perf_counter())]
for name, g, fl in work:
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        n += 20
    el = time.perf_counter() - t0
    marks.append((name + (f" [{fl * n / el / 1e12:.0f} TFLOP/s]" if fl else ""), t0))
    time.sleep(1.5)
    marks.append(("idle", time.perf_counter() - 1.5))
stop = True
th.join()
if samples and isinstance(samples[0][1], str):
    print("sampler failed:", samples[0][1])
    sys.exit(0)
if samples and samples[0][3]:
    print("unparsed rocm-smi output:\n" + samples[0][3][:1500])
bounds = marks + [("end", time.perf_counter())]
for i in range(len(bounds) - 1):
    name, t0 = bounds[i]
    t1 = bounds[i + 1][1]
    seg = [s for s in samples if t0 + 0.3 <= s[0] <= t1 - 0.1 and s[1] is not None]
    if not seg:
        print(f"{name:45s} no samples")
        continue
    clk = sorted(s[1] for s in seg)
    pw = sorted(s[2] for s in seg if s[2] is not None)
    print(f"{name:45s} {len(seg):3d} samples  sclk median {clk[len(clk) // 2]} MHz (min {clk[0]}, max {clk[-1]})"
          + (f"  power median {pw[len(pw) // 2]:.0f} W (max {pw[-1]:.0f})" if pw else "  power n/a"), flush=True)
