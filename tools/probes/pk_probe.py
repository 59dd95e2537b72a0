#!/usr/bin/env python
"""Driver of pk_probe.hip: the packed-fp32 victim kernel on the current stream, a co-runner on a second stream."""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import self_forcing_amd as sfa  # noqa: E402
from self_forcing_amd import ops  # noqa: E402
from self_forcing_amd.vae import repack_conv  # noqa: E402

so = os.path.join(HERE, "pk_probe.so")
if not os.path.exists(so):
    # -fno-slp-vectorize as the library: otherwise the "scalar" reference values are themselves compiler-formed packed
    # instructions (v_pk_mul_f32 ... op_sel_hi:[0,1], v_pk_fma_f32 ... neg_lo) and the table compares packed with packed
    os.system(f"/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -shared -fPIC {HERE}/pk_probe.hip -o {so}")
lib = C.CDLL(so)
lib.pk_probe_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
bf = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(torch.bfloat16).to(DEV)  # noqa: E731
xc, bc = bf(6, 64, 96, 64), bf(64)
wc = repack_conv((torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05).to(torch.bfloat16)).to(DEV)
A, W_, B_ = bf(4680, 1536), bf(4608, 1536, sc=0.03), bf(4608)
O_ = torch.empty(4680, 4608, dtype=torch.bfloat16, device=DEV)
xq, xk = bf(1, 4680, 12, 128), bf(1, 9360, 12, 128)
side = torch.cuda.Stream(device=DEV)
w1 = repack_conv((torch.randn(64, 64, 1, 1, 1, generator=g) * 0.05).to(torch.bfloat16)).to(DEV)
w133 = repack_conv((torch.randn(64, 64, 1, 3, 3, generator=g) * 0.05).to(torch.bfloat16)).to(DEV)
# three convolutions with the SAME k-loop length (54 slices of 32 channels) that differ in what the taps are:
# 3x3x3 / Cin 64 (temporal + spatial taps, out-of-range pieces at the image border: the known trigger),
# 1x3x3 / Cin 192 (spatial taps + out-of-range pieces only), 3x1x1 / Cin 576 (temporal taps only, nothing out of range)
x192, x576 = bf(6, 64, 96, 192), bf(6, 64, 96, 576)
w133_192 = repack_conv((torch.randn(64, 192, 1, 3, 3, generator=g) * 0.05).to(torch.bfloat16)).to(DEV)
w311_576 = repack_conv((torch.randn(64, 576, 3, 1, 1, generator=g) * 0.05).to(torch.bfloat16)).to(DEV)
w311 = repack_conv((torch.randn(64, 64, 3, 1, 1, generator=g) * 0.05).to(torch.bfloat16)).to(DEV)
loads = {
    "nothing": lambda: None,
    "conv_igemm 3x1x1 Cin 64 (temporal taps only, 6-slice k-loop) x600": lambda: [ops.conv_igemm(xc, w311, bc, (3, 1, 1), 4) for _ in range(600)],
    "conv_igemm 3x1x1 Cin 576 (temporal taps only, 54-slice k-loop) x300": lambda: [ops.conv_igemm(x576, w311_576, bc, (3, 1, 1), 4) for _ in range(300)],
    "conv_igemm 1x3x3 Cin 192 (spatial taps + border, 54-slice k-loop) x300": lambda: [ops.conv_igemm(x192, w133_192, bc, (1, 3, 3), 4) for _ in range(300)],
    "conv_igemm 1x1x1 (no padding, no out-of-range pieces) x600": lambda: [ops.conv_igemm(xc, w1, bc, (1, 1, 1), 4) for _ in range(600)],
    "conv_igemm 1x3x3 x400": lambda: [ops.conv_igemm(xc, w133, bc, (1, 3, 3), 4) for _ in range(400)],
    "conv_igemm x300": lambda: [ops.conv_igemm(xc, wc, bc, (3, 3, 3), 4) for _ in range(300)],
    "our gemm x200": lambda: [ops.gemm(A, W_, B_, out=O_) for _ in range(200)],
}
names = ["pk_fma plain", "pk_fma op_sel_hi:[0,1,1]", "pk_mul op_sel:[0,1] op_sel_hi:[0,0]", "pk_fma neg_lo/neg_hi", "scalar fma (control)",
         "pk_mul plain", "pk_mul op_sel:[1,0] op_sel_hi:[1,1]", "pk_add op_sel:[0,1] op_sel_hi:[0,0]", "pk_mul op_sel:[0,1] op_sel_hi:[1,0]"]
for lname, fn in loads.items():
    counts = torch.zeros(9 * 64, dtype=torch.int32, device=DEV)
    for rep in range(6):
        with torch.cuda.stream(side):
            fn()
        for k in range(20):
            lib.pk_probe_launch(counts.data_ptr(), 1024, 200, 0.5 + 0.01 * k, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    c = counts.view(9, 64).cpu()
    print(f"co-runner: {lname}")
    for v in range(9):
        q = [int(c[v, 16 * i:16 * i + 16].sum()) for i in range(4)]
        print(f"   {names[v]:38s} mismatches per lane quarter {q}")


# ---- the 64-row attention kernel (its rescale path holds plain v_pk_mul_f32) as the victim beside the 3x3x3
# convolution: bit-compare every launch with the solo result.  Inputs with a key spike per query block so that the
# rescale path (the only place of its packed instructions) is taken by every wave: the keys of the second half are 10x
# larger, so every row's maximum jumps by more than the 2^8 threshold there, and keeps creeping up afterwards.
q64, k64, v64 = bf(1, 4680, 12, 128), bf(1, 9360, 12, 128, sc=0.3), bf(1, 9360, 12, 128)
k64[:, 4680:] *= 10
solo = ops.attention(q64, k64, v64, structure="r64").clone()
torch.cuda.synchronize()
diff_launches = 0
for rep in range(6):
    with torch.cuda.stream(side):
        loads["conv_igemm x300"]()
    for _ in range(40):
        o = ops.attention(q64, k64, v64, structure="r64")
        diff_launches += int(not torch.equal(o, solo))
    torch.cuda.synchronize()
print(f"attention_r64 (plain v_pk_mul_f32 in its rescale path) beside conv_igemm 3x3x3: {diff_launches} of 240 launches differ from the solo result")
