"""Where a 256 x 256 ping-pong GEMM tile spends its time: s_memtime stamps (shader-clock cycles) at kernel entry, after the
prologue's barrier, after the k-loop and after the epilogue's stores, per workgroup and wave group, from the diagnostic
build tools/probes/build_gemm_stamp.sh (run it first; SF_HIP_LIB selects the library).
    SF_HIP_LIB=tools/probes/abl/libabl_stamp.so python tools/probes/gemm_stamp.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from self_forcing_amd import _lib  # noqa: E402

assert "stamp" in _lib.LIB_PATH, "run with SF_HIP_LIB=tools/probes/abl/libabl_stamp.so"
g = torch.Generator().manual_seed(0)
for (M, N, K) in [(4680, 8960, 1536), (9360, 4608, 1536), (9360, 1536, 1536), (4680, 1536, 8960)]:
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
    b = torch.zeros(N, dtype=torch.bfloat16, device="cuda")
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    dbg = torch.zeros(tiles * 2 * 8, dtype=torch.int64, device="cuda")
    a = _lib.GemmArgs()
    a.a, a.w, a.bias, a.out, a.gate_e0 = x.data_ptr(), w.data_ptr(), b.data_ptr(), o.data_ptr(), dbg.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldo, a.epilogue, a.rows_per_group, a.structure = M, N, K, K, K, N, 0, 1, _lib.GEMM_STRUCTURES["pp256"]
    for _ in range(20):
        _lib.check(_lib.lib().sf_gemm_bf16(a, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    d = dbg.view(tiles, 2, 8).cpu().double()
    clk = ((d[:, :, 3] - d[:, :, 0]) / ((d[:, :, 5] - d[:, :, 4]) / 100.0)).median().item()      # cycles per us (memrealtime: 100 MHz)
    pro, loop, epi = (d[:, :, 1] - d[:, :, 0]) / clk, (d[:, :, 2] - d[:, :, 1]) / clk, (d[:, :, 3] - d[:, :, 2]) / clk
    t0 = d[:, :, 4].min()
    start, end = (d[:, :, 4] - t0) / 100.0, (d[:, :, 5] - t0) / 100.0
    print(f"M={M} N={N} K={K}: {tiles} tiles, shader clock {clk:.0f} MHz | per workgroup (median us): prologue {pro.median():.2f}  k-loop {loop.median():.2f}  "
          f"epilogue {epi.median():.2f} | first start 0.0, median start {start.median():.1f}, last start {start.max():.1f}, last end {end.max():.1f} us", flush=True)
