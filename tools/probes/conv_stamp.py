"""Where a conv_halo patch (workgroup) spends its time: s_memtime stamps at kernel entry, after the prologue's barrier,
after the plane loop and after the epilogue's stores, per workgroup and wave group, from the diagnostic build
tools/probes/build_conv_stamp.sh (run it first; SF_HIP_LIB selects the library).
    SF_HIP_LIB=tools/probes/abl/libabl_cstamp.so python tools/probes/conv_stamp.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from self_forcing_amd import _lib, vae  # noqa: E402

assert "cstamp" in _lib.LIB_PATH, "run with SF_HIP_LIB=tools/probes/abl/libabl_cstamp.so"
g = torch.Generator().manual_seed(0)
# (Cin, Cout, H, W, T, residual epilogue, fused norm output)
for (ci, co, H, W, T, resid, norm) in [(96, 96, 480, 832, 4, False, False), (96, 96, 480, 832, 4, False, True), (96, 96, 480, 832, 4, True, True),
                                       (192, 192, 240, 416, 4, False, False), (192, 192, 240, 416, 4, True, True),
                                       (384, 384, 120, 208, 2, True, False), (384, 384, 60, 104, 1, True, False)]:
    x = torch.randn(T + 2, H, W, ci, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(co, ci, 3, 3, 3, generator=g) * 0.03).to(torch.bfloat16)
    wp = vae.repack_conv(w).cuda()
    b = torch.zeros(co, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(T, H, W, co, dtype=torch.bfloat16, device="cuda")
    r = torch.randn(T, H, W, co, generator=g).to(torch.bfloat16).cuda()
    nout = torch.empty(T, H, W, co, dtype=torch.bfloat16, device="cuda")
    gam = torch.ones(co, dtype=torch.bfloat16, device="cuda")
    a = _lib.ConvArgs()
    a.x, a.w, a.bias, a.out = x.data_ptr(), wp.data_ptr(), b.data_ptr(), out.data_ptr()
    a.Tout, a.H, a.W, a.Hin, a.Win, a.Cin, a.Cout, a.kt, a.kh, a.kw = T, H, W, H, W, ci, co, 3, 3, 3
    a.ldw, a.ldo, a.epilogue, a.structure = wp.stride(0), co, 0, _lib.CONV_STRUCTURES["halo"]
    if resid:
        a.resid, a.ldr, a.epilogue = r.data_ptr(), co, 1
    if norm:
        a.norm_out, a.norm_gamma, a.norm_ld, a.norm_frame_offset = nout.data_ptr(), gam.data_ptr(), co, 0
    nt = 6 if co % 192 == 0 else 3
    tiles_n = co // (32 * nt)
    if nt == 6 and not norm and T * ((H + 15) // 16) * ((W + 15) // 16) * tiles_n < 160:
        tiles_n *= 2
    tiles = T * ((H + 15) // 16) * ((W + 15) // 16) * tiles_n
    dbg = torch.zeros(tiles * 2 * 8, dtype=torch.int64, device="cuda")
    a.out_f32 = dbg.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(6):
        if i == 1:
            e0.record()
        _lib.check(_lib.lib().sf_conv_igemm(a, torch.cuda.current_stream().cuda_stream))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    d = dbg.view(tiles, 2, 8).cpu().double()
    clk = ((d[:, :, 3] - d[:, :, 0]) / ((d[:, :, 5] - d[:, :, 4]) / 100.0)).median().item()
    pro, loop, epi = (d[:, :, 1] - d[:, :, 0]) / clk, (d[:, :, 2] - d[:, :, 1]) / clk, (d[:, :, 3] - d[:, :, 2]) / clk
    fl = 2.0 * T * H * W * co * ci * 27
    print(f"{ci}->{co} {H}x{W} T={T} resid={int(resid)} norm={int(norm)}: {tiles} workgroups, {us:.0f} us = {fl / us / 1e6:.0f} TFLOP/s, shader clock {clk:.0f} MHz | "
          f"per workgroup (median us): prologue {pro.median():.2f}  plane loop {loop.median():.2f}  epilogue {epi.median():.2f}  "
          f"(p90 {pro.quantile(0.9):.2f} / {loop.quantile(0.9):.2f} / {epi.quantile(0.9):.2f})", flush=True)
