"""Per-kernel Python entry points over the C-ABI (torch tensors in, torch tensors out).

torch is used only for device memory and the current stream; every computation below runs in
the hand-written HIP kernels of csrc/.  The operators that also exist as PyTorch custom ops (`torch_ops.py`:
gemm, attention, lincomb, add_noise) are called through `torch.ops.sf_hip.*`; the remaining per-kernel test entry
points bind the C-ABI directly.  All functions require CUDA(ROCm) bf16 tensors and
raise if the library is missing -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from . import torch_ops  # noqa: F401  (registers torch.ops.sf_hip.*)
from ._lib import (ACT_GELU, ACT_NONE, ACT_SILU, CONV_BIAS, CONV_BIAS_CLAMP_F32, CONV_BIAS_RESID, EPI_BIAS,
                   EPI_BIAS_GATE_RESID, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_F32, ConvArgs, GemmArgs, check, lib)

Tensor = torch.Tensor
_EPI = {"bias": EPI_BIAS, "gelu": EPI_BIAS_GELU, "resid": EPI_BIAS_RESID, "gate_resid": EPI_BIAS_GATE_RESID,
        "f32": EPI_F32}
_ACT = {None: ACT_NONE, "none": ACT_NONE, "silu": ACT_SILU, "gelu": ACT_GELU}


def stream_handle() -> int:
    return torch.cuda.current_stream().cuda_stream


def _bf16(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA/ROCm tensor (the HIP path has no CPU fallback)")
    if t.dtype != torch.bfloat16:
        raise ValueError(f"{name}: expected bfloat16, got {t.dtype}")
    return t


def _rows(t: Tensor, name: str) -> Tensor:
    """2-D view with unit inner stride."""
    _bf16(t, name)
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with contiguous rows, got shape {tuple(t.shape)} strides {t.stride()}")
    return t


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _timestep(t: Tensor):
    if t.dtype == torch.int64:
        return t.contiguous(), 1
    return t.to(torch.float32).contiguous(), 0


# --------------------------------------------------------------------------------------
def gemm(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, epilogue: str = "bias",
         resid: Optional[Tensor] = None, gate_mod: Optional[Tensor] = None, gate_e0: Optional[Tensor] = None,
         rows_per_group: int = 1, out: Optional[Tensor] = None, structure: str = "auto") -> Tensor:
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias).  gate_e0: [groups, N] view (row stride free).
    `structure` ("auto" | "t128" | "pp256" | "pp224" | "pp192" | "pp128") forces a tiling (tests / A-B timing)."""
    a, w = _rows(a, "a"), _rows(w, "w")
    epi, st = _EPI[epilogue], _lib.GEMM_STRUCTURES[structure]
    if out is None:
        return torch.ops.sf_hip.gemm(a, w, bias, epi, resid, gate_mod, gate_e0, rows_per_group, st)
    torch.ops.sf_hip.gemm_out(out, a, w, bias, epi, resid, gate_mod, gate_e0, rows_per_group, st)
    return out


def small_linear(x: Tensor, w: Tensor, bias: Optional[Tensor], act_in=None, act_out=None) -> Tensor:
    x, w = _rows(x, "x").contiguous(), _rows(w, "w").contiguous()
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    check(lib().sf_small_linear(x.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K,
                                _ACT[act_in], _ACT[act_out], stream_handle()), "sf_small_linear")
    return out


def sinusoid_embedding(t: Tensor, dim: int) -> Tensor:
    tt, is64 = _timestep(t.flatten())
    out = torch.empty(tt.numel(), dim, dtype=torch.bfloat16, device=t.device)
    check(lib().sf_sinusoid_embedding(tt.data_ptr(), is64, out.data_ptr(), tt.numel(), dim, stream_handle()),
          "sf_sinusoid_embedding")
    return out


def layernorm_modulate(x: Tensor, mod_shift: Tensor, mod_scale: Tensor, e0_shift: Tensor, e0_scale: Tensor,
                       rows_per_group: int, eps: float = 1e-6) -> Tensor:
    """x [M,C]; mod_* [C]; e0_* [groups, C] views sharing one row stride."""
    x = _rows(x, "x").contiguous()
    M, Cc = x.shape
    _rows(e0_shift, "e0_shift"), _rows(e0_scale, "e0_scale")
    if e0_shift.stride(0) != e0_scale.stride(0):
        raise ValueError("layernorm_modulate: e0_shift / e0_scale must share a row stride")
    out = torch.empty_like(x)
    check(lib().sf_layernorm_modulate(x.data_ptr(), out.data_ptr(), M, Cc, eps, mod_shift.data_ptr(), mod_scale.data_ptr(),
                                      e0_shift.data_ptr(), e0_scale.data_ptr(), e0_shift.stride(0), rows_per_group,
                                      stream_handle()), "sf_layernorm_modulate")
    return out


def layernorm_affine(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-6) -> Tensor:
    x = _rows(x, "x").contiguous()
    out = torch.empty_like(x)
    check(lib().sf_layernorm_affine(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                    eps, stream_handle()), "sf_layernorm_affine")
    return out


def rmsnorm(x: Tensor, weight: Tensor, eps: float = 1e-6, out: Optional[Tensor] = None) -> Tensor:
    x = _rows(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().sf_rmsnorm(x.data_ptr(), x.stride(0), weight.data_ptr(), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1],
                           eps, stream_handle()), "sf_rmsnorm")
    return out


def qkv_norm_rope_cache(qkv: Tensor, norm_q_w: Tensor, norm_k_w: Tensor, k_cache: Tensor, v_cache: Tensor,
                        rope_cos: Tensor, rope_sin: Tensor, grid, write_start: int, start_frame: int,
                        eps: float = 1e-6) -> Tensor:
    """qkv [B*L, 3C]; caches [B, S, H, D]; returns roped q [B*L, C]; writes K/V rows in place."""
    qkv = _rows(qkv, "qkv").contiguous()
    B, S, H, D = k_cache.shape
    f, h, w = grid
    Cc = H * D
    if not (k_cache.is_contiguous() and v_cache.is_contiguous()):
        raise ValueError("qkv_norm_rope_cache: caches must be contiguous [B, S, H, D]")
    q = torch.empty(qkv.shape[0], Cc, dtype=torch.bfloat16, device=qkv.device)
    check(lib().sf_qkv_norm_rope_cache(qkv.data_ptr(), norm_q_w.data_ptr(), norm_k_w.data_ptr(), q.data_ptr(), k_cache.data_ptr(),
                                       v_cache.data_ptr(), rope_cos.data_ptr(), rope_sin.data_ptr(), B, f, h, w, Cc, H, S,
                                       write_start, start_frame, eps, stream_handle()), "sf_qkv_norm_rope_cache")
    return q


def kv_evict(cache: Tensor, sink: int, evict: int, keep: int, scratch: Tensor) -> None:
    B, S, H, D = cache.shape
    check(lib().sf_kv_evict(cache.data_ptr(), B, S, H * D, sink, evict, keep, scratch.data_ptr(),
                            scratch.numel() * scratch.element_size(), stream_handle()), "sf_kv_evict")


def attention(q: Tensor, k: Tensor, v: Tensor, structure: str = "auto") -> Tensor:
    """q [B,Lq,H,128], k/v [B,Lk,H,128] (token/batch strides free, [H,D] contiguous) -> [B,Lq,H,128].
    `structure` ("auto" | "r64" | "w8" | "w4") forces a kernel structure (tests / A-B timing)."""
    return torch.ops.sf_hip.attention(q, k, v, _lib.ATTN_STRUCTURES[structure])


def patchify(x: Tensor) -> Tensor:
    """x [B, F, Cin, H, W] -> [B*F*(H/2)*(W/2), Cin*4]."""
    x = _bf16(x, "x").contiguous()
    B, F, Cin, H, W = x.shape
    out = torch.empty(B * F * (H // 2) * (W // 2), Cin * 4, dtype=torch.bfloat16, device=x.device)
    check(lib().sf_patchify(x.data_ptr(), out.data_ptr(), B, F, Cin, H, W, stream_handle()), "sf_patchify")
    return out


def unpatchify_x0(head_out: Tensor, xt: Tensor, timestep: Tensor, sigmas: Tensor, timesteps: Tensor):
    """head_out [B*F*h*w, 4*Cout]; xt [B,F,Cout,H,W]; timestep [B, G] -> (flow, x0) [B,F,Cout,H,W]."""
    xt = _bf16(xt, "xt").contiguous()
    B, F, Cout, H, W = xt.shape
    tt, is64 = _timestep(timestep)
    G = tt.shape[1]
    flow = torch.empty_like(xt)
    x0 = torch.empty_like(xt)
    check(lib().sf_unpatchify_x0(_rows(head_out, "head_out").contiguous().data_ptr(), xt.data_ptr(), tt.data_ptr(), is64,
                                 sigmas.data_ptr(), timesteps.data_ptr(), sigmas.numel(), flow.data_ptr(), x0.data_ptr(),
                                 B, F, G, Cout, H, W, stream_handle()), "sf_unpatchify_x0")
    return flow, x0


def add_noise(x0: Tensor, eps: Tensor, timestep: Tensor, sigmas: Tensor, timesteps: Tensor) -> Tensor:
    """(1 - sigma_t) x0 + sigma_t eps; x0/eps [N, ...], timestep [N]."""
    x0 = _bf16(x0, "x0").contiguous()
    eps = _bf16(eps, "eps").contiguous()
    tt, _ = _timestep(timestep.flatten())
    if tt.numel() != x0.shape[0]:
        raise ValueError(f"add_noise: {x0.shape[0]} samples but {tt.numel()} timesteps")
    return torch.ops.sf_hip.add_noise(x0, eps, tt, sigmas, timesteps)


LINCOMB_MAX = 6


def lincomb(tensors, coefs, out: Optional[Tensor] = None) -> Tensor:
    """out = sum_k coefs[k] * tensors[k]: bf16 tensors of one shape, fp32 accumulation, one final rounding (`out` may be
    one of the inputs).  The tensor arithmetic of the UniPC sampler and of the guidance blend (sf_lincomb_bf16)."""
    if not 1 <= len(tensors) <= LINCOMB_MAX or len(tensors) != len(coefs):
        raise ValueError(f"lincomb: 1..{LINCOMB_MAX} tensors with one coefficient each, got {len(tensors)} / {len(coefs)}")
    xs = [_bf16(t, f"tensors[{i}]").contiguous() for i, t in enumerate(tensors)]
    cf = [float(c) for c in coefs]
    if out is None:
        return torch.ops.sf_hip.lincomb(xs, cf)
    torch.ops.sf_hip.lincomb_out(out, xs, cf)
    return out


# ------------------------------------------------------------------------------------------
# VAE decode kernels (channels-last bf16 volumes)
def conv_igemm(x: Tensor, w_packed: Tensor, bias: Tensor, kernel, t_out: int, upsample: bool = False,
               t_in_offset: int = 0, resid: Optional[Tensor] = None, interleave: bool = False,
               clamp_f32: bool = False, cin: Optional[int] = None, structure: str = "auto") -> Tensor:
    """Implicit-GEMM convolution (sf_conv_igemm).  x [Tin, Hin, Win, Cin] channels-last with the history
    frames in front; w_packed from `vae.repack_conv`; kernel = (kt, kh, kw).  Returns [Tout, H, W, Cout]
    bf16 -- [2 Tout, H, W, Cout/2] with `interleave` -- or float32 [Tout, Cout, H, W] with `clamp_f32`."""
    _bf16(x, "x"), _bf16(w_packed, "w_packed"), _bf16(bias, "bias")
    if x.dim() != 4 or not x.is_contiguous():
        raise ValueError("conv_igemm: x must be a contiguous [T, H, W, C] volume")
    tin, hin, win, c = x.shape
    kt, kh, kw = kernel
    cout = w_packed.shape[0]
    H, W = (2 * hin, 2 * win) if upsample else (hin, win)
    if t_out + (kt - 1) + t_in_offset > tin:
        raise ValueError(f"conv_igemm: {tin} input frames do not cover {t_out} output frames (kt={kt}, offset={t_in_offset})")
    a = ConvArgs()
    a.x, a.w, a.bias = x.data_ptr(), w_packed.data_ptr(), bias.data_ptr()
    a.Tout, a.H, a.W, a.Hin, a.Win = t_out, H, W, hin, win
    a.Cin, a.Cout, a.kt, a.kh, a.kw = cin or c, cout, kt, kh, kw
    a.upsample, a.t_in_offset, a.ldw = int(upsample), t_in_offset, w_packed.stride(0)
    a.structure = _lib.CONV_STRUCTURES[structure]
    if clamp_f32:
        out = torch.empty(t_out, cout, H, W, dtype=torch.float32, device=x.device)
        a.out_f32, a.epilogue = out.data_ptr(), CONV_BIAS_CLAMP_F32
    else:
        co = cout // 2 if interleave else cout
        out = torch.empty(2 * t_out if interleave else t_out, H, W, co, dtype=torch.bfloat16, device=x.device)
        a.out, a.ldo, a.epilogue = out.data_ptr(), co, CONV_BIAS
        a.interleave_c = co if interleave else 0
        if resid is not None:
            _bf16(resid, "resid")
            if tuple(resid.shape) != tuple(out.shape) or not resid.is_contiguous():
                raise ValueError("conv_igemm: resid must match the output volume")
            a.resid, a.ldr, a.epilogue = resid.data_ptr(), cout, CONV_BIAS_RESID
    check(lib().sf_conv_igemm(a, stream_handle()), "sf_conv_igemm")
    return out


def rmsnorm_silu_cl(x: Tensor, gamma: Tensor, silu: bool = True) -> Tensor:
    """VAE RMS_norm over the last (channel) dim of a contiguous channels-last tensor, optional SiLU."""
    _bf16(x, "x"), _bf16(gamma, "gamma")
    if not x.is_contiguous():
        raise ValueError("rmsnorm_silu_cl: x must be contiguous")
    out = torch.empty_like(x)
    c = x.shape[-1]
    check(lib().sf_rmsnorm_silu_cl(x.data_ptr(), gamma.data_ptr(), out.data_ptr(), x.numel() // c, c, int(silu), stream_handle()),
          "sf_rmsnorm_silu_cl")
    return out


def softmax_rows(s: Tensor, scale: float, cols_padded: Optional[int] = None) -> Tensor:
    """bf16 softmax(scale * s) over the rows of a float32 matrix; columns up to cols_padded are zero."""
    if s.dtype != torch.float32 or s.dim() != 2 or s.stride(1) != 1 or not s.is_cuda:
        raise ValueError("softmax_rows: expected a CUDA float32 matrix with contiguous rows")
    rows, cols = s.shape
    cp = cols_padded or cols
    out = torch.empty(rows, cp, dtype=torch.bfloat16, device=s.device)
    check(lib().sf_softmax_rows(s.data_ptr(), s.stride(0), out.data_ptr(), cp, rows, cols, cp, float(scale), stream_handle()),
          "sf_softmax_rows")
    return out


# ------------------------------------------------------------------------------------------
# measured ceilings of the box (bench.py: roofline.measured_peak / hbm_measured_peak)
def probe_mfma(operands: Tensor, sink: Tensor, shape: str = "32x32x16", iters: int = 2000, workgroups: int = 256) -> float:
    """Launches the register-only bf16 MFMA loop (sf_probe_mfma) on the current stream; returns the launch's FLOPs.
    operands: >= 4096 bf16 (random); sink: >= workgroups * 256 float32."""
    _bf16(operands, "operands")
    if operands.numel() < 4096 or sink.dtype != torch.float32 or sink.numel() < workgroups * 256 or not sink.is_cuda:
        raise ValueError("probe_mfma: operands >= 4096 bf16 and a CUDA float32 sink of workgroups * 256 elements expected")
    fl = C.c_double(0.0)
    check(lib().sf_probe_mfma({"32x32x16": 0, "16x16x32": 1}[shape], iters, workgroups, operands.data_ptr(), sink.data_ptr(),
                              C.byref(fl), stream_handle()), "sf_probe_mfma")
    return fl.value


def probe_copy(src: Tensor, dst: Tensor) -> int:
    """Streaming copy src -> dst (sf_probe_copy) on the current stream; returns the bytes READ (as many are written)."""
    if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("probe_copy: contiguous CUDA tensors expected")
    nbytes = src.numel() * src.element_size()
    if nbytes != dst.numel() * dst.element_size() or nbytes % 16:
        raise ValueError("probe_copy: src and dst must hold the same number of bytes, a multiple of 16")
    check(lib().sf_probe_copy(src.data_ptr(), dst.data_ptr(), nbytes, stream_handle()), "sf_probe_copy")
    return nbytes
