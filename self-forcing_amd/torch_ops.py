"""PyTorch custom operators over the C-ABI: `torch.ops.sf_hip.*` take Tensors (BASELINE north_star: "Python host code
calling HIP through PyTorch-ROCm custom ops over a thin C-ABI").

Every operator below is a thin shim: it validates shapes / dtypes, allocates the outputs with torch, reads the current
HIP stream, and makes ONE call into `libsf_hip.so` (`include/sf_hip.h`) with raw device pointers.  Registered with
`torch.library.custom_op`, so each has a schema (mutated arguments declared), a fake ("meta") implementation for
tracing, and is opaque-but-legal to `torch.compile` -- which the reference's second caller wraps around the generator
(demo.py:340); a bare ctypes call would be a graph break with unknown side effects.  `ops.py` (per-kernel wrappers),
`model.py` (the fused forward), `vae.py`, `text_encoder.py` route through these.

    sf_hip::attention(q, k, v, structure) -> out
    sf_hip::gemm(a, w, bias?, epilogue, resid?, gate_mod?, gate_e0?, rows_per_group, structure) -> out
    sf_hip::gemm_out(out!, a, w, ...) -> ()                         (caller-provided / aliased output)
    sf_hip::lincomb(tensors[], coefs[]) -> out ;  sf_hip::lincomb_out(out!, tensors[], coefs[]) -> ()
    sf_hip::add_noise(x0, eps, timestep, sigmas, timesteps) -> out
    sf_hip::dit_forward(model, noisy, timestep, prompt_embeds?, add_condition?, k_cache![], v_cache![], ck_cache![],
                        cv_cache![], workspace!, evict_scratch!?, ..., kv_index!?, global_end) -> (flow, x0)
    sf_hip::dit_forward_pair(model, ctx_noisy, ctx_timestep, noisy, timestep, caches![]..., workspace!, evict_scratch!?, ctx_plan[7],
                             plan[7], kv_index!?, global_end) -> (flow, x0)        (context pass of chunk k + first pass of chunk k + 1)
    sf_hip::vae_decode_frames(model, state!, scratch!, z, out!, h, w, window_frames, frame_index, window, history_at) -> ()
    sf_hip::t5_encode(model, ids, mask, buckets, workspace!) -> out

Models (weights + C descriptors) are Python objects that own device memory; operators take an integer HANDLE from
`register_model` (a constant to a tracer).  There is no CPU implementation: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import threading
import time
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch.library import custom_op

from . import _lib

Tensor = torch.Tensor
NAMESPACE = "sf_hip"

_MODELS: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()


def register_model(obj) -> int:
    """Handle of a model object (CausalWanModel / WanVAEDecoder / UMT5Encoder) for the operators that need its weights."""
    h = id(obj)
    _MODELS[h] = obj
    return h


def _model(handle: int):
    try:
        return _MODELS[handle]
    except KeyError:
        raise RuntimeError(f"sf_hip: model handle {handle} is not registered (or its owner was freed)") from None


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(t: Tensor, name: str, dtype=torch.bfloat16) -> None:
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a CUDA/ROCm tensor (the HIP path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------ attention
@custom_op(f"{NAMESPACE}::attention", mutates_args=())
def attention(q: Tensor, k: Tensor, v: Tensor, structure: int = 0) -> Tensor:
    for n, t in (("q", q), ("k", k), ("v", v)):
        _need_gpu(t, n)
        if t.dim() != 4 or t.shape[3] != 128 or t.stride(3) != 1 or t.stride(2) != 128:
            raise ValueError(f"attention: {n} must be [B, L, H, 128] with contiguous heads, got {tuple(t.shape)} {t.stride()}")
    B, Lq, H, D = q.shape
    if k.stride() != v.stride() or k.shape != v.shape:
        raise ValueError("attention: k and v must share shape and strides")
    out = torch.empty(B, Lq, H, D, dtype=torch.bfloat16, device=q.device)
    _lib.check(_lib.lib().sf_attention_ex(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Lq, k.shape[1],
                                          q.stride(1), q.stride(0), k.stride(1), k.stride(0), out.stride(1), out.stride(0),
                                          structure, _stream(q)), "sf_attention")
    return out


@attention.register_fake
def _(q, k, v, structure=0):
    return q.new_empty(q.shape)


# ------------------------------------------------------------------------------------------ GEMM
def _gemm_launch(out: Tensor, a: Tensor, w: Tensor, bias, epilogue: int, resid, gate_mod, gate_e0, rows_per_group: int,
                 structure: int) -> None:
    for n, t in (("a", a), ("w", w)):
        _need_gpu(t, n)
        if t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"gemm: {n} must be 2-D with contiguous rows, got {tuple(t.shape)} {t.stride()}")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm: a is [{M},{K}] but w is {tuple(w.shape)}")
    want = torch.float32 if epilogue == _lib.EPI_F32 else torch.bfloat16
    _need_gpu(out, "out", want)
    if out.dim() != 2 or out.stride(1) != 1 or tuple(out.shape) != (M, N):
        raise ValueError(f"gemm: out must be [{M},{N}] with contiguous rows, got {tuple(out.shape)} {out.stride()}")
    g = _lib.GemmArgs()
    g.a, g.w, g.bias, g.out = a.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldw, g.ldo = a.stride(0), w.stride(0), out.stride(0)
    g.epilogue, g.rows_per_group, g.structure = epilogue, rows_per_group, structure
    if resid is not None:
        _need_gpu(resid, "resid")
        g.resid, g.ldr = resid.data_ptr(), resid.stride(0)
    if gate_mod is not None:
        _need_gpu(gate_mod, "gate_mod")
        g.gate_mod = gate_mod.data_ptr()
    if gate_e0 is not None:
        _need_gpu(gate_e0, "gate_e0")
        g.gate_e0, g.gate_group_stride = gate_e0.data_ptr(), gate_e0.stride(0)
    _lib.check(_lib.lib().sf_gemm_bf16(g, _stream(a)), "sf_gemm_bf16")


@custom_op(f"{NAMESPACE}::gemm", mutates_args=())
def gemm(a: Tensor, w: Tensor, bias: Optional[Tensor], epilogue: int, resid: Optional[Tensor], gate_mod: Optional[Tensor],
         gate_e0: Optional[Tensor], rows_per_group: int, structure: int) -> Tensor:
    out = torch.empty(a.shape[0], w.shape[0], dtype=torch.float32 if epilogue == _lib.EPI_F32 else torch.bfloat16, device=a.device)
    _gemm_launch(out, a, w, bias, epilogue, resid, gate_mod, gate_e0, rows_per_group, structure)
    return out


@gemm.register_fake
def _(a, w, bias, epilogue, resid, gate_mod, gate_e0, rows_per_group, structure):
    return a.new_empty((a.shape[0], w.shape[0]), dtype=torch.float32 if epilogue == _lib.EPI_F32 else torch.bfloat16)


@custom_op(f"{NAMESPACE}::gemm_out", mutates_args=("out",))
def gemm_out(out: Tensor, a: Tensor, w: Tensor, bias: Optional[Tensor], epilogue: int, resid: Optional[Tensor],
             gate_mod: Optional[Tensor], gate_e0: Optional[Tensor], rows_per_group: int, structure: int) -> None:
    _gemm_launch(out, a, w, bias, epilogue, resid, gate_mod, gate_e0, rows_per_group, structure)


# ------------------------------------------------------------------------------------------ lincomb / add_noise
def _lincomb_launch(out: Tensor, tensors: Sequence[Tensor], coefs: Sequence[float]) -> None:
    n = len(tensors)
    if not 1 <= n <= 6 or n != len(coefs):
        raise ValueError(f"lincomb: 1..6 tensors with one coefficient each, got {n} / {len(coefs)}")
    for i, t in enumerate(tensors):
        _need_gpu(t, f"tensors[{i}]")
        if not t.is_contiguous() or t.shape != tensors[0].shape or t.device != tensors[0].device:
            raise ValueError("lincomb: tensors must be contiguous and share shape and device")
    if out.shape != tensors[0].shape or out.dtype != torch.bfloat16 or not out.is_contiguous():
        raise ValueError("lincomb: out must be a contiguous bf16 tensor of the inputs' shape")
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    cf = (C.c_float * n)(*[float(c) for c in coefs])
    _lib.check(_lib.lib().sf_lincomb_bf16(out.data_ptr(), ptrs, cf, n, tensors[0].numel(), _stream(out)), "sf_lincomb_bf16")


@custom_op(f"{NAMESPACE}::lincomb", mutates_args=())
def lincomb(tensors: List[Tensor], coefs: List[float]) -> Tensor:
    if not tensors:
        raise ValueError("lincomb: 1..6 tensors with one coefficient each, got 0")
    out = torch.empty_like(tensors[0], memory_format=torch.contiguous_format)
    _lincomb_launch(out, tensors, coefs)
    return out


@lincomb.register_fake
def _(tensors, coefs):
    return torch.empty_like(tensors[0])


@custom_op(f"{NAMESPACE}::lincomb_out", mutates_args=("out",))
def lincomb_out(out: Tensor, tensors: List[Tensor], coefs: List[float]) -> None:
    _lincomb_launch(out, tensors, coefs)


@custom_op(f"{NAMESPACE}::add_noise", mutates_args=())
def add_noise(x0: Tensor, eps: Tensor, timestep: Tensor, sigmas: Tensor, timesteps: Tensor) -> Tensor:
    _need_gpu(x0, "x0"), _need_gpu(eps, "eps"), _need_gpu(sigmas, "sigmas", torch.float32), _need_gpu(timesteps, "timesteps", torch.float32)
    if not (x0.is_contiguous() and eps.is_contiguous() and timestep.is_contiguous()):
        raise ValueError("add_noise: contiguous tensors expected")
    n = x0.shape[0]
    if timestep.numel() != n or timestep.dtype not in (torch.float32, torch.int64):
        raise ValueError(f"add_noise: {n} samples need {n} float32 / int64 timesteps, got {timestep.numel()} of {timestep.dtype}")
    out = torch.empty_like(eps)
    _lib.check(_lib.lib().sf_add_noise(x0.data_ptr(), eps.data_ptr(), timestep.data_ptr(), int(timestep.dtype == torch.int64),
                                       sigmas.data_ptr(), timesteps.data_ptr(), sigmas.numel(), out.data_ptr(), n, x0.numel() // n,
                                       _stream(x0)), "sf_add_noise")
    return out


@add_noise.register_fake
def _(x0, eps, timestep, sigmas, timesteps):
    return torch.empty_like(eps)


# ------------------------------------------------------------------------------------------ fused DiT forward
_TABLES: Dict[int, tuple] = {}

# Host time spent INSIDE sf_dit_forward (the C call that enqueues the ~430 launches of a pass), per calling thread:
# {thread id: [calls, wall seconds, CPU seconds of the thread]}.  Wall time includes waiting for room in the stream's
# launch queue when the host runs ahead of the GPU (it does: a pass is ~25-50 ms of GPU work); the thread's CPU time is
# what the call really costs the host.  bench.py reports both -- with one process per GPU and `streams` enqueuing
# threads per process, 8 ranks x that many threads must fit the node's cores.
HOST_ENQUEUE: Dict[int, list] = {}


def host_enqueue_stats(reset: bool = False):
    """(forwards, wall seconds, CPU seconds, threads) summed over the threads that called dit_forward since the last reset."""
    calls = sum(v[0] for v in HOST_ENQUEUE.values())
    wall = sum(v[1] for v in HOST_ENQUEUE.values())
    cpu = sum(v[2] for v in HOST_ENQUEUE.values())
    n = sum(1 for v in HOST_ENQUEUE.values() if v[0])
    if reset:
        HOST_ENQUEUE.clear()
    return calls, wall, cpu, n


def _pointer_tables(handle: int, k: Sequence[Tensor], v: Sequence[Tensor], ck: Sequence[Tensor], cv: Sequence[Tensor],
                    want_kv: tuple, want_cross: tuple, device):
    """Per-layer cache pointer arrays for the C call; rebuilt -- and every tensor validated (bf16, contiguous, the
    expected [B, S, H, D] on the call's device: the kernels WRITE through these pointers) -- only when a cache tensor
    was re-allocated / rebound or the expected shape changed."""
    key = (want_kv, want_cross) + tuple(t.data_ptr() for t in k) + tuple(t.data_ptr() for t in v) + tuple(t.data_ptr() for t in ck) \
        + tuple(t.data_ptr() for t in cv)
    slot = (handle, torch.cuda.current_stream(device).cuda_stream)
    hit = _TABLES.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    for name, tensors, want in (("k_cache", k, want_kv), ("v_cache", v, want_kv), ("ck_cache", ck, want_cross), ("cv_cache", cv, want_cross)):
        for i, t in enumerate(tensors):
            if not t.is_cuda or t.dtype != torch.bfloat16 or tuple(t.shape) != want or not t.is_contiguous() or t.device != device:
                raise ValueError(f"dit_forward: {name}[{i}] must be a contiguous bf16 tensor {want} on {device}, got "
                                 f"{tuple(t.shape)} {t.dtype} {t.device}")
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
    tabs = (arr(k), arr(v), arr(ck), arr(cv))
    if len(_TABLES) > 64:
        _TABLES.clear()
    _TABLES[slot] = (key, tabs)
    return tabs


def _forward_args(m, model: int, noisy: Tensor, timestep: Tensor, prompt_embeds: Optional[Tensor], add_condition: Optional[Tensor],
                  k_cache, v_cache, ck_cache, cv_cache, workspace: Tensor, evict_scratch: Optional[Tensor], init_cross: bool,
                  cache_only: bool, sink: int, evict: int, keep: int, write_start: int, attn_start: int, attn_end: int,
                  start_frame: int, kv_index: Optional[Tensor], global_end: int):
    """Validate one pass's tensors and fill its sf_forward_args; returns (args, flow, x0) with empty outputs for cache_only."""
    _need_gpu(noisy, "noisy")
    _need_gpu(timestep, "timestep", None)
    if noisy.dim() != 5 or not noisy.is_contiguous() or timestep.dim() != 2 or not timestep.is_contiguous():
        raise ValueError("dit_forward: noisy must be contiguous [B, F, C, H, W], timestep contiguous [B, groups]")
    if timestep.dtype not in (torch.float32, torch.int64):
        raise ValueError(f"dit_forward: timestep must be float32 or int64, got {timestep.dtype}")
    L = m.num_layers
    if not (len(k_cache) == len(v_cache) == len(ck_cache) == len(cv_cache) == L):
        raise ValueError(f"dit_forward: cache lists must have {L} entries")
    B, F, _, H, W = noisy.shape
    sh = m.shape
    if noisy.shape[2] != sh.in_dim or timestep.shape[0] != B:
        raise ValueError(f"dit_forward: noisy must carry {sh.in_dim} channels and timestep one row per sample")
    cap = k_cache[0].shape[1] if k_cache[0].dim() == 4 else -1
    # the C call writes through these pointers: every cache tensor must really be what the kernels assume (checked when
    # the pointer tables are (re)built, i.e. whenever any cache tensor is new to this model / stream)
    want_kv, want_cross = (B, cap, sh.num_heads, sh.head_dim), (B, sh.text_len, sh.num_heads, sh.head_dim)
    n_new = F * (H // 2) * (W // 2)
    if not (0 <= attn_start < attn_end <= cap and 0 <= write_start and write_start + n_new == attn_end):
        raise ValueError(f"dit_forward: cache plan does not fit the cache (write_start {write_start} + {n_new} new tokens, "
                         f"window [{attn_start}, {attn_end}), capacity {cap})")
    if init_cross:
        if prompt_embeds is None:
            raise ValueError("dit_forward: init_cross needs prompt_embeds")
        _need_gpu(prompt_embeds, "prompt_embeds")
        if tuple(prompt_embeds.shape) != (B, sh.text_len, sh.text_dim) or not prompt_embeds.is_contiguous():
            raise ValueError(f"dit_forward: prompt_embeds must be contiguous [{B}, {sh.text_len}, {sh.text_dim}], got {tuple(prompt_embeds.shape)}")
    if add_condition is not None:
        _need_gpu(add_condition, "add_condition")
        if add_condition.dim() != 3 or add_condition.shape[:2] != (B, n_new) or not add_condition.is_contiguous():
            raise ValueError(f"dit_forward: add_condition must be contiguous [{B}, {n_new}, pose_dim], got {tuple(add_condition.shape)}")
    if not workspace.is_cuda or workspace.dtype != torch.uint8 or not workspace.is_contiguous():
        raise ValueError("dit_forward: workspace must be a contiguous CUDA uint8 tensor")
    if evict_scratch is not None and (not evict_scratch.is_cuda or evict_scratch.dtype != torch.uint8 or not evict_scratch.is_contiguous()):
        raise ValueError("dit_forward: evict_scratch must be a contiguous CUDA uint8 tensor")
    k_ptrs, v_ptrs, ck_ptrs, cv_ptrs = _pointer_tables(model, k_cache, v_cache, ck_cache, cv_cache, want_kv, want_cross, noisy.device)
    a = _lib.ForwardArgs()
    a.batch, a.frames, a.lat_h, a.lat_w, a.groups = B, F, H, W, timestep.shape[1]
    a.noisy, a.timestep = noisy.data_ptr(), timestep.data_ptr()
    a.t_is_int64 = 1 if timestep.dtype == torch.int64 else 0
    a.prompt_embeds = _ptr(prompt_embeds)
    a.init_cross = 1 if init_cross else 0
    a.add_condition = _ptr(add_condition)
    a.k_cache_host, a.v_cache_host, a.ck_cache_host, a.cv_cache_host = k_ptrs, v_ptrs, ck_ptrs, cv_ptrs
    a.cache_tokens = cap
    a.sink_tokens, a.evict, a.keep = sink, evict, keep
    a.write_start, a.attn_start, a.attn_end, a.start_frame = write_start, attn_start, attn_end, start_frame
    if evict_scratch is not None:
        a.evict_scratch, a.evict_scratch_bytes = evict_scratch.data_ptr(), evict_scratch.numel() * evict_scratch.element_size()
    a.cache_only = 1 if cache_only else 0
    if cache_only:
        flow = torch.empty(0, dtype=torch.bfloat16, device=noisy.device)
        x0 = torch.empty(0, dtype=torch.bfloat16, device=noisy.device)
    else:
        flow = torch.empty(B, F, m.shape.out_dim, H, W, dtype=torch.bfloat16, device=noisy.device)
        x0 = torch.empty_like(flow)
        a.flow_out, a.x0_out = flow.data_ptr(), x0.data_ptr()
    a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    if kv_index is not None:
        if kv_index.dtype != torch.int64 or not kv_index.is_cuda or not kv_index.is_contiguous() or tuple(kv_index.shape) != (L, 2):
            raise ValueError(f"dit_forward: kv_index must be a contiguous CUDA int64 [{L}, 2] tensor")
        a.kv_index_out, a.global_end = kv_index.data_ptr(), global_end
    return a, flow, x0


def _timed_call(fn, *args) -> int:
    """The C call, with its wall and CPU time booked to the calling thread (HOST_ENQUEUE)."""
    t0, c0 = time.perf_counter(), time.thread_time()
    rc = fn(*args)
    dt, dc = time.perf_counter() - t0, time.thread_time() - c0
    rec = HOST_ENQUEUE.setdefault(threading.get_ident(), [0, 0.0, 0.0])
    rec[0] += 1
    rec[1] += dt
    rec[2] += dc
    return rc


@custom_op(f"{NAMESPACE}::dit_forward",
           mutates_args=("k_cache", "v_cache", "ck_cache", "cv_cache", "workspace", "evict_scratch", "kv_index"))
def dit_forward(model: int, noisy: Tensor, timestep: Tensor, prompt_embeds: Optional[Tensor], add_condition: Optional[Tensor],
                k_cache: List[Tensor], v_cache: List[Tensor], ck_cache: List[Tensor], cv_cache: List[Tensor],
                workspace: Tensor, evict_scratch: Optional[Tensor], init_cross: bool, cache_only: bool, sink: int, evict: int,
                keep: int, write_start: int, attn_start: int, attn_end: int, start_frame: int,
                kv_index: Optional[Tensor], global_end: int) -> Tuple[Tensor, Tensor]:
    """One denoiser pass (CausalWanModel._forward_inference + flow -> x0, sf_dit_forward).  Writes the new K/V rows
    into k_cache / v_cache (and, with init_cross, the text K/V into ck_cache / cv_cache); returns (flow, x0), or two
    empty tensors with cache_only.  kv_index (optional, int64 [num_layers, 2]): every row <- (global_end, attn_end),
    the cache dicts' index tensors when they are views of one buffer."""
    m = _model(model)
    a, flow, x0 = _forward_args(m, model, noisy, timestep, prompt_embeds, add_condition, k_cache, v_cache, ck_cache, cv_cache, workspace,
                                evict_scratch, init_cross, cache_only, sink, evict, keep, write_start, attn_start, attn_end, start_frame,
                                kv_index, global_end)
    _lib.check(_timed_call(_lib.lib().sf_dit_forward, C.byref(m.cmodel), C.byref(a), _stream(noisy)), "sf_dit_forward")
    return flow, x0


@custom_op(f"{NAMESPACE}::dit_forward_pair",
           mutates_args=("k_cache", "v_cache", "ck_cache", "cv_cache", "workspace", "evict_scratch", "kv_index"))
def dit_forward_pair(model: int, ctx_noisy: Tensor, ctx_timestep: Tensor, noisy: Tensor, timestep: Tensor,
                     k_cache: List[Tensor], v_cache: List[Tensor], ck_cache: List[Tensor], cv_cache: List[Tensor],
                     workspace: Tensor, evict_scratch: Optional[Tensor], ctx_plan: List[int], plan: List[int],
                     kv_index: Optional[Tensor], global_end: int) -> Tuple[Tensor, Tensor]:
    """The context pass of one chunk (cache_only) and the first denoising pass of the next in ONE call
    (sf_dit_forward_pair): bit-identical to the two calls, twice the rows per GEMM.  `ctx_plan` / `plan` =
    [sink, evict, keep, write_start, attn_start, attn_end, start_frame] of the two passes; workspace sized for 2 x batch.
    Returns (flow, x0) of the denoising pass."""
    m = _model(model)
    if len(ctx_plan) != 7 or len(plan) != 7 or ctx_noisy.shape != noisy.shape or ctx_timestep.shape != timestep.shape:
        raise ValueError("dit_forward_pair: two passes of one shape with 7 plan integers each expected")
    a0, _, _ = _forward_args(m, model, ctx_noisy, ctx_timestep, None, None, k_cache, v_cache, ck_cache, cv_cache, workspace, evict_scratch,
                             False, True, *ctx_plan, None, 0)
    a1, flow, x0 = _forward_args(m, model, noisy, timestep, None, None, k_cache, v_cache, ck_cache, cv_cache, workspace, evict_scratch,
                                 False, False, *plan, kv_index, global_end)
    _lib.check(_timed_call(_lib.lib().sf_dit_forward_pair, C.byref(m.cmodel), C.byref(a0), C.byref(a1), _stream(noisy)), "sf_dit_forward_pair")
    return flow, x0


@dit_forward_pair.register_fake
def _(model, ctx_noisy, ctx_timestep, noisy, timestep, k_cache, v_cache, ck_cache, cv_cache, workspace, evict_scratch, ctx_plan, plan,
      kv_index, global_end):
    B, F, _, H, W = noisy.shape
    out_dim = _model(model).shape.out_dim
    return noisy.new_empty((B, F, out_dim, H, W)), noisy.new_empty((B, F, out_dim, H, W))


@dit_forward.register_fake
def _(model, noisy, timestep, prompt_embeds, add_condition, k_cache, v_cache, ck_cache, cv_cache, workspace, evict_scratch,
      init_cross, cache_only, sink, evict, keep, write_start, attn_start, attn_end, start_frame, kv_index, global_end):
    if cache_only:
        return noisy.new_empty((0,)), noisy.new_empty((0,))
    B, F, _, H, W = noisy.shape
    out_dim = _model(model).shape.out_dim
    return noisy.new_empty((B, F, out_dim, H, W)), noisy.new_empty((B, F, out_dim, H, W))


# ------------------------------------------------------------------------------------------ VAE decode / T5 encode
@custom_op(f"{NAMESPACE}::vae_decode_frames", mutates_args=("state", "scratch", "out"))
def vae_decode_frames(model: int, state: Tensor, scratch: Tensor, z: Tensor, out: Tensor, h: int, w: int, window_frames: int,
                      frame_index: int, window: int, history_at: int) -> None:
    """Consecutive latent frames z [F, z_dim, h, w] -> 1 (frame_index 0: F = 1) or 4 F pixel frames written to the front
    of `out` (float32 [T, 3, 8h, 8w]); `state` carries every convolution's two-frame history between calls; `frame_index`
    counts the latent frames decoded into it since its reset, `window` / `history_at` place the sliding history windows
    (sf_vae_decode_frames in include/sf_hip.h; `WanVAEDecoder.cached_decode` does the bookkeeping)."""
    m = _model(model)
    _need_gpu(z, "z")
    _need_gpu(out, "out", torch.float32)
    if z.dim() != 4 or not z.is_contiguous() or not out.is_contiguous():
        raise ValueError("vae_decode_frames: contiguous z [F, z_dim, h, w] and out expected")
    if tuple(z.shape[1:]) != (m.shape.z_dim, h, w):
        raise ValueError(f"vae_decode_frames: z must be [F, {m.shape.z_dim}, {h}, {w}], got {tuple(z.shape)}")
    for name, t in (("state", state), ("scratch", scratch)):       # their element counts are passed on as BYTE counts
        if not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous():
            raise ValueError(f"vae_decode_frames: {name} must be a contiguous CUDA uint8 tensor")
    sf_, tf_ = m.shape.spatial_factor, m.shape.temporal_factor
    frames = 1 if frame_index == 0 else tf_ * z.shape[0]
    if frame_index == 0 and z.shape[0] != 1:
        raise ValueError("vae_decode_frames: the frame that follows a reset (frame_index 0) is decoded alone")
    if out.numel() < frames * 3 * (sf_ * h) * (sf_ * w):
        raise ValueError(f"vae_decode_frames: out holds {out.numel()} floats, {frames} frames of 3 x {sf_ * h} x {sf_ * w} need "
                         f"{frames * 3 * sf_ * h * sf_ * w}")
    _lib.check(_lib.lib().sf_vae_decode_frames(C.byref(m.cmodel), state.data_ptr(), state.numel(), scratch.data_ptr(), scratch.numel(),
                                               z.data_ptr(), h, w, window_frames, frame_index, z.shape[0], window, history_at,
                                               out.data_ptr(), _stream(z)),
               "sf_vae_decode_frames")


@custom_op(f"{NAMESPACE}::t5_encode", mutates_args=("workspace",))
def t5_encode(model: int, ids: Tensor, mask: Tensor, buckets: Tensor, workspace: Tensor) -> Tensor:
    """umT5 encoder pass: ids, mask int64 [B, L] -> bf16 [B, L, dim], rows past each prompt's length zeroed (sf_t5_encode)."""
    m = _model(model)
    _need_gpu(ids, "ids", torch.int64), _need_gpu(mask, "mask", torch.int64), _need_gpu(buckets, "buckets", torch.int32)
    if ids.dim() != 2 or ids.shape != mask.shape or not ids.is_contiguous() or not mask.is_contiguous():
        raise ValueError("t5_encode: ids and mask must be contiguous [B, L]")
    B, L = ids.shape
    out = torch.empty(B, L, m.shape.dim, dtype=torch.bfloat16, device=ids.device)
    _lib.check(_lib.lib().sf_t5_encode(C.byref(m.cmodel), ids.data_ptr(), mask.data_ptr(), buckets.data_ptr(), B, L, out.data_ptr(),
                                       workspace.data_ptr(), workspace.numel(), _stream(ids)), "sf_t5_encode")
    return out


@t5_encode.register_fake
def _(model, ids, mask, buckets, workspace):
    return ids.new_empty((ids.shape[0], ids.shape[1], _model(model).shape.dim), dtype=torch.bfloat16)


OPS = ("attention", "gemm", "gemm_out", "lincomb", "lincomb_out", "add_noise", "dit_forward", "dit_forward_pair", "vae_decode_frames", "t5_encode")
