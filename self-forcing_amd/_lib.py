"""ctypes binding of the C-ABI in include/sf_hip.h (self-forcing_amd/csrc/libsf_hip.so).

There is no fallback: if the library is missing, loading raises with the build command.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

# A default for the HIP runtime, set before anything has initialised it (it reads the variable once, at start-up; a value
# the user has set wins): with the runtime's default signal pool a steady stream of ~430 launches per forward keeps one of
# its helper threads spinning for signals to come free -- 0.83 of a host core per process, measured on MI355X / ROCm 7.0
# (bench.py: host_busy_cores_by_thread); with 256 or more signals in the pool that thread idles (0.004).  Together with
# the wrapper's pacing (wan_wrapper.py) a rank needs 0.17 of a core instead of 1.0-1.7; throughput is unchanged.
# (The C library itself reads no environment variables; this is the Python host layer configuring the runtime under it.)
os.environ.setdefault("ROC_SIGNAL_POOL_SIZE", "1024")

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("SF_HIP_LIB") or os.path.join(CSRC, "libsf_hip.so")   # SF_HIP_LIB: alternate builds (kernel ablation timing)

ABI_VERSION = 8

# epilogue codes (enum sf_epilogue)
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_BIAS_GATE_RESID, EPI_F32 = 0, 1, 2, 3, 4
# enum sf_conv_epilogue
CONV_BIAS, CONV_BIAS_RESID, CONV_BIAS_CLAMP_F32 = 0, 1, 2
VAE_MAX_STAGES = 4
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
# enum sf_attn_structure / sf_gemm_structure
ATTN_STRUCTURES = {"auto": 0, "r64": 1, "w8": 2, "w4": 3}
CONV_STRUCTURES = {"auto": 0, "igemm": 1, "halo": 2}
GEMM_STRUCTURES = {"auto": 0, "t128": 1, "pp256": 2, "pp128": 3, "pp224": 4, "pp192": 5}


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
        ("resid", C.c_void_p), ("gate_mod", C.c_void_p), ("gate_e0", C.c_void_p),
        ("gate_group_stride", C.c_int64), ("rows_per_group", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldo", C.c_int32), ("ldr", C.c_int32),
        ("epilogue", C.c_int32), ("batch", C.c_int32),
        ("a_bstride", C.c_int64), ("w_bstride", C.c_int64), ("o_bstride", C.c_int64), ("r_bstride", C.c_int64),
        ("structure", C.c_int32),
    ]


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "modulation", "norm3_w", "norm3_b", "qkv_w", "qkv_b", "norm_q_w", "norm_k_w", "o_w", "o_b",
        "cq_w", "cq_b", "ckv_w", "ckv_b", "cnorm_q_w", "cnorm_k_w", "co_w", "co_b",
        "ffn0_w", "ffn0_b", "ffn2_w", "ffn2_b")]


class Model(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("dim", "ffn_dim", "num_heads", "num_layers", "in_dim", "out_dim",
                                  "freq_dim", "text_dim", "text_len")]
        + [("eps", C.c_float)]
        + [(n, C.c_void_p) for n in ("patch_w", "patch_b", "text0_w", "text0_b", "text2_w", "text2_b",
                                     "time0_w", "time0_b", "time2_w", "time2_b", "tproj_w", "tproj_b",
                                     "head_w", "head_b", "head_mod", "pose_w", "pose_b")]
        + [("pose_dim", C.c_int32)]
        + [("layers_host", C.POINTER(LayerWeights)),
           ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
           ("sched_sigmas", C.c_void_p), ("sched_timesteps", C.c_void_p),
           ("n_table", C.c_int32)]
    )


class ForwardArgs(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("frames", C.c_int32), ("lat_h", C.c_int32), ("lat_w", C.c_int32),
        ("groups", C.c_int32),
        ("noisy", C.c_void_p), ("timestep", C.c_void_p), ("t_is_int64", C.c_int32),
        ("prompt_embeds", C.c_void_p), ("init_cross", C.c_int32), ("add_condition", C.c_void_p),
        ("k_cache_host", C.POINTER(C.c_void_p)), ("v_cache_host", C.POINTER(C.c_void_p)),
        ("ck_cache_host", C.POINTER(C.c_void_p)), ("cv_cache_host", C.POINTER(C.c_void_p)),
        ("cache_tokens", C.c_int64),
        ("sink_tokens", C.c_int32), ("evict", C.c_int32), ("keep", C.c_int32),
        ("write_start", C.c_int32), ("attn_start", C.c_int32), ("attn_end", C.c_int32),
        ("start_frame", C.c_int32),
        ("evict_scratch", C.c_void_p), ("evict_scratch_bytes", C.c_size_t),
        ("cache_only", C.c_int32),
        ("flow_out", C.c_void_p), ("x0_out", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("kv_index_out", C.c_void_p), ("global_end", C.c_int64),
    ]


class ConvArgs(C.Structure):
    _fields_ = (
        [(n, C.c_void_p) for n in ("x", "w", "bias", "out", "resid", "out_f32")]
        + [(n, C.c_int32) for n in ("Tout", "H", "W", "Hin", "Win", "Cin", "Cout", "kt", "kh", "kw", "upsample",
                                    "t_in_offset", "ldw", "ldo", "ldr", "out_frame_offset", "interleave_c", "epilogue", "structure")]
        + [("norm_out", C.c_void_p), ("norm_gamma", C.c_void_p), ("norm_ld", C.c_int32), ("norm_frame_offset", C.c_int32)]
    )


class VaeConv(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p)] + [(n, C.c_int32) for n in ("cin", "cout", "kt", "kh", "kw", "ldw")]


class VaeResBlock(C.Structure):
    _fields_ = [("gamma1", C.c_void_p), ("gamma2", C.c_void_p), ("conv1", VaeConv), ("conv2", VaeConv), ("shortcut", VaeConv)]


class VaeModel(C.Structure):
    _fields_ = [
        ("z_dim", C.c_int32), ("n_stages", C.c_int32), ("res_per_stage", C.c_int32),
        ("temporal_up", C.c_int32 * VAE_MAX_STAGES),
        ("latent_mean", C.c_void_p), ("latent_std", C.c_void_p), ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p),
        ("conv1", VaeConv), ("mid0", VaeResBlock), ("mid2", VaeResBlock),
        ("attn_gamma", C.c_void_p), ("attn_qk_w", C.c_void_p), ("attn_qk_b", C.c_void_p),
        ("attn_v_w", C.c_void_p), ("attn_v_b", C.c_void_p), ("attn_proj_w", C.c_void_p), ("attn_proj_b", C.c_void_p),
        ("res_host", C.POINTER(VaeResBlock)),
        ("time_conv", VaeConv * VAE_MAX_STAGES), ("up_conv", VaeConv * VAE_MAX_STAGES),
        ("head_gamma", C.c_void_p), ("head_conv", VaeConv),
    ]


class T5Layer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm1_w", "qk_w", "v_w", "o_w", "norm2_w", "gate_w", "fc1_w", "fc2_w", "pos_emb")]


class T5Model(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("vocab", "dim", "dim_attn", "dim_ffn", "num_heads", "num_layers", "num_buckets")]
                + [("eps", C.c_float), ("token_embedding", C.c_void_p), ("layers_host", C.POINTER(T5Layer)),
                   ("final_norm_w", C.c_void_p)])


# name -> (restype, argtypes); every symbol include/sf_hip.h declares
_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    "sf_abi_version": (C.c_int, []),
    "sf_last_error": (C.c_char_p, []),
    "sf_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), _vp]),
    "sf_small_linear": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sf_sinusoid_embedding": (C.c_int, [_vp, _i, _vp, _i, _i, _vp]),
    "sf_layernorm_modulate": (C.c_int, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "sf_layernorm_affine": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "sf_rmsnorm": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _vp]),
    "sf_qkv_norm_rope_cache": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i64,
                                         _i, _i, _f, _vp]),
    "sf_kv_evict": (C.c_int, [_vp, _i, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sf_attention": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "sf_attention_ex": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i, _vp]),
    "sf_patchify": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sf_unpatchify_x0": (C.c_int, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sf_add_noise": (C.c_int, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i64, _vp]),
    "sf_lincomb_bf16": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_float), _i, _i64, _vp]),
    "sf_dit_workspace_bytes": (C.c_size_t, [C.POINTER(Model), _i, _i, _i, _i, _i]),
    "sf_dit_forward": (C.c_int, [C.POINTER(Model), C.POINTER(ForwardArgs), _vp]),
    "sf_dit_forward_pair": (C.c_int, [C.POINTER(Model), C.POINTER(ForwardArgs), C.POINTER(ForwardArgs), _vp]),
    "sf_conv_igemm": (C.c_int, [C.POINTER(ConvArgs), _vp]),
    "sf_conv_pick_nt": (C.c_int, [_i]),
    "sf_rmsnorm_silu_cl": (C.c_int, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "sf_softmax_rows": (C.c_int, [_vp, _i64, _vp, _i64, _i, _i, _i, _f, _vp]),
    "sf_vae_prepare_latent": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sf_vae_state_bytes": (C.c_size_t, [C.POINTER(VaeModel), _i, _i, _i]),
    "sf_vae_scratch_bytes": (C.c_size_t, [C.POINTER(VaeModel), _i, _i, _i]),
    "sf_vae_reset": (C.c_int, [C.POINTER(VaeModel), _vp, _sz, _i, _i, _i, _vp]),
    "sf_vae_decode_frames": (C.c_int, [C.POINTER(VaeModel), _vp, _sz, _vp, _sz, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sf_embedding_gather": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "sf_t5_softmax_bias": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sf_mul_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "sf_zero_masked_rows": (C.c_int, [_vp, _vp, _i, _i, _vp]),
    "sf_t5_workspace_bytes": (C.c_size_t, [C.POINTER(T5Model), _i, _i]),
    "sf_t5_encode": (C.c_int, [C.POINTER(T5Model), _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "sf_probe_mfma": (C.c_int, [_i, _i, _i, _vp, _vp, C.POINTER(C.c_double), _vp]),
    "sf_probe_copy": (C.c_int, [_vp, _vp, _sz, _vp]),
}

_lib: Optional[C.CDLL] = None


class SfHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into csrc/libsf_hip.so (hipcc cross-compiles without
    a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise SfHipError(f"building {LIB_PATH} failed (exit {res.returncode})")
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded C-ABI library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SfHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C {CSRC}`). "
            "There is no CPU/eager fallback for this path.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    ver = handle.sf_abi_version()
    if ver != ABI_VERSION:
        raise SfHipError(f"{LIB_PATH}: ABI version {ver}, expected {ABI_VERSION}; rebuild")
    _lib = handle
    return handle


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().sf_last_error()
        raise SfHipError(f"{what or 'sf_hip'} failed (rc={rc}): {msg.decode() if msg else '?'}")
