// Shared device/host helpers for the gfx950 kernels of the Self-Forcing hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SF_WAVE 64

// ---- error plumbing (thread-local last-error string, see capi.cpp)
extern "C" void sf_set_error(const char* fmt, ...);
#define SF_CHECK(cond, ...)                  \
  do {                                       \
    if (!(cond)) {                           \
      sf_set_error(__VA_ARGS__);             \
      return -1;                             \
    }                                        \
  } while (0)
#define SF_HIP_LAUNCH_CHECK(name)                                              \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      sf_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return -2;                                                               \
    }                                                                          \
  } while (0)

#ifdef __HIPCC__
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

// Wave-wide reductions by DPP (six VALU instructions with the lane movement built in, then one v_readlane broadcast)
// instead of six __shfl_xor steps: hipcc lowers those to ds_bpermute_b32 -- a round trip through the LDS crossbar each,
// ~100 cycles of latency, six of them dependent per reduction -- which is most of what a one-pass norm kernel does
// between its loads and its stores.  All 64 lanes must be active.  (The order of the fp32 additions differs from the
// butterfly's; every caller rounds the result into bf16 outputs.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float sf_dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += sf_dpp<0xB1, 0xF>(0.f, v);     // quad_perm [1,0,3,2]
  v += sf_dpp<0x4E, 0xF>(0.f, v);     // quad_perm [2,3,0,1]
  v += sf_dpp<0x141, 0xF>(0.f, v);    // row_half_mirror
  v += sf_dpp<0x140, 0xF>(0.f, v);    // row_mirror: every lane of a row of 16 holds that row's sum
  v += sf_dpp<0x142, 0xA>(0.f, v);    // row_bcast15: rows 1 and 3 add lane 15 of the row before
  v += sf_dpp<0x143, 0xC>(0.f, v);    // row_bcast31: rows 2 and 3 add lane 31 -> lanes 48..63 hold the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, sf_dpp<0xB1, 0xF>(v, v));
  v = fmaxf(v, sf_dpp<0x4E, 0xF>(v, v));
  v = fmaxf(v, sf_dpp<0x141, 0xF>(v, v));
  v = fmaxf(v, sf_dpp<0x140, 0xF>(v, v));
  v = fmaxf(v, sf_dpp<0x142, 0xA>(v, v));
  v = fmaxf(v, sf_dpp<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// tanh-approximated GELU, nn.GELU(approximate='tanh'):  0.5 x (1 + tanh(u)), u = k0 (x + k1 x^3)
//   = x * sigmoid(2u) = x / (1 + exp2(-2 u log2e)): one v_exp_f32 and one v_rcp_f32 instead of an exp and a full-
// precision division (7 instructions per element; the epilogue of ffn.0 runs it 42 M times per GEMM).
// exp2 overflow -> inf -> rcp 0 -> x*0; underflow -> 0 -> x.
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float c0 = 2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
  const float u2 = x * (c0 + c1 * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u2));
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// x * sigmoid(x) with one v_exp_f32 and one v_rcp_f32 (no full-precision division: ~6 instructions instead of ~16); used
// where the result is rounded to bf16 right away (the VAE's RMS_norm + SiLU passes, fused or not)
__device__ __forceinline__ float silu_fast_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * x));
}
#endif
