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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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
