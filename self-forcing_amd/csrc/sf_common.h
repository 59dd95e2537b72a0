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

// tanh-approximated GELU, nn.GELU(approximate='tanh')
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  // tanh(u) = 1 - 2 / (1 + exp(2u)); exp overflow -> inf -> tanh = 1, underflow -> -1
  float t = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
  return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
#endif
