// Measured ceilings of THIS box ("re-measure on the box (a peak-MFMA micro-kernel and a streaming-copy
// kernel) and use the measured ceilings in the fraction").  Two measurement entry points, no product kernel calls them:
//   sf_probe_mfma -- a register-only bf16 MFMA loop (no LDS, no memory) on every SIMD of the chip, random operands
//                    (the clock the chip holds depends on the data: zeros run 19 % faster than random bf16);
//   sf_probe_copy -- a 16-byte-per-lane streaming copy src -> dst.
// bench.py times them under sustained clocks and reports roofline.measured_peak / hbm_measured_peak beside the
// datasheet constants.
#include "sf_common.h"
#include "../../include/sf_hip.h"

template <int SHAPE>   // 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_16x16x32_bf16; 4 independent accumulators
__global__ __launch_bounds__(256) void probe_mfma_kernel(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  const bf16x8 a = in[tid], b = in[256 + tid];
  f32x16 acc[4];
  f32x4 acc4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
  }
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 64; ++i) acc4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i & 3], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) s += acc4[i][r];
  }
  out[blockIdx.x * 256 + tid] = s;
}

// 4 x 16 bytes per lane in flight per iteration (loads first, then stores), grid-stride over whole 16 KiB pieces per workgroup
__global__ __launch_bounds__(256) void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * 1024;
  int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  for (; i + 768 < n16; i += stride) {
    const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256);
    const u32x4 c = __builtin_nontemporal_load(src + i + 512), d = __builtin_nontemporal_load(src + i + 768);
    __builtin_nontemporal_store(a, dst + i);
    __builtin_nontemporal_store(b, dst + i + 256);
    __builtin_nontemporal_store(c, dst + i + 512);
    __builtin_nontemporal_store(d, dst + i + 768);
  }
  for (; i < n16; i += 256) dst[i] = src[i];     // this lane's share of the ragged last piece (at most 3 elements)
}

extern "C" int sf_probe_mfma(int shape, int iters, int workgroups, const void* operands, float* sink, double* flops_out, void* stream) {
  SF_CHECK(operands && sink, "sf_probe_mfma: null operand / sink pointer");
  SF_CHECK((shape == 0 || shape == 1) && iters > 0 && workgroups > 0 && workgroups <= 65536, "sf_probe_mfma: bad shape %d / iters %d / workgroups %d",
           shape, iters, workgroups);
  hipStream_t st = (hipStream_t)stream;
  if (shape == 0)
    hipLaunchKernelGGL(probe_mfma_kernel<0>, dim3(workgroups), dim3(256), 0, st, (const bf16x8*)operands, sink, iters);
  else
    hipLaunchKernelGGL(probe_mfma_kernel<1>, dim3(workgroups), dim3(256), 0, st, (const bf16x8*)operands, sink, iters);
  SF_HIP_LAUNCH_CHECK("sf_probe_mfma");
  // 4 waves per workgroup; per iteration 32 x (32x32x16) or 64 x (16x16x32) MFMAs = 32 x 32768 flop either way
  if (flops_out) *flops_out = (double)workgroups * 4.0 * (double)iters * 32.0 * 32768.0;
  return 0;
}

extern "C" int sf_probe_copy(const void* src, void* dst, size_t bytes, void* stream) {
  SF_CHECK(src && dst && bytes >= 16 && bytes % 16 == 0, "sf_probe_copy: null pointer or byte count %zu not a multiple of 16", bytes);
  SF_CHECK(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "sf_probe_copy: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, (int64_t)(bytes / 16));
  SF_HIP_LAUNCH_CHECK("sf_probe_copy");
  return 0;
}
