// Small-M linear layer for gfx950: out[M,N] = act_out(act_in(x)[M,K] @ w[N,K]^T + bias), M <= 32.
//
// Replaces the time-embedding MLPs of the reference (wan/modules/causal_model.py:464-467,
// :829-832: Linear(256,C)-SiLU-Linear(C,C) and SiLU-Linear(C,6C)) where M = batch*frames is 1..21:
// pure weight streaming (28 MB for the 1.3B time_projection), so no MFMA, no LDS round trip for the
// weights: every wave streams whole weight rows with 16-byte loads straight to VGPRs, the M
// activation rows sit in LDS (bf16), partial dot products are reduced with wave shuffles.
#include "sf_common.h"
#include "../../include/sf_hip.h"

namespace {

constexpr int SL_MB_MAX = 8;    // activation rows per pass (the kernel is instantiated for 1, 2, 3, 4 and 8 rows: with the
                                // rollout's M = 3 an 8-row pass spent 2.7x the FMAs and wave reductions the rows need and
                                // ran VALU-bound at 1.3 TB/s of weights)
constexpr int SL_THREADS = 256;
constexpr int SL_NPW = 4;       // output columns per wave per iteration (independent 16-byte loads in flight)

__device__ __forceinline__ float apply_act(float v, int act) {
  return act == 1 ? silu_f(v) : (act == 2 ? gelu_tanh_f(v) : v);
}

template <int SL_MB>
__global__ __launch_bounds__(SL_THREADS) void small_linear_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                  const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                                                  int m0, int mcount, int N, int K, int act_in, int act_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);  // [SL_MB][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < SL_MB * K; i += SL_THREADS) {
    const int m = i / K, k = i - m * K;
    float v = 0.f;
    if (m < mcount) v = apply_act((float)x[(long)(m0 + m) * K + k], act_in);
    xs[i] = (bf16_t)v;
  }
  __syncthreads();
  const int waves_total = gridDim.x * (SL_THREADS / 64);
  const int wave_id = blockIdx.x * (SL_THREADS / 64) + wave;
  for (int n0 = wave_id * SL_NPW; n0 < N; n0 += waves_total * SL_NPW) {
    float acc[SL_NPW][SL_MB];
#pragma unroll
    for (int c = 0; c < SL_NPW; ++c)
#pragma unroll
      for (int m = 0; m < SL_MB; ++m) acc[c][m] = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
      bf16x8 wv[SL_NPW];
#pragma unroll
      for (int c = 0; c < SL_NPW; ++c) {
        const int n = min(n0 + c, N - 1);
        wv[c] = *reinterpret_cast<const bf16x8*>(w + (long)n * K + k);
      }
#pragma unroll
      for (int m = 0; m < SL_MB; ++m) {
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(xs + m * K + k);
#pragma unroll
        for (int c = 0; c < SL_NPW; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[c][m] += (float)xv[j] * (float)wv[c][j];
      }
    }
#pragma unroll
    for (int c = 0; c < SL_NPW; ++c)
#pragma unroll
      for (int m = 0; m < SL_MB; ++m) acc[c][m] = wave_sum(acc[c][m]);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < SL_NPW; ++c) {
        const int n = n0 + c;
        if (n >= N) continue;
        const float b = bias ? (float)bias[n] : 0.f;
        for (int m = 0; m < mcount; ++m)
          out[(long)(m0 + m) * N + n] = (bf16_t)apply_act(acc[c][m] + b, act_out);
      }
    }
  }
}

}  // namespace

extern "C" int sf_small_linear(const void* x, const void* w, const void* bias, void* out, int M, int N, int K,
                               int act_in, int act_out, void* stream) {
  SF_CHECK(x && w && out, "sf_small_linear: null tensor");
  SF_CHECK(M > 0 && M <= 32 && N > 0 && K > 0 && K % 8 == 0, "sf_small_linear: unsupported shape M=%d N=%d K=%d (M<=32, K%%8==0)", M, N, K);
  SF_CHECK((size_t)SL_MB_MAX * K * 2 <= 160 * 1024, "sf_small_linear: K=%d too large for the LDS activation stage", K);
  SF_CHECK(act_in >= 0 && act_in <= 2 && act_out >= 0 && act_out <= 2, "sf_small_linear: bad activation code");
  const int waves_needed = (N + SL_NPW - 1) / SL_NPW;
  const int blocks = min(1024, (waves_needed + 3) / 4);
  for (int m0 = 0; m0 < M; m0 += SL_MB_MAX) {
    const int mc = min(SL_MB_MAX, M - m0);
#define SF_SL_LAUNCH(R)                                                                                                  \
  hipLaunchKernelGGL(small_linear_kernel<R>, dim3(blocks), dim3(SL_THREADS), (size_t)R * K * 2, (hipStream_t)stream,      \
                     (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, (bf16_t*)out, m0, mc, N, K, act_in, act_out)
    if (mc == 1) SF_SL_LAUNCH(1);
    else if (mc == 2) SF_SL_LAUNCH(2);
    else if (mc == 3) SF_SL_LAUNCH(3);
    else if (mc == 4) SF_SL_LAUNCH(4);
    else SF_SL_LAUNCH(8);
#undef SF_SL_LAUNCH
  }
  SF_HIP_LAUNCH_CHECK("sf_small_linear");
  return 0;
}
