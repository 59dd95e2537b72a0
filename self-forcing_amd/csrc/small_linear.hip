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

// ------------------------------------------------------------------------------------------
// Register-resident form for the rollout's shapes (M <= 8 rows, K <= 1536): the kernel is a pure weight stream (28 MB for
// time_projection at 1.3B), so the ONLY thing that matters is how early and how many bytes are in flight:
//   * a wave requests ALL its weights first -- 4 output columns x KS steps x 16 bytes per lane (12 loads, 48 VGPRs at
//     K = 1536); with 9 waves per CU that is 110 KB in flight per CU from the first cycle on.  The LDS form above staged
//     the activations first (scalar loads + SiLU + a workgroup barrier: ~2 us before the first weight byte was asked for);
//   * the M activation rows are 9 KB shared by every wave: read straight from L2 into registers (packed bf16, activation
//     applied and rounded to bf16 exactly as the LDS form stores them), no LDS, no barrier;
//   * products by v_dot2c_f32_bf16 (two bf16 products + fp32 accumulate per instruction: a quarter of the VALU work of
//     convert + FMA), wave reduction by six DPP adds per value (sf_common.h) instead of six ds_bpermute round trips.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

constexpr int SLR_MAX_THREADS = 640;   // up to 10 waves x 4 columns per workgroup; the host picks the wave count so that the
                                       // grid is ONE workgroup per CU where it can (6 x 1536 columns: 9 waves x 256 workgroups)

template <int MB, int KS, int act_in>   // act_in compile-time: straight-line code, so the waits are counted (vmcnt(12)), not drained
__global__ __launch_bounds__(SLR_MAX_THREADS) void small_linear_reg_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                           const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                                                           int N, int K, int act_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [MB][K] bf16, only when act_in != 0
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int n0 = (blockIdx.x * (nthreads >> 6) + wave) * SL_NPW;
  // 0. with an input activation: this thread's share of the activation rows, requested BEFORE the weights -- vmcnt
  // retires in order, so the (L2-resident) rows can be waited for while the twelve weight loads behind them are in flight
  constexpr int XCH = (MB * KS * 64 + 255) / 256;                 // chunks of 8 per thread at >= 256 threads
  bf16x8 xraw[XCH];
  const int chunks = MB * K / 8;
  if (act_in != 0) {
#pragma unroll
    for (int i = 0; i < XCH; ++i)      // (addresses clamped, never predicated: branch-free code keeps the waits counted)
      xraw[i] = *reinterpret_cast<const bf16x8*>(x + (long)min(tid + i * nthreads, chunks - 1) * 8);
    __builtin_amdgcn_sched_barrier(0);   // (the rows' requests stay in FRONT of the weights': vmcnt retires in order)
  }
  // 1. every weight byte this wave needs
  bf16x8 wv[SL_NPW][KS];
#pragma unroll
  for (int c = 0; c < SL_NPW; ++c) {
    const bf16_t* wrow = w + (long)min(n0 + c, N - 1) * K;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = min(s * 512 + lane * 8, K - 8);      // (lanes past K re-read the last chunk; their activations are zero)
      wv[c][s] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wrow + k));
    }
  }
  // 2. the activation rows.  With an input activation (SiLU in front of time_projection) the workgroup applies it ONCE,
  // cooperatively, and shares the rounded bf16 values through LDS: done per wave it was ~20 VALU instructions on 72
  // elements per lane -- more time than the weight stream takes.  Without one the rows come straight from L2.
  bf16x8 xv[MB][KS];
  if (act_in != 0) {
    bf16x8* xs = reinterpret_cast<bf16x8*>(smem);
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      bf16x8 t = xraw[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = (bf16_t)apply_act((float)t[j], act_in);
      xs[tid + i * nthreads] = t;      // (the LDS stage is sized XCH x nthreads chunks: slots past `chunks` are padding)
    }
    // raw barrier: __syncthreads() carries a fence that drains vmcnt -- the weight loads must stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = s * 512 + lane * 8;
        const bf16x8 t = xs[(m * K + min(k, K - 8)) / 8];
        xv[m][s] = k < K ? t : bf16x8{};
      }
  } else {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = s * 512 + lane * 8;
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + (long)m * K + min(k, K - 8));
        xv[m][s] = k < K ? t : bf16x8{};
      }
  }
  __builtin_amdgcn_sched_barrier(0);   // (every load above stays above: nothing below may be scheduled in front of a request)
  // 3. dot products (fp32 accumulation), 4. wave reduction, lane 0 writes.  Waves past N (last workgroup only) run on
  // clamped rows and store nothing: an early return would let the compiler sink the weight loads behind it.
  float acc[SL_NPW][MB];
#pragma unroll
  for (int c = 0; c < SL_NPW; ++c)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x2_t xa = {xv[m][s][2 * j], xv[m][s][2 * j + 1]}, wa = {wv[c][s][2 * j], wv[c][s][2 * j + 1]};
          a = __builtin_amdgcn_fdot2_f32_bf16(xa, wa, a, false);
        }
      acc[c][m] = wave_sum(a);       // (DPP adds + one readlane: sf_common.h)
    }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < SL_NPW; ++c) {
      const int n = n0 + c;
      if (n >= N) continue;
      const float b = bias ? (float)bias[n] : 0.f;
#pragma unroll
      for (int m = 0; m < MB; ++m) out[(long)m * N + n] = (bf16_t)apply_act(acc[c][m] + b, act_out);
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
  const int ks = (K + 511) / 512;
  if (M <= 8 && (ks == 1 || ks == 3) && act_in <= 1) {   // the rollout's shapes: K = freq_dim 256 and K = dim 1536, M = batch x groups
    // waves per workgroup: enough that 256 workgroups cover N (one per CU, no second round on a few CUs), 4 .. 10
    const int wpb = min(SLR_MAX_THREADS / 64, max(4, (waves_needed + 255) / 256));
    const dim3 grid((waves_needed + wpb - 1) / wpb), block(64 * wpb);
    const int xch = (M * ks * 64 + 255) / 256;                               // = XCH of the instantiation
    const size_t lds = act_in != 0 ? (size_t)xch * 64 * wpb * 16 : 0;        // chunks of 16 bytes, padded to whole passes
#define SF_SLR(R, S)                                                                                                                  \
  do {                                                                                                                                \
    if (act_in == 0)                                                                                                                  \
      hipLaunchKernelGGL((small_linear_reg_kernel<R, S, 0>), grid, block, lds, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, \
                         (const bf16_t*)bias, (bf16_t*)out, N, K, act_out);                                                           \
    else                                                                                                                              \
      hipLaunchKernelGGL((small_linear_reg_kernel<R, S, 1>), grid, block, lds, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, \
                         (const bf16_t*)bias, (bf16_t*)out, N, K, act_out);                                                           \
  } while (0)
#define SF_SLR_M(S)                                                                                                                   \
  switch (M) { case 1: SF_SLR(1, S); break; case 2: SF_SLR(2, S); break; case 3: SF_SLR(3, S); break; case 4: SF_SLR(4, S); break; \
               case 5: SF_SLR(5, S); break; case 6: SF_SLR(6, S); break; case 7: SF_SLR(7, S); break; default: SF_SLR(8, S); break; }
    if (ks == 1) { SF_SLR_M(1) } else { SF_SLR_M(3) }
#undef SF_SLR_M
#undef SF_SLR
    SF_HIP_LAUNCH_CHECK("sf_small_linear");
    return 0;
  }
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
