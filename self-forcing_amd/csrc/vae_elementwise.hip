// HBM-bound kernels of the VAE decode path (gfx950): channel RMS-norm (+SiLU), the row softmax of the
// single-head attention block, and the latent-frame preparation.  Channels-last bf16 rows throughout.
#include "sf_common.h"
#include "../../include/sf_hip.h"

namespace {

// RMS_norm (vae.py:41-56) + optional SiLU.  A row of C channels is shared by LPR lanes (the power of
// two >= C/8), each holding one 16-byte chunk; the reduction is LPR-wide xor-shuffles.
template <int LPR>
__global__ __launch_bounds__(256) void rmsnorm_silu_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma,
                                                           bf16_t* __restrict__ out, long rows, int C, int silu) {
  constexpr int RPB = 256 / LPR;   // rows per block
  const int tid = threadIdx.x;
  const int sub = tid % LPR;
  const long row = (long)blockIdx.x * RPB + tid / LPR;
  const bool active = row < rows && sub * 8 < C;
  float v[8];
  float ss = 0.f;
  if (active) {
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(x + row * C + sub * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = (float)d[j]; ss += v[j] * v[j]; }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (!active) return;
  const float inv = sqrtf((float)C) / fmaxf(sqrtf(ss), 1e-12f);
  const bf16x8 g = *reinterpret_cast<const bf16x8*>(gamma + sub * 8);
  bf16x8 o8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float y = v[j] * inv * (float)g[j];
    if (silu) y = silu_fast_f(y);
    o8[j] = (bf16_t)y;
  }
  *reinterpret_cast<bf16x8*>(out + row * C + sub * 8) = o8;
}

// p[r][c] = softmax_c(scale * s[r][c]); one 256-thread block per row, the row is read twice from L2
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, long lds, bf16_t* __restrict__ p, long ldp,
                                                           int cols, int cols_padded, float scale) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sr = s + (long)blockIdx.x * lds;
  bf16_t* pr = p + (long)blockIdx.x * ldp;
  float mx = -3.0e38f;
  for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, sr[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid; c < cols; c += 256) sum += __expf((sr[c] - mx) * scale);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = tid; c < cols_padded; c += 256) pr[c] = c < cols ? (bf16_t)(__expf((sr[c] - mx) * scale) * inv) : (bf16_t)0.f;
}

// one thread per latent position: u = z * std + mean (fp32), y = conv2_w u + conv2_b, channels-last out
__global__ __launch_bounds__(256) void prepare_latent_kernel(const bf16_t* __restrict__ lat, const float* __restrict__ mean,
                                                             const float* __restrict__ stdv, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ b, bf16_t* __restrict__ out, int z, int hw, int c_pad) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= hw) return;
  float u[32];
  for (int c = 0; c < z; ++c) u[c] = (float)lat[(long)c * hw + pos] * stdv[c] + mean[c];
  bf16_t* o = out + (long)pos * c_pad;
  for (int n = 0; n < z; ++n) {
    float acc = (float)b[n];
    for (int c = 0; c < z; ++c) acc += (float)w[n * z + c] * u[c];
    o[n] = (bf16_t)acc;
  }
  for (int n = z; n < c_pad; ++n) o[n] = (bf16_t)0.f;
}

}  // namespace

extern "C" int sf_rmsnorm_silu_cl(const void* x, const void* gamma, void* out, int64_t rows, int C, int silu, void* stream) {
  SF_CHECK(x && gamma && out, "sf_rmsnorm_silu_cl: null tensor");
  SF_CHECK(rows > 0 && C > 0 && C % 8 == 0 && C <= 512, "sf_rmsnorm_silu_cl: rows=%lld C=%d (need C %% 8 == 0, C <= 512)", (long long)rows, C);
  SF_CHECK(((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)gamma % 16 == 0), "sf_rmsnorm_silu_cl: misaligned tensor");
  hipStream_t s = (hipStream_t)stream;
  int lpr = 1;
  while (lpr * 8 < C) lpr *= 2;
  const int rpb = 256 / lpr;
  const dim3 grid((unsigned)((rows + rpb - 1) / rpb)), block(256);
  const bf16_t* xp = (const bf16_t*)x; const bf16_t* gp = (const bf16_t*)gamma; bf16_t* op = (bf16_t*)out;
  switch (lpr) {
    case 1: hipLaunchKernelGGL(rmsnorm_silu_kernel<1>, grid, block, 0, s, xp, gp, op, (long)rows, C, silu); break;
    case 2: hipLaunchKernelGGL(rmsnorm_silu_kernel<2>, grid, block, 0, s, xp, gp, op, (long)rows, C, silu); break;
    case 4: hipLaunchKernelGGL(rmsnorm_silu_kernel<4>, grid, block, 0, s, xp, gp, op, (long)rows, C, silu); break;
    case 8: hipLaunchKernelGGL(rmsnorm_silu_kernel<8>, grid, block, 0, s, xp, gp, op, (long)rows, C, silu); break;
    case 16: hipLaunchKernelGGL(rmsnorm_silu_kernel<16>, grid, block, 0, s, xp, gp, op, (long)rows, C, silu); break;
    case 32: hipLaunchKernelGGL(rmsnorm_silu_kernel<32>, grid, block, 0, s, xp, gp, op, (long)rows, C, silu); break;
    default: hipLaunchKernelGGL(rmsnorm_silu_kernel<64>, grid, block, 0, s, xp, gp, op, (long)rows, C, silu); break;
  }
  SF_HIP_LAUNCH_CHECK("sf_rmsnorm_silu_cl");
  return 0;
}

extern "C" int sf_softmax_rows(const float* sm, int64_t lds, void* p, int64_t ldp, int rows, int cols, int cols_padded,
                               float scale, void* stream) {
  SF_CHECK(sm && p, "sf_softmax_rows: null tensor");
  SF_CHECK(rows > 0 && cols > 0 && cols_padded >= cols && lds >= cols && ldp >= cols_padded, "sf_softmax_rows: bad shape rows=%d cols=%d padded=%d", rows, cols, cols_padded);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, sm, (long)lds, (bf16_t*)p, (long)ldp, cols, cols_padded, scale);
  SF_HIP_LAUNCH_CHECK("sf_softmax_rows");
  return 0;
}

extern "C" int sf_vae_prepare_latent(const void* latent, const float* mean, const float* stdv, const void* conv2_w, const void* conv2_b,
                                     void* out, int z, int h, int w, int c_pad, void* stream) {
  SF_CHECK(latent && mean && stdv && conv2_w && conv2_b && out, "sf_vae_prepare_latent: null tensor");
  SF_CHECK(z > 0 && z <= 32 && c_pad >= z && h > 0 && w > 0, "sf_vae_prepare_latent: bad shape z=%d c_pad=%d", z, c_pad);
  const int hw = h * w;
  hipLaunchKernelGGL(prepare_latent_kernel, dim3((hw + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)latent, mean, stdv,
                     (const bf16_t*)conv2_w, (const bf16_t*)conv2_b, (bf16_t*)out, z, hw, c_pad);
  SF_HIP_LAUNCH_CHECK("sf_vae_prepare_latent");
  return 0;
}
