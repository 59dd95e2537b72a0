// HBM-bound fused kernels of the Self-Forcing hot path for gfx950 (MI355X).
//
// All of them are one-pass, 16-byte-per-lane vectorised, one wavefront (64 lanes) per row so the
// row statistics are wave shuffles only (no LDS, no barrier):
//   layernorm_modulate  LN(x) * (1 + scale) + shift     causal_model.py:315, :327-328, :366
//   layernorm_affine    norm3                           causal_model.py:324, model.py:89-99
//   rmsnorm             WanRMSNorm                      model.py:70-86
//   qkv_norm_rope_cache q/k RMSNorm + 3-axis RoPE + KV-cache append   causal_model.py:112-114,
//                                                       :28-56, :195-200, :221-229
//   copy_rows           rolling-window eviction         causal_model.py:212-217
//   patchify / unpatchify_x0 / add_noise / sinusoid     causal_model.py:775-781, :1081-1104,
//                                                       wan_wrapper.py:204-228, scheduler.py:159-176,
//                                                       model.py:15-25
#include "sf_common.h"
#include "../../include/sf_hip.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves, one row each

// ------------------------------------------------------------------------------------------
// row loaders: lane l holds chunks i = 0..NCH-1, chunk i = elements (i*64 + l)*8 .. +8
template <int NCH>
__device__ __forceinline__ void load_row(const bf16_t* row, int lane, float (&v)[NCH][8]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(row + (i * 64 + lane) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[i][j] = (float)t[j];
  }
}

template <int NCH>
__device__ __forceinline__ void ln_stats(const float (&v)[NCH][8], int C, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[i][j];
  mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[i][j] - mean;
      q += d * d;
    }
  rstd = rsqrtf(wave_sum(q) / (float)C + eps);
}

// MODE 0: modulate (shift/scale = mod + e0[group]); MODE 1: affine weight/bias.
// One wave per workgroup, LN_RPW consecutive rows per wave: the four (two) parameter vectors are as many bytes as four
// (two) rows, so loading them once per ROW made the kernel's load path (64 B/clk/CU) move 3x the row's bytes (3.4 TB/s
// of HBM traffic measured); here they are combined once per wave (again when a row starts a new modulation group) and
// the rows' loads are all issued before the first statistic.
#ifndef SF_LN_RPW
#define SF_LN_RPW 2
#endif
template <int NCH> struct LnRows { static constexpr int value = NCH <= 4 ? SF_LN_RPW : (NCH <= 6 ? 2 : 1); };   // register budget: rows in flight
template <int NCH, int MODE>
__global__ __launch_bounds__(64) void layernorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                       int M, int C, float eps, const bf16_t* __restrict__ p0,
                                                       const bf16_t* __restrict__ p1, const bf16_t* __restrict__ e0_shift,
                                                       const bf16_t* __restrict__ e0_scale, long e0_group_stride,
                                                       int rows_per_group) {
  constexpr int LN_RPW = LnRows<NCH>::value;
  const int lane = threadIdx.x;
  const int row0 = blockIdx.x * LN_RPW;
  bf16x8 raw[LN_RPW][NCH];
#pragma unroll
  for (int r = 0; r < LN_RPW; ++r) {
    const long row = min(row0 + r, M - 1);
#pragma unroll
    for (int i = 0; i < NCH; ++i) raw[r][i] = *reinterpret_cast<const bf16x8*>(x + row * C + (i * 64 + lane) * 8);
  }
  bf16x8 scl[NCH], add[NCH];                // y = (x - mean) * rstd * (1 + scl) + add  (MODE 1: * scl + add); kept packed
  int cur_group = -1;
#pragma unroll
  for (int r = 0; r < LN_RPW; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    const int grp = MODE == 0 ? row / rows_per_group : 0;
    if (grp != cur_group) {
      cur_group = grp;
      const long goff = (long)grp * e0_group_stride;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int e = (i * 64 + lane) * 8;
        if (MODE == 0) {
          const bf16x8 ms = *reinterpret_cast<const bf16x8*>(p0 + e);       // modulation shift
          const bf16x8 mc = *reinterpret_cast<const bf16x8*>(p1 + e);       // modulation scale
          const bf16x8 es = *reinterpret_cast<const bf16x8*>(e0_shift + goff + e);
          const bf16x8 ec = *reinterpret_cast<const bf16x8*>(e0_scale + goff + e);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            // (modulation + e0) is a bf16 tensor in the reference (causal_model.py:310)
            add[i][j] = (bf16_t)((float)ms[j] + (float)es[j]);
            scl[i][j] = (bf16_t)((float)mc[j] + (float)ec[j]);
          }
        } else {
          scl[i] = *reinterpret_cast<const bf16x8*>(p0 + e);
          add[i] = *reinterpret_cast<const bf16x8*>(p1 + e);
        }
      }
    }
    float v[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = (float)raw[r][i][j];
    float mean, rstd;
    ln_stats<NCH>(v, C, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        o[j] = (bf16_t)((v[i][j] - mean) * rstd * (MODE == 0 ? 1.0f + (float)scl[i][j] : (float)scl[i][j]) + (float)add[i][j]);
      *reinterpret_cast<bf16x8*>(out + (long)row * C + (i * 64 + lane) * 8) = o;
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ out, int ldo, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[NCH][8];
  load_row<NCH>(x + (long)row * ldx, lane, v);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) q += v[i][j] * v[i][j];
  const float r = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int e = (i * 64 + lane) * 8;
    const bf16x8 wt = *reinterpret_cast<const bf16x8*>(w + e);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)(bf16_t)(v[i][j] * r) * (float)wt[j]);
    *reinterpret_cast<bf16x8*>(out + (long)row * ldo + e) = o;
  }
}

// q/k RMSNorm + RoPE + cache append.  One wave per token row of qkv [B*L, 3C].
template <int NCH>
__global__ __launch_bounds__(256) void qkv_norm_rope_cache_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
    bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_cache, bf16_t* __restrict__ v_cache,
    const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, int rows, int L, int hgt, int wid, int C,
    long cache_tokens, int write_start, int start_frame, float eps, int c0, int c1) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / L, t = row - b * L;
  const int fi = t / (hgt * wid), rem = t - fi * (hgt * wid);
  const int hi = rem / wid, wi = rem - hi * wid;
  const int pos_t = start_frame + fi;
  const bf16_t* src = qkv + (long)row * 3 * C;
  const long crow = ((long)b * cache_tokens + write_start + t) * C;

  // v: plain copy into the cache
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int e = (i * 64 + lane) * 8;
    *reinterpret_cast<u32x4*>(v_cache + crow + e) = *reinterpret_cast<const u32x4*>(src + 2 * C + e);
  }

#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float v[NCH][8];
    load_row<NCH>(src + which * C, lane, v);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) q += v[i][j] * v[i][j];
    const float r = rsqrtf(wave_sum(q) / (float)C + eps);
    const bf16_t* wn = which ? wk : wq;
    bf16_t* dst = which ? (k_cache + crow) : (q_out + (long)row * C);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int e = (i * 64 + lane) * 8;
      const bf16x8 wt = *reinterpret_cast<const bf16x8*>(wn + e);
      const int pair0 = (e & 127) >> 1;  // first of the 4 rotary pairs of this chunk (head_dim 128)
      bf16x8 o;
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int pj = pair0 + pp;
        const int pos = pj < c0 ? pos_t : (pj < c0 + c1 ? hi : wi);
        const float cs = rope_cos[pos * 64 + pj];
        const float sn = rope_sin[pos * 64 + pj];
        // RMSNorm output is bf16, then * weight in bf16 (model.py:83), then RoPE (fp64 in the
        // reference; fp32 here on bf16-exact inputs), one rounding at the end
        const float re = (float)(bf16_t)((float)(bf16_t)(v[i][2 * pp] * r) * (float)wt[2 * pp]);
        const float im = (float)(bf16_t)((float)(bf16_t)(v[i][2 * pp + 1] * r) * (float)wt[2 * pp + 1]);
        o[2 * pp] = (bf16_t)(re * cs - im * sn);
        o[2 * pp + 1] = (bf16_t)(re * sn + im * cs);
      }
      *reinterpret_cast<bf16x8*>(dst + e) = o;
    }
  }
}

// generic 16-byte row copy: dst[b][r][:] = src[b][r][:], r < nrows, row_bytes % 16 == 0
__global__ __launch_bounds__(256) void copy_rows_kernel(const char* __restrict__ src, long src_bstride,
                                                        char* __restrict__ dst, long dst_bstride, long bytes_per_batch) {
  const long n16 = bytes_per_batch >> 4;
  const char* s = src + (long)blockIdx.y * src_bstride;
  char* d = dst + (long)blockIdx.y * dst_bstride;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
    reinterpret_cast<u32x4*>(d)[i] = reinterpret_cast<const u32x4*>(s)[i];
}

// x [B, F, Cin, H, W] -> cols [B*F*h*w, Cin*4], column c*4 + p*2 + q
__global__ __launch_bounds__(256) void patchify_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ cols, long total,
                                                       int Cin, int H, int W) {
  const int h2 = H >> 1, w2 = W >> 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // i enumerates (bf, hh, ww, c)
    const int cch = (int)(i % Cin);
    long r = i / Cin;
    const int ww = (int)(r % w2); r /= w2;
    const int hh = (int)(r % h2);
    const long bf = r / h2;
    const bf16_t* s = x + ((bf * Cin + cch) * H + 2 * hh) * (long)W + 2 * ww;
    bf16x4 o;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[W]; o[3] = s[W + 1];
    *reinterpret_cast<bf16x4*>(cols + ((bf * h2 + hh) * w2 + ww) * (long)(Cin * 4) + cch * 4) = o;
  }
}

// nearest-timestep sigma lookup by one workgroup (first index of the minimum, as torch.argmin)
__device__ __forceinline__ double block_sigma_lookup(double t, const float* __restrict__ sigmas,
                                                     const float* __restrict__ timesteps, int n_table) {
  __shared__ double s_val[4];
  __shared__ int s_idx[4];
  double best = 1e300;
  int bi = 0x7fffffff;
  for (int j = threadIdx.x; j < n_table; j += blockDim.x) {
    const double d = fabs((double)timesteps[j] - t);
    if (d < best) { best = d; bi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_val[wv] = best; s_idx[wv] = bi; }
  __syncthreads();
  best = s_val[0]; bi = s_idx[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
    if (s_val[k] < best || (s_val[k] == best && s_idx[k] < bi)) { best = s_val[k]; bi = s_idx[k]; }
  return (double)sigmas[bi];
}

// double -> bf16 with a single round-to-nearest-even (torch's `.to(bfloat16)` on a float64 tensor);
// going through float would round twice.
__device__ __forceinline__ bf16_t double_to_bf16(double d) {
  const float f = (float)d;
  const unsigned u = __float_as_uint(f);
  if ((u & 0xFFFFu) == 0x8000u) {      // f sits on a bf16 tie point: decide by the residual
    const double r = d - (double)f;
    if (r != 0.0) {
      const bool up = (f > 0.f) ? (r > 0.0) : (r < 0.0);  // magnitude up?
      const unsigned hi = (u >> 16) + (up ? 1u : 0u);
      return __builtin_bit_cast(bf16_t, (unsigned short)hi);
    }
  }
  return (bf16_t)f;
}

__device__ __forceinline__ double read_timestep(const void* t, int is_i64, long idx) {
  return is_i64 ? (double)reinterpret_cast<const long long*>(t)[idx] : (double)reinterpret_cast<const float*>(t)[idx];
}

// grid (blocks over Cout*H*(W/2), B*F): flow[b,f,c,y,x] from head_out, x0 = xt - sigma*flow (fp64)
__global__ __launch_bounds__(256) void unpatchify_x0_kernel(const bf16_t* __restrict__ head_out, const bf16_t* __restrict__ xt,
                                                            const void* __restrict__ timestep, int t_is_i64,
                                                            const float* __restrict__ sigmas, const float* __restrict__ timesteps,
                                                            int n_table, bf16_t* __restrict__ flow, bf16_t* __restrict__ x0,
                                                            int F, int groups, int Cout, int H, int W) {
#pragma clang fp contract(off)  // xt - sigma*flow as a rounded product then a subtraction (fp64)
  const int bf = blockIdx.y;
  const int b = bf / F, f = bf - b * F;
  const int grp = f / (F / groups);
  const double sigma = block_sigma_lookup(read_timestep(timestep, t_is_i64, (long)b * groups + grp), sigmas, timesteps, n_table);
  const int h2 = H >> 1, w2 = W >> 1;
  const int total = Cout * H * w2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int ww = i % w2;
    int r = i / w2;
    const int y = r % H;
    const int cch = r / H;
    const int hh = y >> 1, pq = y & 1;
    const long tok = ((long)bf * h2 + hh) * w2 + ww;
    const bf16_t* hs = head_out + tok * (4 * Cout) + (pq * 2) * Cout + cch;
    const long o = (((long)bf * Cout + cch) * H + y) * W + 2 * ww;
    const bf16_t f0 = hs[0], f1 = hs[Cout];
    bf16x2 fo; fo[0] = f0; fo[1] = f1;
    *reinterpret_cast<bf16x2*>(flow + o) = fo;
    const bf16x2 xv = *reinterpret_cast<const bf16x2*>(xt + o);
    bf16x2 xo;
    xo[0] = double_to_bf16((double)(float)xv[0] - sigma * (double)(float)f0);
    xo[1] = double_to_bf16((double)(float)xv[1] - sigma * (double)(float)f1);
    *reinterpret_cast<bf16x2*>(x0 + o) = xo;
  }
}

// grid (blocks over inner/8, n_outer)
__global__ __launch_bounds__(256) void add_noise_kernel(const bf16_t* __restrict__ x0, const bf16_t* __restrict__ eps,
                                                        const void* __restrict__ timestep, int t_is_i64,
                                                        const float* __restrict__ sigmas, const float* __restrict__ timesteps,
                                                        int n_table, bf16_t* __restrict__ out, long inner) {
#pragma clang fp contract(off)  // torch evaluates (1-s)*x0, s*eps and their sum as separate fp32 ops
  const long base = (long)blockIdx.y * inner;
  const float sigma = (float)block_sigma_lookup(read_timestep(timestep, t_is_i64, blockIdx.y), sigmas, timesteps, n_table);
  const float om = 1.0f - sigma;
  const long n8 = inner >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(x0 + base + i * 8);
    const bf16x8 e = *reinterpret_cast<const bf16x8*>(eps + base + i * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(om * (float)a[j] + sigma * (float)e[j]);
    *reinterpret_cast<bf16x8*>(out + base + i * 8) = o;
  }
}

// out[i] = sum_k c[k] x_k[i]: the UniPC predictor / corrector and the classifier-free-guidance blend are linear in
// their tensors (fm_solvers_unipc.py:350-626, causal_diffusion_inference.py:423-424); the host evaluates the scalars.
struct LinCombP {
  const bf16_t* x[SF_LINCOMB_MAX];
  float c[SF_LINCOMB_MAX];
  bf16_t* out;
  long n8;
  int n_terms;
};
template <int NT>
__global__ __launch_bounds__(256) void lincomb_kernel(const LinCombP p) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n8; i += (long)gridDim.x * blockDim.x) {
    float acc[8];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.x[k] + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = k == 0 ? p.c[0] * (float)v[j] : acc[j] + p.c[k] * (float)v[j];
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)acc[j];
    *reinterpret_cast<bf16x8*>(p.out + i * 8) = o;
  }
}

__global__ __launch_bounds__(256) void sinusoid_kernel(const void* __restrict__ t, int t_is_i64, bf16_t* __restrict__ out,
                                                       int n, int dim) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * half) return;
  const int r = i / half, k = i - r * half;
  const double pos = read_timestep(t, t_is_i64, r);
  const double ang = pos * pow(10000.0, -(double)k / (double)half);
  out[(long)r * dim + k] = double_to_bf16(cos(ang));
  out[(long)r * dim + half + k] = double_to_bf16(sin(ang));
}

template <int MODE>
int launch_layernorm(const bf16_t* x, bf16_t* out, int M, int C, float eps, const bf16_t* p0, const bf16_t* p1,
                     const bf16_t* e0s, const bf16_t* e0c, long gs, int rpg, hipStream_t s) {
  const dim3 block(64);
  switch (C / 512) {
#define SF_LN_CASE(N) \
  case N: hipLaunchKernelGGL((layernorm_kernel<N, MODE>), dim3((M + LnRows<N>::value - 1) / LnRows<N>::value), block, 0, s, x, out, M, C, eps, p0, p1, e0s, e0c, gs, rpg); break;
    SF_LN_CASE(1) SF_LN_CASE(2) SF_LN_CASE(3) SF_LN_CASE(4) SF_LN_CASE(5) SF_LN_CASE(6) SF_LN_CASE(8) SF_LN_CASE(10)
#undef SF_LN_CASE
    default: sf_set_error("layernorm: unsupported channel count %d", C); return -1;
  }
  return 0;
}

bool channels_ok(int C) {
  if (C % 512 != 0) return false;
  const int n = C / 512;
  return n == 1 || n == 2 || n == 3 || n == 4 || n == 5 || n == 6 || n == 8 || n == 10;
}

}  // namespace

extern "C" int sf_layernorm_modulate(const void* x, void* out, int M, int C, float eps, const void* mod_shift,
                                     const void* mod_scale, const void* e0_shift, const void* e0_scale,
                                     int64_t e0_group_stride, int rows_per_group, void* stream) {
  SF_CHECK(x && out && mod_shift && mod_scale && e0_shift && e0_scale, "sf_layernorm_modulate: null tensor");
  SF_CHECK(M > 0 && channels_ok(C), "sf_layernorm_modulate: unsupported shape M=%d C=%d (C must be 512*{1,2,3,4,5,6,8,10})", M, C);
  SF_CHECK(rows_per_group > 0 && e0_group_stride % 8 == 0, "sf_layernorm_modulate: bad group layout");
  if (launch_layernorm<0>((const bf16_t*)x, (bf16_t*)out, M, C, eps, (const bf16_t*)mod_shift, (const bf16_t*)mod_scale,
                          (const bf16_t*)e0_shift, (const bf16_t*)e0_scale, e0_group_stride, rows_per_group,
                          (hipStream_t)stream) != 0)
    return -1;
  SF_HIP_LAUNCH_CHECK("sf_layernorm_modulate");
  return 0;
}

extern "C" int sf_layernorm_affine(const void* x, const void* weight, const void* bias, void* out, int M, int C,
                                   float eps, void* stream) {
  SF_CHECK(x && out && weight && bias, "sf_layernorm_affine: null tensor");
  SF_CHECK(M > 0 && channels_ok(C), "sf_layernorm_affine: unsupported shape M=%d C=%d", M, C);
  if (launch_layernorm<1>((const bf16_t*)x, (bf16_t*)out, M, C, eps, (const bf16_t*)weight, (const bf16_t*)bias, nullptr,
                          nullptr, 0, 1, (hipStream_t)stream) != 0)
    return -1;
  SF_HIP_LAUNCH_CHECK("sf_layernorm_affine");
  return 0;
}

extern "C" int sf_rmsnorm(const void* x, int ldx, const void* weight, void* out, int ldo, int M, int C, float eps,
                          void* stream) {
  SF_CHECK(x && out && weight, "sf_rmsnorm: null tensor");
  SF_CHECK(M > 0 && channels_ok(C) && ldx >= C && ldo >= C && ldx % 8 == 0 && ldo % 8 == 0, "sf_rmsnorm: unsupported shape M=%d C=%d", M, C);
  const dim3 grid((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (C / 512) {
#define SF_RMS_CASE(N) \
  case N: hipLaunchKernelGGL((rmsnorm_kernel<N>), grid, block, 0, s, (const bf16_t*)x, ldx, (const bf16_t*)weight, (bf16_t*)out, ldo, M, C, eps); break;
    SF_RMS_CASE(1) SF_RMS_CASE(2) SF_RMS_CASE(3) SF_RMS_CASE(4) SF_RMS_CASE(5) SF_RMS_CASE(6) SF_RMS_CASE(8) SF_RMS_CASE(10)
#undef SF_RMS_CASE
  }
  SF_HIP_LAUNCH_CHECK("sf_rmsnorm");
  return 0;
}

extern "C" int sf_qkv_norm_rope_cache(const void* qkv, const void* norm_q_w, const void* norm_k_w, void* q_out,
                                      void* k_cache, void* v_cache, const float* rope_cos, const float* rope_sin,
                                      int B, int f, int h, int w, int C, int num_heads, int64_t cache_tokens,
                                      int write_start, int start_frame, float eps, void* stream) {
  SF_CHECK(qkv && norm_q_w && norm_k_w && q_out && k_cache && v_cache && rope_cos && rope_sin, "sf_qkv_norm_rope_cache: null tensor");
  SF_CHECK(B > 0 && f > 0 && h > 0 && w > 0 && channels_ok(C), "sf_qkv_norm_rope_cache: unsupported shape");
  SF_CHECK(num_heads > 0 && C == num_heads * 128, "sf_qkv_norm_rope_cache: head_dim must be 128 (C=%d heads=%d)", C, num_heads);
  const int L = f * h * w;
  SF_CHECK(write_start >= 0 && (int64_t)write_start + L <= cache_tokens, "sf_qkv_norm_rope_cache: cache overflow: write_start=%d + L=%d > capacity=%lld",
           write_start, L, (long long)cache_tokens);
  SF_CHECK(start_frame >= 0 && start_frame + f <= 1024 && h <= 1024 && w <= 1024, "sf_qkv_norm_rope_cache: RoPE position out of the 1024-entry table");
  const int rows = B * L;
  const dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
  hipStream_t s = (hipStream_t)stream;
  // column split of the reference's freqs (causal_model.py:32): c = 64 -> (22, 21, 21)
  const int c = 64, c1 = c / 3, c0 = c - 2 * c1;
  switch (C / 512) {
#define SF_QKV_CASE(N)                                                                                                   \
  case N: hipLaunchKernelGGL((qkv_norm_rope_cache_kernel<N>), grid, block, 0, s, (const bf16_t*)qkv, (const bf16_t*)norm_q_w, \
                             (const bf16_t*)norm_k_w, (bf16_t*)q_out, (bf16_t*)k_cache, (bf16_t*)v_cache, rope_cos, rope_sin,  \
                             rows, L, h, w, C, (long)cache_tokens, write_start, start_frame, eps, c0, c1); break;
    SF_QKV_CASE(1) SF_QKV_CASE(2) SF_QKV_CASE(3) SF_QKV_CASE(4) SF_QKV_CASE(5) SF_QKV_CASE(6) SF_QKV_CASE(8) SF_QKV_CASE(10)
#undef SF_QKV_CASE
  }
  SF_HIP_LAUNCH_CHECK("sf_qkv_norm_rope_cache");
  return 0;
}

extern "C" int sf_kv_evict(void* cache, int B, int64_t cache_tokens, int row_elems, int sink, int evict, int keep,
                           void* scratch, size_t scratch_bytes, void* stream) {
  SF_CHECK(cache && B > 0 && row_elems > 0 && (row_elems % 8) == 0, "sf_kv_evict: bad arguments");
  SF_CHECK(sink >= 0 && evict >= 0 && keep >= 0 && (int64_t)sink + evict + keep <= cache_tokens, "sf_kv_evict: window out of range");
  if (evict == 0 || keep == 0) return 0;
  const long row_b = (long)row_elems * 2;
  const long bytes = (long)keep * row_b;
  SF_CHECK(scratch && scratch_bytes >= (size_t)(bytes * B), "sf_kv_evict: scratch too small (%zu < %ld)", scratch_bytes, bytes * B);
  char* base = (char*)cache;
  const long bstride = (long)cache_tokens * row_b;
  const int gx = (int)min((long)2048, (bytes / 16 + 255) / 256);
  const dim3 grid(gx, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  // the source and destination windows overlap and workgroups run in no defined order:
  // go through a scratch buffer (cache -> scratch -> cache)
  hipLaunchKernelGGL(copy_rows_kernel, grid, block, 0, s, base + (long)(sink + evict) * row_b, bstride, (char*)scratch, bytes, bytes);
  hipLaunchKernelGGL(copy_rows_kernel, grid, block, 0, s, (const char*)scratch, bytes, base + (long)sink * row_b, bstride, bytes);
  SF_HIP_LAUNCH_CHECK("sf_kv_evict");
  return 0;
}

extern "C" int sf_patchify(const void* x, void* cols, int B, int F, int Cin, int H, int W, void* stream) {
  SF_CHECK(x && cols && B > 0 && F > 0 && Cin > 0, "sf_patchify: bad arguments");
  SF_CHECK(H % 2 == 0 && W % 2 == 0, "sf_patchify: latent H=%d W=%d must be even (patch 2x2)", H, W);
  const long total = (long)B * F * (H / 2) * (W / 2) * Cin;
  const int gx = (int)min((long)4096, (total + 255) / 256);
  hipLaunchKernelGGL(patchify_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)cols, total, Cin, H, W);
  SF_HIP_LAUNCH_CHECK("sf_patchify");
  return 0;
}

extern "C" int sf_unpatchify_x0(const void* head_out, const void* xt, const void* timestep, int t_is_int64,
                                const float* sigmas, const float* timesteps, int n_table, void* flow, void* x0,
                                int B, int F, int groups, int Cout, int H, int W, void* stream) {
  SF_CHECK(head_out && xt && timestep && sigmas && timesteps && flow && x0, "sf_unpatchify_x0: null tensor");
  SF_CHECK(B > 0 && F > 0 && groups > 0 && F % groups == 0, "sf_unpatchify_x0: frames=%d not divisible by groups=%d", F, groups);
  SF_CHECK(H % 2 == 0 && W % 2 == 0 && n_table > 0, "sf_unpatchify_x0: bad shape");
  const int total = Cout * H * (W / 2);
  const dim3 grid(min(64, (total + 255) / 256), B * F), block(256);
  hipLaunchKernelGGL(unpatchify_x0_kernel, grid, block, 0, (hipStream_t)stream, (const bf16_t*)head_out, (const bf16_t*)xt, timestep,
                     t_is_int64, sigmas, timesteps, n_table, (bf16_t*)flow, (bf16_t*)x0, F, groups, Cout, H, W);
  SF_HIP_LAUNCH_CHECK("sf_unpatchify_x0");
  return 0;
}

extern "C" int sf_add_noise(const void* x0, const void* eps, const void* timestep, int t_is_int64, const float* sigmas,
                            const float* timesteps, int n_table, void* out, int n_outer, int64_t inner, void* stream) {
  SF_CHECK(x0 && eps && timestep && sigmas && timesteps && out, "sf_add_noise: null tensor");
  SF_CHECK(n_outer > 0 && inner > 0 && inner % 8 == 0 && n_table > 0, "sf_add_noise: bad shape (inner must be a multiple of 8)");
  const dim3 grid((unsigned)min((long)64, (long)((inner / 8 + 255) / 256)), n_outer), block(256);
  hipLaunchKernelGGL(add_noise_kernel, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x0, (const bf16_t*)eps, timestep, t_is_int64,
                     sigmas, timesteps, n_table, (bf16_t*)out, (long)inner);
  SF_HIP_LAUNCH_CHECK("sf_add_noise");
  return 0;
}

extern "C" int sf_lincomb_bf16(void* out, const void* const* xs, const float* coefs, int n_terms, int64_t n, void* stream) {
  SF_CHECK(out && xs && coefs, "sf_lincomb_bf16: null argument");
  SF_CHECK(n_terms >= 1 && n_terms <= SF_LINCOMB_MAX, "sf_lincomb_bf16: n_terms must be 1..%d, got %d", SF_LINCOMB_MAX, n_terms);
  SF_CHECK(n > 0 && n % 8 == 0, "sf_lincomb_bf16: element count must be a positive multiple of 8");
  LinCombP p;
  for (int k = 0; k < SF_LINCOMB_MAX; ++k) {
    p.x[k] = k < n_terms ? (const bf16_t*)xs[k] : nullptr;
    p.c[k] = k < n_terms ? coefs[k] : 0.0f;
    SF_CHECK(k >= n_terms || xs[k], "sf_lincomb_bf16: null input %d", k);
  }
  p.out = (bf16_t*)out;
  p.n8 = n / 8;
  p.n_terms = n_terms;
  const dim3 grid((unsigned)min((long)2048, (p.n8 + 255) / 256)), block(256);
  switch (n_terms) {
    case 1: hipLaunchKernelGGL(lincomb_kernel<1>, grid, block, 0, (hipStream_t)stream, p); break;
    case 2: hipLaunchKernelGGL(lincomb_kernel<2>, grid, block, 0, (hipStream_t)stream, p); break;
    case 3: hipLaunchKernelGGL(lincomb_kernel<3>, grid, block, 0, (hipStream_t)stream, p); break;
    case 4: hipLaunchKernelGGL(lincomb_kernel<4>, grid, block, 0, (hipStream_t)stream, p); break;
    case 5: hipLaunchKernelGGL(lincomb_kernel<5>, grid, block, 0, (hipStream_t)stream, p); break;
    default: hipLaunchKernelGGL(lincomb_kernel<6>, grid, block, 0, (hipStream_t)stream, p); break;
  }
  SF_HIP_LAUNCH_CHECK("sf_lincomb_bf16");
  return 0;
}

// the cache dicts' index tensors (causal_model.py:235-236), all layers at once: buf int64 [layers][2] <- (global_end, local_end)
__global__ void kv_index_kernel(long long* __restrict__ buf, int layers, long long global_end, long long local_end) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * layers) buf[i] = (i & 1) ? local_end : global_end;
}
// (library-internal: called by sf_dit_forward only)
__attribute__((visibility("hidden"))) int sf_internal_write_kv_indices(void* buf, int layers, int64_t global_end, int64_t local_end, void* stream) {
  hipLaunchKernelGGL(kv_index_kernel, dim3((2 * layers + 63) / 64), dim3(64), 0, (hipStream_t)stream, (long long*)buf, layers,
                     (long long)global_end, (long long)local_end);
  SF_HIP_LAUNCH_CHECK("kv_index_kernel");
  return 0;
}

extern "C" int sf_sinusoid_embedding(const void* t, int t_is_int64, void* out, int n, int dim, void* stream) {
  SF_CHECK(t && out && n > 0 && dim > 0 && dim % 2 == 0, "sf_sinusoid_embedding: bad arguments");
  const int total = n * (dim / 2);
  hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, t_is_int64, (bf16_t*)out, n, dim);
  SF_HIP_LAUNCH_CHECK("sf_sinusoid_embedding");
  return 0;
}
