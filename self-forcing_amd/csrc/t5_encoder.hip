// umT5 text encoder for gfx950: the HBM-bound helper kernels and the host sequencer of one encoder pass
// (T5Encoder.forward, wan/modules/t5.py:299-312).  All matrix products go through sf_gemm_bf16; the
// attention (64 heads of 64, logits + relative-position bias + key mask, softmax in fp32, t5.py:104-118) is
// two batched GEMM launches (one grid row per head) around a bias/softmax kernel.  The pass runs once per
// prompt (0.5 % of a rollout's FLOPs): 14 ms for umT5-XXL on 512 tokens.
#include <cmath>
#include <cstring>
#include "sf_common.h"
#include "../../include/sf_hip.h"

namespace {

__global__ __launch_bounds__(256) void embedding_gather_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table,
                                                               bf16_t* __restrict__ out, int dim, int vocab) {
  long id = ids[blockIdx.x];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const bf16x8* src = reinterpret_cast<const bf16x8*>(table + id * dim);
  bf16x8* dst = reinterpret_cast<bf16x8*>(out + (long)blockIdx.x * dim);
  for (int c = threadIdx.x; c < dim / 8; c += 256) dst[c] = src[c];
}

// one 256-thread block per (head, query) row
__global__ __launch_bounds__(256) void t5_softmax_bias_kernel(const float* __restrict__ s, bf16_t* __restrict__ p, const bf16_t* __restrict__ emb,
                                                              const int* __restrict__ rel_bucket, const int64_t* __restrict__ key_mask,
                                                              int H, int L, int ld) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x / L, i = blockIdx.x - h * L;
  const float* sr = s + (long)blockIdx.x * ld;
  bf16_t* pr = p + (long)blockIdx.x * ld;
  const int* rb = rel_bucket + (L - 1 - i);       // rb[j] = bucket of (j - i)
  auto logit = [&](int j) { return key_mask[j] != 0 ? sr[j] + (float)emb[rb[j] * H + h] : -3.0e38f; };
  float mx = -3.4e38f;
  for (int j = tid; j < L; j += 256) mx = fmaxf(mx, logit(j));
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int j = tid; j < L; j += 256) sum += __expf(logit(j) - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int j = tid; j < ld; j += 256) pr[j] = j < L ? (bf16_t)(__expf(logit(j) - mx) * inv) : (bf16_t)0.f;
}

__global__ __launch_bounds__(256) void mul_bf16_kernel(const bf16x8* __restrict__ a, const bf16x8* __restrict__ b, bf16x8* __restrict__ out, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const bf16x8 x = a[i], y = b[i];
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)x[j] * (float)y[j]);
  out[i] = o;
}

__global__ __launch_bounds__(256) void zero_masked_rows_kernel(bf16_t* __restrict__ x, const int64_t* __restrict__ mask, int dim) {
  if (mask[blockIdx.x] != 0) return;
  bf16x8* row = reinterpret_cast<bf16x8*>(x + (long)blockIdx.x * dim);
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.f;
  for (int c = threadIdx.x; c < dim / 8; c += 256) row[c] = z;
}

struct Carve {
  char* base;
  size_t off;
  explicit Carve(void* p) : base((char*)p), off(0) {}
  char* take(size_t bytes) {
    char* r = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return r;
  }
};

struct Work {
  char *x, *xn, *qk, *vt, *sc, *p, *ao, *g, *h;
  int lpad;
  size_t total;
};

Work carve(const sf_t5_model* m, void* ws, int B, int L) {
  Work w;
  const size_t M = (size_t)B * L, D = m->dim, Da = m->dim_attn, F = m->dim_ffn, H = m->num_heads;
  w.lpad = (L + 63) & ~63;
  Carve c(ws);
  w.x = c.take(M * D * 2);
  w.xn = c.take(M * D * 2);
  w.qk = c.take(M * 2 * Da * 2);
  w.vt = c.take(Da * (size_t)w.lpad * 2);
  w.sc = c.take(H * L * (size_t)w.lpad * 4);
  w.p = c.take(H * L * (size_t)w.lpad * 2);
  w.ao = c.take(M * Da * 2);
  w.g = c.take(M * F * 2);
  w.h = c.take(M * F * 2);
  w.total = c.off;
  return w;
}

int gemm(const void* a, int lda, const void* w, int ldw, void* out, int ldo, int M, int N, int K, int epi, const void* resid, int ldr, void* stream,
         int batch = 1, long a_bs = 0, long w_bs = 0, long o_bs = 0) {
  sf_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.a = a; g.w = w; g.out = out; g.resid = resid; g.rows_per_group = 1;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo; g.ldr = ldr; g.epilogue = epi;
  g.batch = batch; g.a_bstride = a_bs; g.w_bstride = w_bs; g.o_bstride = o_bs;
  return sf_gemm_bf16(&g, stream);
}

int check_model(const sf_t5_model* m, int B, int L) {
  SF_CHECK(m && m->layers_host && m->token_embedding && m->final_norm_w, "sf_t5: null model");
  SF_CHECK(B > 0 && L > 0 && L % 4 == 0, "sf_t5: batch=%d seq_len=%d (seq_len must be a multiple of 4)", B, L);
  SF_CHECK(m->num_heads > 0 && m->dim_attn == m->num_heads * 64, "sf_t5: head_dim must be 64 (dim_attn=%d heads=%d)", m->dim_attn, m->num_heads);
  SF_CHECK(m->dim % 512 == 0 && m->dim_ffn % 64 == 0 && m->dim_attn % 64 == 0, "sf_t5: unsupported widths dim=%d ffn=%d", m->dim, m->dim_ffn);
  return 0;
}

}  // namespace

#define SF_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != 0) return rc__; \
  } while (0)

extern "C" int sf_embedding_gather(const int64_t* ids, const void* table, void* out, int n_tokens, int dim, int vocab, void* stream) {
  SF_CHECK(ids && table && out && n_tokens > 0 && dim > 0 && dim % 8 == 0 && vocab > 0, "sf_embedding_gather: bad arguments");
  hipLaunchKernelGGL(embedding_gather_kernel, dim3(n_tokens), dim3(256), 0, (hipStream_t)stream, ids, (const bf16_t*)table, (bf16_t*)out, dim, vocab);
  SF_HIP_LAUNCH_CHECK("sf_embedding_gather");
  return 0;
}

extern "C" int sf_t5_softmax_bias(const float* s, void* p, const void* emb, const int32_t* rel_bucket, const int64_t* key_mask,
                                  int H, int L, int ld, void* stream) {
  SF_CHECK(s && p && emb && rel_bucket && key_mask && H > 0 && L > 0 && ld >= L, "sf_t5_softmax_bias: bad arguments");
  hipLaunchKernelGGL(t5_softmax_bias_kernel, dim3(H * L), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)p, (const bf16_t*)emb, rel_bucket, key_mask, H, L, ld);
  SF_HIP_LAUNCH_CHECK("sf_t5_softmax_bias");
  return 0;
}

extern "C" int sf_mul_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  SF_CHECK(a && b && out && n > 0 && n % 8 == 0, "sf_mul_bf16: n=%lld must be a positive multiple of 8", (long long)n);
  const long n8 = n / 8;
  hipLaunchKernelGGL(mul_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)a, (const bf16x8*)b, (bf16x8*)out, n8);
  SF_HIP_LAUNCH_CHECK("sf_mul_bf16");
  return 0;
}

extern "C" int sf_zero_masked_rows(void* x, const int64_t* mask, int rows, int dim, void* stream) {
  SF_CHECK(x && mask && rows > 0 && dim > 0 && dim % 8 == 0, "sf_zero_masked_rows: bad arguments");
  hipLaunchKernelGGL(zero_masked_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, mask, dim);
  SF_HIP_LAUNCH_CHECK("sf_zero_masked_rows");
  return 0;
}

extern "C" size_t sf_t5_workspace_bytes(const sf_t5_model* m, int batch, int seq_len) {
  if (check_model(m, batch, seq_len) != 0) return 0;
  return carve(m, nullptr, batch, seq_len).total;
}

extern "C" int sf_t5_encode(const sf_t5_model* m, const int64_t* ids, const int64_t* mask, const int32_t* rel_bucket, int B, int L,
                            void* out, void* workspace, size_t workspace_bytes, void* stream) {
  SF_TRY(check_model(m, B, L));
  SF_CHECK(ids && mask && rel_bucket && out, "sf_t5_encode: null tensor");
  const Work w = carve(m, workspace, B, L);
  SF_CHECK(workspace && workspace_bytes >= w.total, "sf_t5_encode: workspace too small (%zu < %zu)", workspace_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  const int M = B * L, D = m->dim, Da = m->dim_attn, F = m->dim_ffn, H = m->num_heads, lp = w.lpad;

  SF_TRY(sf_embedding_gather(ids, m->token_embedding, w.x, M, D, m->vocab, stream));
  if (lp > L) {   // padded key columns of V^T stay zero for the whole pass
    hipError_t e = hipMemsetAsync(w.vt, 0, (size_t)Da * lp * 2, s);
    SF_CHECK(e == hipSuccess, "sf_t5_encode: memset failed: %s", hipGetErrorString(e));
  }
  for (int l = 0; l < m->num_layers; ++l) {
    const sf_t5_layer& ly = m->layers_host[l];
    // x = x + attn(norm1(x))   (t5.py:176)
    SF_TRY(sf_rmsnorm(w.x, D, ly.norm1_w, w.xn, D, M, D, m->eps, stream));
    SF_TRY(gemm(w.xn, D, ly.qk_w, D, w.qk, 2 * Da, M, 2 * Da, D, SF_EPI_BIAS, nullptr, 0, stream));
    for (int b = 0; b < B; ++b) {
      const char* xn_b = w.xn + (size_t)b * L * D * 2;
      const char* q_b = w.qk + (size_t)b * L * 2 * Da * 2;
      const char* k_b = q_b + (size_t)Da * 2;
      // V^T [Da][L] = Wv . xn_b^T straight from the projection
      SF_TRY(gemm(ly.v_w, D, xn_b, D, w.vt, lp, Da, L, D, SF_EPI_BIAS, nullptr, 0, stream));
      // logits of all heads in one batched launch, unscaled (t5.py:115): head h reads columns [64 h, 64 h + 64) of q and k
      SF_TRY(gemm(q_b, 2 * Da, k_b, 2 * Da, w.sc, lp, L, L, 64, SF_EPI_F32, nullptr, 0, stream, H, 64, 64, (long)L * lp));
      SF_TRY(sf_t5_softmax_bias((const float*)w.sc, w.p, ly.pos_emb, rel_bucket, mask + (size_t)b * L, H, L, lp, stream));
      SF_TRY(gemm(w.p, lp, w.vt, lp, w.ao + (size_t)b * L * Da * 2, Da, L, 64, lp, SF_EPI_BIAS, nullptr, 0, stream, H, (long)L * lp, 64L * lp, 64));
    }
    SF_TRY(gemm(w.ao, Da, ly.o_w, Da, w.x, D, M, D, Da, SF_EPI_BIAS_RESID, w.x, D, stream));
    // x = x + fc2(fc1(norm2(x)) * gelu(gate(norm2(x))))   (t5.py:137-142, :177)
    SF_TRY(sf_rmsnorm(w.x, D, ly.norm2_w, w.xn, D, M, D, m->eps, stream));
    SF_TRY(gemm(w.xn, D, ly.gate_w, D, w.g, F, M, F, D, SF_EPI_BIAS_GELU, nullptr, 0, stream));
    SF_TRY(gemm(w.xn, D, ly.fc1_w, D, w.h, F, M, F, D, SF_EPI_BIAS, nullptr, 0, stream));
    SF_TRY(sf_mul_bf16(w.h, w.g, w.h, (int64_t)M * F, stream));
    SF_TRY(gemm(w.h, F, ly.fc2_w, F, w.x, D, M, D, F, SF_EPI_BIAS_RESID, w.x, D, stream));
  }
  SF_TRY(sf_rmsnorm(w.x, D, m->final_norm_w, out, D, M, D, m->eps, stream));
  SF_TRY(sf_zero_masked_rows(out, mask, M, D, stream));
  return 0;
}
