// One whole denoiser pass of the causal Wan DiT as a single host call that enqueues every kernel
// on the caller's stream: CausalWanModel._forward_inference (wan/modules/causal_model.py:725-893)
// + the wrapper's flow -> x0 conversion (utils/wan_wrapper.py:288-300, :340-344).
//
// Host-side only (no kernels here): carves the caller's workspace and sequences the C-ABI
// launchers.  No allocation, no synchronisation, no device read-back: cache indices arrive as
// host integers (the pipeline always knows them), so the reference's >= 60 `.item()` syncs per
// forward (causal_model.py:207-226) disappear.
#include <cstring>
#include "sf_common.h"
#include "../../include/sf_hip.h"

int sf_internal_write_kv_indices(void* buf, int layers, int64_t global_end, int64_t local_end, void* stream);   // elementwise.hip

namespace {

struct Carve {
  char* base;
  size_t off;
  explicit Carve(void* p) : base((char*)p), off(0) {}
  void* take(size_t bytes) {
    void* r = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return r;
  }
};

struct Work {
  void *x, *xn, *qkv, *q, *att, *hbuf, *cols, *headout, *sin, *etmp, *e, *e0, *ctx1, *ctx;
  size_t total;
};

Work carve(const sf_model* m, void* ws, int B, int F, int lat_h, int lat_w, int groups) {
  const size_t M = (size_t)B * F * (lat_h / 2) * (lat_w / 2);
  const size_t C = m->dim, BG = (size_t)B * groups, T = (size_t)B * m->text_len;
  Carve c(ws);
  Work w;
  w.x = c.take(M * C * 2);
  w.xn = c.take(M * C * 2);
  w.qkv = c.take(M * 3 * C * 2);
  w.q = c.take(M * C * 2);
  w.att = c.take(M * C * 2);
  w.hbuf = c.take(M * (size_t)m->ffn_dim * 2);
  w.cols = c.take(M * (size_t)m->in_dim * 4 * 2);
  w.headout = c.take(M * (size_t)m->out_dim * 4 * 2);
  w.sin = c.take(BG * m->freq_dim * 2);
  w.etmp = c.take(BG * C * 2);
  w.e = c.take(BG * C * 2);
  w.e0 = c.take(BG * 6 * C * 2);
  w.ctx1 = c.take(T * C * 2);
  w.ctx = c.take(T * C * 2);
  w.total = c.off;
  return w;
}

int gemm(const void* a, int lda, const void* w, const void* bias, void* out, int ldo, int M, int N, int K, int epi,
         const void* resid, int ldr, const void* gate_mod, const void* gate_e0, long gstride, int rpg, void* stream) {
  sf_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.a = a; g.w = w; g.bias = bias; g.out = out; g.resid = resid; g.gate_mod = gate_mod; g.gate_e0 = gate_e0;
  g.gate_group_stride = gstride; g.rows_per_group = rpg;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = K; g.ldo = ldo; g.ldr = ldr; g.epilogue = epi;
  return sf_gemm_bf16(&g, stream);
}

inline const char* bptr(const void* p, size_t elems) { return (const char*)p + elems * 2; }

}  // namespace

#define SF_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != 0) return rc__; \
  } while (0)

extern "C" size_t sf_dit_workspace_bytes(const sf_model* model, int batch, int frames, int lat_h, int lat_w, int groups) {
  if (!model || batch <= 0 || frames <= 0 || lat_h <= 0 || lat_w <= 0 || groups <= 0) return 0;
  return carve(model, nullptr, batch, frames, lat_h, lat_w, groups).total;
}

extern "C" int sf_dit_forward(const sf_model* m, const sf_forward_args* a, void* stream) {
  SF_CHECK(m && a, "sf_dit_forward: null argument");
  SF_CHECK(m->layers_host && m->num_layers > 0, "sf_dit_forward: model has no layers");
  SF_CHECK(m->dim == m->num_heads * 128, "sf_dit_forward: head_dim must be 128 (dim=%d heads=%d)", m->dim, m->num_heads);
  SF_CHECK(a->batch > 0 && a->frames > 0 && a->groups > 0, "sf_dit_forward: empty input");
  SF_CHECK(a->lat_h % 2 == 0 && a->lat_w % 2 == 0, "sf_dit_forward: latent size must be even");
  const int B = a->batch, F = a->frames, h = a->lat_h / 2, w = a->lat_w / 2;
  const int L = F * h * w, M = B * L, C = m->dim, G = a->groups, BG = B * G;
  SF_CHECK(F % G == 0, "sf_dit_forward: frames=%d not divisible by timestep groups=%d", F, G);
  const int rpg = L / G;
  SF_CHECK(BG <= 32, "sf_dit_forward: batch*groups=%d exceeds the small-linear limit of 32", BG);
  SF_CHECK(a->noisy && a->timestep, "sf_dit_forward: null tensor");
  SF_CHECK(a->cache_only || (a->flow_out && a->x0_out), "sf_dit_forward: null output tensor");
  SF_CHECK(a->k_cache_host && a->v_cache_host && a->ck_cache_host && a->cv_cache_host, "sf_dit_forward: null cache table");
  SF_CHECK(a->attn_start >= 0 && a->attn_end > a->attn_start && a->attn_end <= a->cache_tokens, "sf_dit_forward: bad attention window [%d, %d) of %lld",
           a->attn_start, a->attn_end, (long long)a->cache_tokens);
  SF_CHECK(a->write_start >= 0 && (int64_t)a->write_start + L <= a->cache_tokens,
           "sf_dit_forward: KV cache overflow: write_start=%d + %d new tokens > capacity %lld", a->write_start, L, (long long)a->cache_tokens);
  SF_CHECK(a->write_start + L == a->attn_end, "sf_dit_forward: the new tokens must end the attention window");
  if (a->evict > 0) SF_CHECK(a->evict_scratch && a->keep >= 0, "sf_dit_forward: eviction needs evict_scratch");
  const Work ws = carve(m, a->workspace, B, F, a->lat_h, a->lat_w, G);
  SF_CHECK(a->workspace && a->workspace_bytes >= ws.total, "sf_dit_forward: workspace too small (%zu < %zu)", a->workspace_bytes, ws.total);
  SF_CHECK(!a->init_cross || a->prompt_embeds, "sf_dit_forward: init_cross needs prompt_embeds");

  const int Kp = m->in_dim * 4, Nh = m->out_dim * 4;
  const long cache_b = (long)a->cache_tokens * C;
  const long ctx_b = (long)m->text_len * C;

  // ---- patch embedding
  SF_TRY(sf_patchify(a->noisy, ws.cols, B, F, m->in_dim, a->lat_h, a->lat_w, stream));
  SF_TRY(gemm(ws.cols, Kp, m->patch_w, m->patch_b, ws.x, C, M, C, Kp, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));

  // ---- pose conditioning of the fork: x += pose_proj(add_condition)  (causal_model.py:786-819)
  if (a->add_condition) {
    if (m->pose_w) {
      SF_CHECK(m->pose_dim > 0, "sf_dit_forward: pose_proj weights without pose_dim");
      SF_TRY(gemm(a->add_condition, m->pose_dim, m->pose_w, m->pose_b, ws.x, C, M, C, m->pose_dim, SF_EPI_BIAS_RESID, ws.x, C, nullptr, nullptr, 0, 1, stream));
    } else {
      // dim == 5120: `pose_proj = nn.Identity()` (causal_model.py:500-503): x += add_condition, [M, C] bf16, fp32 add, one rounding
      SF_CHECK(m->pose_dim == C, "sf_dit_forward: add_condition without pose_proj weights needs pose_dim == dim (%d != %d)", m->pose_dim, C);
      const void* terms[2] = {ws.x, a->add_condition};
      const float ones[2] = {1.0f, 1.0f};
      SF_TRY(sf_lincomb_bf16(ws.x, terms, ones, 2, (int64_t)M * C, stream));
    }
  }

  // ---- time embeddings: e [BG, C], e0 [BG, 6C]
  SF_TRY(sf_sinusoid_embedding(a->timestep, a->t_is_int64, ws.sin, BG, m->freq_dim, stream));
  SF_TRY(sf_small_linear(ws.sin, m->time0_w, m->time0_b, ws.etmp, BG, C, m->freq_dim, 0, 1, stream));
  SF_TRY(sf_small_linear(ws.etmp, m->time2_w, m->time2_b, ws.e, BG, C, C, 0, 0, stream));
  SF_TRY(sf_small_linear(ws.e, m->tproj_w, m->tproj_b, ws.e0, BG, 6 * C, C, 1, 0, stream));

  // ---- text embedding + cross-attention K/V: once per prompt (the reference recomputes the text
  // MLP on every forward although only the first call consumes it, causal_model.py:837-842)
  if (a->init_cross) {
    const int T = B * m->text_len;
    SF_TRY(gemm(a->prompt_embeds, m->text_dim, m->text0_w, m->text0_b, ws.ctx1, C, T, C, m->text_dim, SF_EPI_BIAS_GELU, nullptr, 0,
                nullptr, nullptr, 0, 1, stream));
    SF_TRY(gemm(ws.ctx1, C, m->text2_w, m->text2_b, ws.ctx, C, T, C, C, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));
    for (int l = 0; l < m->num_layers; ++l) {
      const sf_layer_weights& lw = m->layers_host[l];
      SF_TRY(gemm(ws.ctx, C, lw.ckv_w, lw.ckv_b, a->ck_cache_host[l], C, T, C, C, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));
      SF_TRY(sf_rmsnorm(a->ck_cache_host[l], C, lw.cnorm_k_w, a->ck_cache_host[l], C, T, C, m->eps, stream));
      SF_TRY(gemm(ws.ctx, C, bptr(lw.ckv_w, (size_t)C * C), bptr(lw.ckv_b, C), a->cv_cache_host[l], C, T, C, C, SF_EPI_BIAS, nullptr, 0,
                  nullptr, nullptr, 0, 1, stream));
    }
  }

  // ---- transformer blocks
  for (int l = 0; l < m->num_layers; ++l) {
    const sf_layer_weights& lw = m->layers_host[l];
    const void* mod = lw.modulation;
    // self attention
    SF_TRY(sf_layernorm_modulate(ws.x, ws.xn, M, C, m->eps, bptr(mod, 0), bptr(mod, C), bptr(ws.e0, 0), bptr(ws.e0, C), 6L * C, rpg, stream));
    SF_TRY(gemm(ws.xn, C, lw.qkv_w, lw.qkv_b, ws.qkv, 3 * C, M, 3 * C, C, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));
    if (a->evict > 0) {
      SF_TRY(sf_kv_evict(a->k_cache_host[l], B, a->cache_tokens, C, a->sink_tokens, a->evict, a->keep, a->evict_scratch, a->evict_scratch_bytes, stream));
      SF_TRY(sf_kv_evict(a->v_cache_host[l], B, a->cache_tokens, C, a->sink_tokens, a->evict, a->keep, a->evict_scratch, a->evict_scratch_bytes, stream));
    }
    SF_TRY(sf_qkv_norm_rope_cache(ws.qkv, lw.norm_q_w, lw.norm_k_w, ws.q, a->k_cache_host[l], a->v_cache_host[l], m->rope_cos, m->rope_sin,
                                  B, F, h, w, C, m->num_heads, a->cache_tokens, a->write_start, a->start_frame, m->eps, stream));
    if (a->cache_only && l == m->num_layers - 1) {           // nothing downstream of this K/V write is read
      if (a->kv_index_out) SF_TRY(sf_internal_write_kv_indices(a->kv_index_out, m->num_layers, a->global_end, a->attn_end, stream));
      return 0;
    }
    SF_TRY(sf_attention(ws.q, bptr(a->k_cache_host[l], (size_t)a->attn_start * C), bptr(a->v_cache_host[l], (size_t)a->attn_start * C), ws.att,
                        B, m->num_heads, L, a->attn_end - a->attn_start, C, (long)L * C, C, cache_b, C, (long)L * C, stream));
    SF_TRY(gemm(ws.att, C, lw.o_w, lw.o_b, ws.x, C, M, C, C, SF_EPI_BIAS_GATE_RESID, ws.x, C, bptr(mod, 2 * (size_t)C), bptr(ws.e0, 2 * (size_t)C),
                6L * C, rpg, stream));
    // cross attention
    SF_TRY(sf_layernorm_affine(ws.x, lw.norm3_w, lw.norm3_b, ws.xn, M, C, m->eps, stream));
    SF_TRY(gemm(ws.xn, C, lw.cq_w, lw.cq_b, ws.q, C, M, C, C, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));
    SF_TRY(sf_rmsnorm(ws.q, C, lw.cnorm_q_w, ws.q, C, M, C, m->eps, stream));
    SF_TRY(sf_attention(ws.q, a->ck_cache_host[l], a->cv_cache_host[l], ws.att, B, m->num_heads, L, m->text_len, C, (long)L * C, C, ctx_b, C,
                        (long)L * C, stream));
    SF_TRY(gemm(ws.att, C, lw.co_w, lw.co_b, ws.x, C, M, C, C, SF_EPI_BIAS_RESID, ws.x, C, nullptr, nullptr, 0, 1, stream));
    // feed forward
    SF_TRY(sf_layernorm_modulate(ws.x, ws.xn, M, C, m->eps, bptr(mod, 3 * (size_t)C), bptr(mod, 4 * (size_t)C), bptr(ws.e0, 3 * (size_t)C),
                                 bptr(ws.e0, 4 * (size_t)C), 6L * C, rpg, stream));
    SF_TRY(gemm(ws.xn, C, lw.ffn0_w, lw.ffn0_b, ws.hbuf, m->ffn_dim, M, m->ffn_dim, C, SF_EPI_BIAS_GELU, nullptr, 0, nullptr, nullptr, 0, 1, stream));
    SF_TRY(gemm(ws.hbuf, m->ffn_dim, lw.ffn2_w, lw.ffn2_b, ws.x, C, M, C, m->ffn_dim, SF_EPI_BIAS_GATE_RESID, ws.x, C, bptr(mod, 5 * (size_t)C),
                bptr(ws.e0, 5 * (size_t)C), 6L * C, rpg, stream));
  }

  // ---- head (modulated by e, not e0: causal_model.py:890, :364-366), unpatchify, flow -> x0
  SF_TRY(sf_layernorm_modulate(ws.x, ws.xn, M, C, m->eps, bptr(m->head_mod, 0), bptr(m->head_mod, C), ws.e, ws.e, (long)C, rpg, stream));
  SF_TRY(gemm(ws.xn, C, m->head_w, m->head_b, ws.headout, Nh, M, Nh, C, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));
  SF_TRY(sf_unpatchify_x0(ws.headout, a->noisy, a->timestep, a->t_is_int64, m->sched_sigmas, m->sched_timesteps, m->n_table, a->flow_out,
                          a->x0_out, B, F, G, m->out_dim, a->lat_h, a->lat_w, stream));
  if (a->kv_index_out) SF_TRY(sf_internal_write_kv_indices(a->kv_index_out, m->num_layers, a->global_end, a->attn_end, stream));
  return 0;
}
