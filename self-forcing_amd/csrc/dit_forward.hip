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

// ------------------------------------------------------------------------------------------
// The pass sequencer.  `np` passes (1, or 2 for sf_dit_forward_pair) over the SAME model, caches and latent geometry run
// as ONE batch of np * B samples through everything that is row-wise -- patch embedding, time MLPs, norms, all ten GEMMs
// of a layer, the head -- and one after the other through what touches the KV cache (eviction, K / V write, attention:
// pass p + 1's write and eviction must not be visible to pass p's attention, exactly as when the passes run as separate
// calls).  Row-wise kernels compute every output row from that row alone and every GEMM tiling keeps the k order per
// output element, so the results are bit-identical to separate calls; what changes is M (two passes of 4680 tokens give
// the GEMMs 9360 rows: 1040-1440 instead of 810-1160 TFLOP/s) and the launch count.
// cache_only passes must come first; past the LAST layer's K / V write only the remaining passes' rows continue.
static int forward_passes(const sf_model* m, const sf_forward_args* const* ps, int np, void* stream) {
  const sf_forward_args* a = ps[0];
  SF_CHECK(m && a, "sf_dit_forward: null argument");
  SF_CHECK(m->layers_host && m->num_layers > 0, "sf_dit_forward: model has no layers");
  SF_CHECK(m->dim == m->num_heads * 128, "sf_dit_forward: head_dim must be 128 (dim=%d heads=%d)", m->dim, m->num_heads);
  SF_CHECK(a->batch > 0 && a->frames > 0 && a->groups > 0, "sf_dit_forward: empty input");
  SF_CHECK(a->lat_h % 2 == 0 && a->lat_w % 2 == 0, "sf_dit_forward: latent size must be even");
  const int B = a->batch, F = a->frames, h = a->lat_h / 2, w = a->lat_w / 2;
  const int L = F * h * w, M = B * L, C = m->dim, G = a->groups, BG = B * G;
  const int Mt = np * M, BGt = np * BG;                    // rows / modulation groups of the whole batch of passes
  SF_CHECK(F % G == 0, "sf_dit_forward: frames=%d not divisible by timestep groups=%d", F, G);
  const int rpg = L / G;
  SF_CHECK(BGt <= 32, "sf_dit_forward: passes*batch*groups=%d exceeds the small-linear limit of 32", BGt);
  int first_full = np;                                      // first pass that runs to the end
  for (int p = 0; p < np; ++p) {
    const sf_forward_args* q = ps[p];
    SF_CHECK(q, "sf_dit_forward: null pass");
    SF_CHECK(q->batch == B && q->frames == F && q->lat_h == a->lat_h && q->lat_w == a->lat_w && q->groups == G,
             "sf_dit_forward: the passes of one call must share batch, frames, latent size and timestep groups");
    SF_CHECK(q->k_cache_host == a->k_cache_host && q->v_cache_host == a->v_cache_host && q->ck_cache_host == a->ck_cache_host &&
             q->cv_cache_host == a->cv_cache_host && q->cache_tokens == a->cache_tokens, "sf_dit_forward: the passes of one call must share their caches");
    SF_CHECK(q->noisy && q->timestep, "sf_dit_forward: null tensor");
    SF_CHECK(q->cache_only || (q->flow_out && q->x0_out), "sf_dit_forward: null output tensor");
    SF_CHECK(q->attn_start >= 0 && q->attn_end > q->attn_start && q->attn_end <= q->cache_tokens, "sf_dit_forward: bad attention window [%d, %d) of %lld",
             q->attn_start, q->attn_end, (long long)q->cache_tokens);
    SF_CHECK(q->write_start >= 0 && (int64_t)q->write_start + L <= q->cache_tokens,
             "sf_dit_forward: KV cache overflow: write_start=%d + %d new tokens > capacity %lld", q->write_start, L, (long long)q->cache_tokens);
    SF_CHECK(q->write_start + L == q->attn_end, "sf_dit_forward: the new tokens must end the attention window");
    if (q->evict > 0) SF_CHECK(q->evict_scratch && q->keep >= 0, "sf_dit_forward: eviction needs evict_scratch");
    SF_CHECK(np == 1 || !q->init_cross, "sf_dit_forward_pair: the cross-attention cache must be initialised by an earlier single pass");
    if (!q->cache_only && first_full == np) first_full = p;
    SF_CHECK(!(q->cache_only && first_full != np), "sf_dit_forward_pair: cache_only passes must come first");
  }
  SF_CHECK(a->k_cache_host && a->v_cache_host && a->ck_cache_host && a->cv_cache_host, "sf_dit_forward: null cache table");
  const Work ws = carve(m, a->workspace, np * B, F, a->lat_h, a->lat_w, G);
  SF_CHECK(a->workspace && a->workspace_bytes >= ws.total, "sf_dit_forward: workspace too small (%zu < %zu)", a->workspace_bytes, ws.total);
  SF_CHECK(!a->init_cross || a->prompt_embeds, "sf_dit_forward: init_cross needs prompt_embeds");

  const int Kp = m->in_dim * 4, Nh = m->out_dim * 4;
  const long cache_b = (long)a->cache_tokens * C;
  const long ctx_b = (long)m->text_len * C;
  const sf_forward_args* last = ps[np - 1];
  auto finish_indices = [&]() -> int {
    if (last->kv_index_out) return sf_internal_write_kv_indices(last->kv_index_out, m->num_layers, last->global_end, last->attn_end, stream);
    return 0;
  };

  // ---- patch embedding (rows of pass p at [p M, (p + 1) M))
  for (int p = 0; p < np; ++p)
    SF_TRY(sf_patchify(ps[p]->noisy, (void*)bptr(ws.cols, (size_t)p * M * Kp), B, F, m->in_dim, a->lat_h, a->lat_w, stream));
  SF_TRY(gemm(ws.cols, Kp, m->patch_w, m->patch_b, ws.x, C, Mt, C, Kp, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));

  // ---- pose conditioning of the fork: x += pose_proj(add_condition)  (causal_model.py:786-819), per pass
  for (int p = 0; p < np; ++p) {
    if (!ps[p]->add_condition) continue;
    void* xp = (void*)bptr(ws.x, (size_t)p * M * C);
    if (m->pose_w) {
      SF_CHECK(m->pose_dim > 0, "sf_dit_forward: pose_proj weights without pose_dim");
      SF_TRY(gemm(ps[p]->add_condition, m->pose_dim, m->pose_w, m->pose_b, xp, C, M, C, m->pose_dim, SF_EPI_BIAS_RESID, xp, C, nullptr, nullptr, 0, 1, stream));
    } else {
      // dim == 5120: `pose_proj = nn.Identity()` (causal_model.py:500-503): x += add_condition, [M, C] bf16, fp32 add, one rounding
      SF_CHECK(m->pose_dim == C, "sf_dit_forward: add_condition without pose_proj weights needs pose_dim == dim (%d != %d)", m->pose_dim, C);
      const void* terms[2] = {xp, ps[p]->add_condition};
      const float ones[2] = {1.0f, 1.0f};
      SF_TRY(sf_lincomb_bf16(xp, terms, ones, 2, (int64_t)M * C, stream));
    }
  }

  // ---- time embeddings: e [BGt, C], e0 [BGt, 6C]  (group rows pass-major, like the token rows)
  for (int p = 0; p < np; ++p)
    SF_TRY(sf_sinusoid_embedding(ps[p]->timestep, ps[p]->t_is_int64, (void*)bptr(ws.sin, (size_t)p * BG * m->freq_dim), BG, m->freq_dim, stream));
  SF_TRY(sf_small_linear(ws.sin, m->time0_w, m->time0_b, ws.etmp, BGt, C, m->freq_dim, 0, 1, stream));
  SF_TRY(sf_small_linear(ws.etmp, m->time2_w, m->time2_b, ws.e, BGt, C, C, 0, 0, stream));
  SF_TRY(sf_small_linear(ws.e, m->tproj_w, m->tproj_b, ws.e0, BGt, 6 * C, C, 1, 0, stream));

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
    const bool last_layer = l == m->num_layers - 1;
    // self attention: LN + q|k|v projection for every pass at once ...
    SF_TRY(sf_layernorm_modulate(ws.x, ws.xn, Mt, C, m->eps, bptr(mod, 0), bptr(mod, C), bptr(ws.e0, 0), bptr(ws.e0, C), 6L * C, rpg, stream));
    SF_TRY(gemm(ws.xn, C, lw.qkv_w, lw.qkv_b, ws.qkv, 3 * C, Mt, 3 * C, C, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));
    // ... then pass by pass through the cache: eviction, K / V write, attention over the pass's own window
    for (int p = 0; p < np; ++p) {
      const sf_forward_args* q = ps[p];
      const size_t roff = (size_t)p * M;
      if (q->evict > 0) {
        SF_TRY(sf_kv_evict(q->k_cache_host[l], B, q->cache_tokens, C, q->sink_tokens, q->evict, q->keep, q->evict_scratch, q->evict_scratch_bytes, stream));
        SF_TRY(sf_kv_evict(q->v_cache_host[l], B, q->cache_tokens, C, q->sink_tokens, q->evict, q->keep, q->evict_scratch, q->evict_scratch_bytes, stream));
      }
      SF_TRY(sf_qkv_norm_rope_cache(bptr(ws.qkv, roff * 3 * C), lw.norm_q_w, lw.norm_k_w, (void*)bptr(ws.q, roff * C), q->k_cache_host[l], q->v_cache_host[l],
                                    m->rope_cos, m->rope_sin, B, F, h, w, C, m->num_heads, q->cache_tokens, q->write_start, q->start_frame, m->eps, stream));
      if (q->cache_only && last_layer) continue;            // nothing downstream of this K/V write is read
      SF_TRY(sf_attention(bptr(ws.q, roff * C), bptr(q->k_cache_host[l], (size_t)q->attn_start * C), bptr(q->v_cache_host[l], (size_t)q->attn_start * C),
                          (void*)bptr(ws.att, roff * C), B, m->num_heads, L, q->attn_end - q->attn_start, C, (long)L * C, C, cache_b, C, (long)L * C, stream));
    }
    // rows that go on: all of them, except behind the last layer's K / V write, where the cache_only passes are done
    const int p0 = last_layer ? first_full : 0;
    if (p0 == np) return finish_indices();
    const size_t r0 = (size_t)p0 * M;
    const int Mr = (np - p0) * M;
    void* x = (void*)bptr(ws.x, r0 * C);
    void* xn = (void*)bptr(ws.xn, r0 * C);
    void* qb = (void*)bptr(ws.q, r0 * C);
    void* att = (void*)bptr(ws.att, r0 * C);
    void* hb = (void*)bptr(ws.hbuf, r0 * m->ffn_dim);
    const void* e0r = bptr(ws.e0, (size_t)p0 * BG * 6 * C);
    SF_TRY(gemm(att, C, lw.o_w, lw.o_b, x, C, Mr, C, C, SF_EPI_BIAS_GATE_RESID, x, C, bptr(mod, 2 * (size_t)C), bptr(e0r, 2 * (size_t)C), 6L * C, rpg, stream));
    // cross attention
    SF_TRY(sf_layernorm_affine(x, lw.norm3_w, lw.norm3_b, xn, Mr, C, m->eps, stream));
    SF_TRY(gemm(xn, C, lw.cq_w, lw.cq_b, qb, C, Mr, C, C, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));
    SF_TRY(sf_rmsnorm(qb, C, lw.cnorm_q_w, qb, C, Mr, C, m->eps, stream));
    for (int p = p0; p < np; ++p)                            // every pass reads the same text K / V: one launch per pass
      SF_TRY(sf_attention(bptr(ws.q, (size_t)p * M * C), a->ck_cache_host[l], a->cv_cache_host[l], (void*)bptr(ws.att, (size_t)p * M * C), B, m->num_heads, L,
                          m->text_len, C, (long)L * C, C, ctx_b, C, (long)L * C, stream));
    SF_TRY(gemm(att, C, lw.co_w, lw.co_b, x, C, Mr, C, C, SF_EPI_BIAS_RESID, x, C, nullptr, nullptr, 0, 1, stream));
    // feed forward
    SF_TRY(sf_layernorm_modulate(x, xn, Mr, C, m->eps, bptr(mod, 3 * (size_t)C), bptr(mod, 4 * (size_t)C), bptr(e0r, 3 * (size_t)C),
                                 bptr(e0r, 4 * (size_t)C), 6L * C, rpg, stream));
    SF_TRY(gemm(xn, C, lw.ffn0_w, lw.ffn0_b, hb, m->ffn_dim, Mr, m->ffn_dim, C, SF_EPI_BIAS_GELU, nullptr, 0, nullptr, nullptr, 0, 1, stream));
    SF_TRY(gemm(hb, m->ffn_dim, lw.ffn2_w, lw.ffn2_b, x, C, Mr, C, m->ffn_dim, SF_EPI_BIAS_GATE_RESID, x, C, bptr(mod, 5 * (size_t)C),
                bptr(e0r, 5 * (size_t)C), 6L * C, rpg, stream));
  }

  // ---- head (modulated by e, not e0: causal_model.py:890, :364-366), unpatchify, flow -> x0: the passes that run to the end
  {
    const size_t r0 = (size_t)first_full * M;
    const int Mr = (np - first_full) * M;
    const void* er = bptr(ws.e, (size_t)first_full * BG * C);
    SF_TRY(sf_layernorm_modulate(bptr(ws.x, r0 * C), (void*)bptr(ws.xn, r0 * C), Mr, C, m->eps, bptr(m->head_mod, 0), bptr(m->head_mod, C), er, er, (long)C, rpg, stream));
    SF_TRY(gemm(bptr(ws.xn, r0 * C), C, m->head_w, m->head_b, (void*)bptr(ws.headout, r0 * Nh), Nh, Mr, Nh, C, SF_EPI_BIAS, nullptr, 0, nullptr, nullptr, 0, 1, stream));
    for (int p = first_full; p < np; ++p)
      SF_TRY(sf_unpatchify_x0(bptr(ws.headout, (size_t)p * M * Nh), ps[p]->noisy, ps[p]->timestep, ps[p]->t_is_int64, m->sched_sigmas, m->sched_timesteps, m->n_table,
                              ps[p]->flow_out, ps[p]->x0_out, B, F, G, m->out_dim, a->lat_h, a->lat_w, stream));
  }
  return finish_indices();
}

extern "C" int sf_dit_forward(const sf_model* m, const sf_forward_args* a, void* stream) {
  SF_CHECK(m && a, "sf_dit_forward: null argument");
  const sf_forward_args* one[1] = {a};
  return forward_passes(m, one, 1, stream);
}

extern "C" int sf_dit_forward_pair(const sf_model* m, const sf_forward_args* context_pass, const sf_forward_args* next_pass, void* stream) {
  SF_CHECK(m && context_pass && next_pass, "sf_dit_forward_pair: null argument");
  SF_CHECK(context_pass->workspace == next_pass->workspace && context_pass->workspace_bytes == next_pass->workspace_bytes,
           "sf_dit_forward_pair: both passes name the same workspace (sized for 2 x batch)");
  const sf_forward_args* two[2] = {context_pass, next_pass};
  return forward_passes(m, two, 2, stream);
}
