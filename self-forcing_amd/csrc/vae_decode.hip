// The Wan VAE's per-latent-frame decode loop -- WanVAE_.decode / cached_decode (wan/modules/vae.py:556-593) ->
// Decoder3d.forward with feat_cache (vae.py:423-472) -- as a host call that enqueues every kernel on the caller's stream,
// for ONE OR SEVERAL consecutive latent frames at a time: the causal convolutions see the same inputs whether the frames
// arrive one per call (as the reference iterates) or together, every output element is computed by the same arithmetic in
// the same order, so the result is bit-identical -- but a group of frames fills the chip at the low-resolution stages
// (one frame of 60 x 104 is 28 patches) and leaves shorter tails at the others.
//
// Host-side only (no kernels here).  The reference threads a list of 32 cache tensors and an index
// counter through the modules and re-concatenates [cache, x] in front of every convolution; here each
// cached convolution owns ONE input volume in the per-stream state: its producer (the RMS-norm/SiLU kernel, or the
// previous convolution's epilogue) writes the T new frames behind the two history frames, the convolution gathers its
// temporal taps from frames t, t+1, t+2, and the last two frames become the next call's history (the cache update of
// vae.py:206-216; for T = 1 that is the "borrow the last frame of the previous cache" branch).  The volume holds
// 2 + K T frames (K = `window_frames` latent frames, T = the stage's frames per latent frame) and the window
// [history | new frames] SLIDES through it: the next call's history is where this call's last two frames already are;
// when the next window would not fit the caller restarts at slot 0 and the two history frames are copied there first
// (round 1 copied them after every convolution: 2756 copies and 5 % of a clip's decode time).  The window's position is
// passed by the caller (`window`, `history_at`: no host state here).  A zeroed history IS the reference's zero
// padding of the first chunk, so there is no first-chunk special case in the convolutions.  The two
// quirks of Resample's bookkeeping (vae.py:104-132) are kept: the first chunk after a reset skips the
// time convolution (one output frame), and its features never enter that convolution's history.
#include <cmath>
#include <cstring>
#include "sf_common.h"
#include "../../include/sf_hip.h"

namespace {

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

struct BlockBufs { char *a1, *a2; };

struct Plan {
  int n_stages, rps, K;
  int H[SF_VAE_MAX_STAGES], W[SF_VAE_MAX_STAGES], Tmax[SF_VAE_MAX_STAGES];
  // state
  char* c1_in;
  BlockBufs mid0, mid2;
  BlockBufs blk[SF_VAE_MAX_STAGES * 8];
  char* tc[SF_VAE_MAX_STAGES];
  char* head_in;
  size_t state_total;
  // scratch
  char *xi[SF_VAE_MAX_STAGES], *x[SF_VAE_MAX_STAGES], *ty[SF_VAE_MAX_STAGES];
  char *y1, *sc;
  char *att_xn, *att_qk, *att_vt, *att_s, *att_p, *att_o;
  int att_npad;
  size_t scratch_total;
};

inline size_t vol(int T, int H, int W, int C) { return (size_t)T * H * W * C * 2; }

// Sliding history window of a cached convolution's input volume (capacity 2 + K * Tmax frames, K latent frames).
// A lap = the calls between two restarts at slot 0; slot q of a lap holds the frames of the lap's q-th latent frame.
// The lap that begins at the reset is special: its slot 0 is the first chunk, which has ONE frame at every stage
// (vae.py:109-111) and never enters a time convolution's volume (the quirk of vae.py:104-132).
inline int hist_frames(int K, int Tmax) { return 2 + K * Tmax; }
inline int vol_off(int lap_start, int slot, int Tmax, bool time_conv) {   // first history frame of the window that starts at `slot`
  if (lap_start != 0 || slot == 0) return slot * Tmax;
  return time_conv ? (slot - 1) * Tmax : 1 + (slot - 1) * Tmax;
}

struct Call {                 // one sf_vae_decode_frames call
  int n, F, window, history_at;
  hipStream_t s;
  bool first_chunk() const { return n == 0; }
  int off(int Tmax, bool tc = false) const { return vol_off(n - window, window, Tmax, tc); }
};

// the two history frames of a volume are where the previous call left them (slot `history_at` of ITS lap); a call that
// restarts the window copies them to the front first (frame by frame: the ranges may overlap by one frame)
int place_history(const Call& c, char* buf, int Tmax, size_t frame_bytes, bool tc = false) {
  if (c.history_at == c.window) return 0;
  if (tc && c.n - c.history_at == 0 && c.history_at <= 1) return 0;   // only the first chunk so far: this volume is still all zero
  const int src = vol_off(c.n - c.history_at, c.history_at, Tmax, tc), dst = c.off(Tmax, tc);
  for (int k = 0; k < 2; ++k) {
    hipError_t e = hipMemcpyAsync(buf + (size_t)(dst + k) * frame_bytes, buf + (size_t)(src + k) * frame_bytes, frame_bytes, hipMemcpyDeviceToDevice, c.s);
    SF_CHECK(e == hipSuccess, "sf_vae: history copy failed: %s", hipGetErrorString(e));
  }
  return 0;
}

const sf_vae_resblock& res_at(const sf_vae_model* m, int stage, int j) { return m->res_host[stage * m->res_per_stage + j]; }

Plan make_plan(const sf_vae_model* m, void* state, void* scratch, int h, int w, int K) {
  Plan p;
  memset(&p, 0, sizeof(p));
  p.n_stages = m->n_stages;
  p.rps = m->res_per_stage;
  p.K = K;
  const int Fmax = K - 1;        // latent frames per call (the lap that begins at the reset holds the first chunk + Fmax)
  int T = 1;
  for (int i = 0; i < m->n_stages; ++i) {
    p.H[i] = h << i;
    p.W[i] = w << i;
    p.Tmax[i] = T;
    if (i + 1 < m->n_stages && m->temporal_up[i]) T *= 2;
  }
  Carve st(state);
  p.c1_in = st.take(vol(hist_frames(K, 1), h, w, m->conv1.cin));
  const int C0 = m->conv1.cout;
  p.mid0.a1 = st.take(vol(hist_frames(K, 1), h, w, C0)); p.mid0.a2 = st.take(vol(hist_frames(K, 1), h, w, C0));
  p.mid2.a1 = st.take(vol(hist_frames(K, 1), h, w, C0)); p.mid2.a2 = st.take(vol(hist_frames(K, 1), h, w, C0));
  for (int i = 0; i < m->n_stages; ++i) {
    for (int j = 0; j < m->res_per_stage; ++j) {
      const sf_vae_resblock& r = res_at(m, i, j);
      BlockBufs& b = p.blk[i * m->res_per_stage + j];
      b.a1 = st.take(vol(hist_frames(K, p.Tmax[i]), p.H[i], p.W[i], r.conv1.cin));
      b.a2 = st.take(vol(hist_frames(K, p.Tmax[i]), p.H[i], p.W[i], r.conv2.cin));
    }
    p.tc[i] = (i + 1 < m->n_stages && m->time_conv[i].w) ? st.take(vol(hist_frames(K, p.Tmax[i]), p.H[i], p.W[i], m->time_conv[i].cin)) : nullptr;
  }
  const int L = m->n_stages - 1;
  p.head_in = st.take(vol(hist_frames(K, p.Tmax[L]), p.H[L], p.W[L], m->head_conv.cin));
  p.state_total = st.off;

  Carve sc(scratch);
  size_t y1_max = vol(Fmax, h, w, C0), sc_max = 256;
  for (int i = 0; i < m->n_stages; ++i) {
    const int cin = res_at(m, i, 0).conv1.cin, cout = res_at(m, i, 0).conv1.cout;
    p.xi[i] = sc.take(vol(Fmax * p.Tmax[i], p.H[i], p.W[i], i == 0 ? C0 : cin));
    p.x[i] = sc.take(vol(Fmax * p.Tmax[i], p.H[i], p.W[i], cout));
    p.ty[i] = (i + 1 < m->n_stages && m->time_conv[i].w) ? sc.take(vol(2 * Fmax * p.Tmax[i], p.H[i], p.W[i], m->time_conv[i].cin)) : nullptr;
    for (int j = 0; j < m->res_per_stage; ++j) {
      const sf_vae_resblock& r = res_at(m, i, j);
      const size_t v = vol(Fmax * p.Tmax[i], p.H[i], p.W[i], r.conv1.cout);
      if (v > y1_max) y1_max = v;
      if (r.shortcut.w && v > sc_max) sc_max = v;
    }
  }
  p.y1 = sc.take(y1_max);
  p.sc = sc.take(sc_max);
  const int n = h * w;
  p.att_npad = (n + 63) & ~63;
  p.att_xn = sc.take((size_t)n * C0 * 2);
  p.att_qk = sc.take((size_t)n * 2 * C0 * 2);
  p.att_vt = sc.take((size_t)C0 * p.att_npad * 2);
  p.att_s = sc.take((size_t)n * p.att_npad * 4);
  p.att_p = sc.take((size_t)n * p.att_npad * 2);
  p.att_o = sc.take((size_t)n * C0 * 2);
  p.scratch_total = sc.off;
  return p;
}

int check_model(const sf_vae_model* m, int h, int w, int K) {
  SF_CHECK(m != nullptr, "sf_vae: null model");
  SF_CHECK(K >= 2 && K <= 64, "sf_vae: window_frames %d (2..64: a call decodes up to window_frames - 1 latent frames)", K);
  SF_CHECK(m->n_stages >= 1 && m->n_stages <= SF_VAE_MAX_STAGES && m->res_per_stage >= 1 && m->res_per_stage <= 8, "sf_vae: bad stage counts");
  SF_CHECK(m->res_host && m->conv1.w && m->head_conv.w && m->latent_mean && m->latent_std && m->conv2_w && m->conv2_b, "sf_vae: model has null weights");
  SF_CHECK(h > 0 && w > 0 && (h * w) % 4 == 0, "sf_vae: latent size %dx%d (h*w must be a multiple of 4)", h, w);
  SF_CHECK(m->conv1.cout % 64 == 0, "sf_vae: decoder width %d must be a multiple of 64 (attention block GEMMs)", m->conv1.cout);
  SF_CHECK(m->z_dim > 0 && m->z_dim <= 32 && m->conv1.cin >= m->z_dim, "sf_vae: bad z_dim");
  return 0;
}

// every cached convolution's input volume: fn(buffer, frames per latent frame at its stage, bytes per frame, is it a
// time convolution's volume)
template <typename Fn>
int for_each_volume(const sf_vae_model* m, const Plan& p, int h, int w, Fn fn) {
  const int C0 = m->conv1.cout, L = m->n_stages - 1;
  int rc = fn(p.c1_in, 1, vol(1, h, w, m->conv1.cin), false);
  if (rc) return rc;
  char* mids[4] = {p.mid0.a1, p.mid0.a2, p.mid2.a1, p.mid2.a2};
  for (char* b : mids)
    if ((rc = fn(b, 1, vol(1, h, w, C0), false)) != 0) return rc;
  for (int i = 0; i < m->n_stages; ++i) {
    for (int j = 0; j < m->res_per_stage; ++j) {
      const sf_vae_resblock& r = res_at(m, i, j);
      const BlockBufs& b = p.blk[i * m->res_per_stage + j];
      if ((rc = fn(b.a1, p.Tmax[i], vol(1, p.H[i], p.W[i], r.conv1.cin), false)) != 0) return rc;
      if ((rc = fn(b.a2, p.Tmax[i], vol(1, p.H[i], p.W[i], r.conv2.cin), false)) != 0) return rc;
    }
    if (p.tc[i] && (rc = fn(p.tc[i], p.Tmax[i], vol(1, p.H[i], p.W[i], m->time_conv[i].cin), true)) != 0) return rc;
  }
  return fn(p.head_in, p.Tmax[L], vol(1, p.H[L], p.W[L], m->head_conv.cin), false);
}

#define SF_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != 0) return rc__; \
  } while (0)

// RMS_norm + SiLU of a convolution's output can ride in its epilogue (second output of the halo kernel) when the
// convolution is 3 x 3 spatial with 96 or 192 output channels at a resolution the halo kernel takes
struct NormOut { void* dst; const void* gamma; int ld; int frame_off; };   // dst: base of the consumer's input volume; its new frames start at frame_off

bool can_fuse_norm(const sf_vae_conv& c, int H, int W) { return c.kh == 3 && c.kw == 3 && (c.cout == 96 || c.cout == 192) && H >= 16 && W >= 16; }

int conv(const sf_vae_conv& c, const void* x, int Tout, int H, int W, int upsample, int t_off, void* out, int ldo, int out_frame0,
         int interleave_c, int epi, const void* resid, int ldr, float* out_f32, void* stream, const NormOut* norm = nullptr) {
  sf_conv_args a;
  memset(&a, 0, sizeof(a));
  if (norm) { a.norm_out = norm->dst; a.norm_gamma = norm->gamma; a.norm_ld = norm->ld; a.norm_frame_offset = norm->frame_off; }
  a.x = x; a.w = c.w; a.bias = c.bias; a.out = out; a.resid = resid; a.out_f32 = out_f32;
  a.Tout = Tout; a.H = H; a.W = W; a.Hin = upsample ? H / 2 : H; a.Win = upsample ? W / 2 : W;
  a.Cin = c.cin; a.Cout = c.cout; a.kt = c.kt; a.kh = c.kh; a.kw = c.kw; a.upsample = upsample; a.t_in_offset = t_off;
  a.ldw = c.ldw; a.ldo = ldo; a.ldr = ldr; a.out_frame_offset = out_frame0; a.interleave_c = interleave_c; a.epilogue = epi;
  return sf_conv_igemm(&a, stream);
}

int gemm(const void* a, int lda, const void* w, int ldw, const void* bias, void* out, int ldo, int M, int N, int K, int epi,
         const void* resid, int ldr, void* stream) {
  sf_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.a = a; g.w = w; g.bias = bias; g.out = out; g.resid = resid; g.rows_per_group = 1;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldo = ldo; g.ldr = ldr; g.epilogue = epi;
  return sf_gemm_bf16(&g, stream);
}

// ResidualBlock.forward (vae.py:202-221) on T frames of H x W.  `in_normed`: the producer of x_in already wrote
// SiLU(RMS_norm(x_in)) into conv1's input volume (fused epilogue); `next`: where (and with which gamma) this block's
// output should ALSO be written normalised -- the next block's conv1 input or the head's --, if its conv2 can do that.
// Returns through *out_normed whether it did.
int resblock(const sf_vae_resblock& r, const BlockBufs& b, const Plan& p, const char* x_in, char* out, const Call& cl, int T, int Tmax, int H, int W,
             void* stream, bool in_normed = false, const NormOut* next = nullptr, bool* out_normed = nullptr) {
  const long rows = (long)T * H * W;
  const int cin = r.conv1.cin, cout = r.conv1.cout;
  const size_t f1 = vol(1, H, W, cin), f2 = vol(1, H, W, cout);
  const int c = cl.off(Tmax);                         // both volumes of the block slide alike
  if (!in_normed) SF_TRY(sf_rmsnorm_silu_cl(x_in, r.gamma1, b.a1 + (size_t)(c + 2) * f1, rows, cin, 1, stream));
  if (can_fuse_norm(r.conv1, H, W)) {   // conv1's raw output is only ever read by the norm in front of conv2
    const NormOut n2 = {b.a2, r.gamma2, cout, c + 2};
    SF_TRY(conv(r.conv1, b.a1, T, H, W, 0, c, nullptr, cout, 0, 0, SF_CONV_BIAS, nullptr, 0, nullptr, stream, &n2));
  } else {
    SF_TRY(conv(r.conv1, b.a1, T, H, W, 0, c, p.y1, cout, 0, 0, SF_CONV_BIAS, nullptr, 0, nullptr, stream));
    SF_TRY(sf_rmsnorm_silu_cl(p.y1, r.gamma2, b.a2 + (size_t)(c + 2) * f2, rows, cout, 1, stream));
  }
  const char* resid = x_in;
  if (r.shortcut.w) {
    SF_TRY(conv(r.shortcut, x_in, T, H, W, 0, 0, p.sc, cout, 0, 0, SF_CONV_BIAS, nullptr, 0, nullptr, stream));
    resid = p.sc;
  }
  const bool fuse_next = next && next->ld == r.conv2.cout && can_fuse_norm(r.conv2, H, W);
  SF_TRY(conv(r.conv2, b.a2, T, H, W, 0, c, out, cout, 0, 0, SF_CONV_BIAS_RESID, resid, cout, nullptr, stream, fuse_next ? next : nullptr));
  if (out_normed) *out_normed = fuse_next;
  return 0;
}

// AttentionBlock.forward (vae.py:241-264) on one frame of n = h*w positions, in place on x [n][C]
int attention_block(const sf_vae_model* m, const Plan& p, char* x, int n, int C, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int np = p.att_npad;
  SF_TRY(sf_rmsnorm_silu_cl(x, m->attn_gamma, p.att_xn, n, C, 0, stream));
  SF_TRY(gemm(p.att_xn, C, m->attn_qk_w, C, m->attn_qk_b, p.att_qk, 2 * C, n, 2 * C, C, SF_EPI_BIAS, nullptr, 0, stream));
  // V^T [C][np] = Wv . xn^T straight from the projection (no transpose pass); its bias is added after
  // the P.V product instead (softmax rows sum to one), the padded key columns stay zero
  hipError_t e = hipMemsetAsync(p.att_vt, 0, (size_t)C * np * 2, s);
  SF_CHECK(e == hipSuccess, "sf_vae: memset failed: %s", hipGetErrorString(e));
  SF_TRY(gemm(m->attn_v_w, C, p.att_xn, C, nullptr, p.att_vt, np, C, n, C, SF_EPI_BIAS, nullptr, 0, stream));
  SF_TRY(gemm(p.att_qk, 2 * C, p.att_qk + (size_t)C * 2, 2 * C, nullptr, p.att_s, np, n, n, C, SF_EPI_F32, nullptr, 0, stream));
  SF_TRY(sf_softmax_rows((const float*)p.att_s, np, p.att_p, np, n, n, np, 1.0f / sqrtf((float)C), stream));
  SF_TRY(gemm(p.att_p, np, p.att_vt, np, m->attn_v_b, p.att_o, C, n, C, np, SF_EPI_BIAS, nullptr, 0, stream));
  SF_TRY(gemm(p.att_o, C, m->attn_proj_w, C, m->attn_proj_b, x, C, n, C, C, SF_EPI_BIAS_RESID, x, C, stream));
  return 0;
}

}  // namespace

extern "C" size_t sf_vae_state_bytes(const sf_vae_model* m, int h, int w, int window_frames) {
  if (check_model(m, h, w, window_frames) != 0) return 0;
  return make_plan(m, nullptr, nullptr, h, w, window_frames).state_total;
}

extern "C" size_t sf_vae_scratch_bytes(const sf_vae_model* m, int h, int w, int window_frames) {
  if (check_model(m, h, w, window_frames) != 0) return 0;
  return make_plan(m, nullptr, nullptr, h, w, window_frames).scratch_total;
}

extern "C" int sf_vae_reset(const sf_vae_model* m, void* state, size_t state_bytes, int h, int w, int window_frames, void* stream) {
  SF_TRY(check_model(m, h, w, window_frames));
  const Plan p = make_plan(m, state, nullptr, h, w, window_frames);
  SF_CHECK(state && state_bytes >= p.state_total, "sf_vae_reset: state too small (%zu < %zu)", state_bytes, p.state_total);
  // Only the two history frames at the front of every volume are ever read before they are written (the first window
  // after a reset starts at slot 0; its new frames, and every later window's, are written by their producer first):
  // zeroing them -- 2 of 2 + K T frames -- is WanVAE_.clear_cache; the whole state would be ~20 GB of stores per clip.
  hipStream_t s = (hipStream_t)stream;
  return for_each_volume(m, p, h, w, [&](char* buf, int, size_t frame_bytes, bool) -> int {
    hipError_t e = hipMemsetAsync(buf, 0, 2 * frame_bytes, s);
    SF_CHECK(e == hipSuccess, "sf_vae_reset: memset failed: %s", hipGetErrorString(e));
    return 0;
  });
}

extern "C" int sf_vae_decode_frames(const sf_vae_model* m, void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                                    const void* latent_frames, int h, int w, int window_frames, int frame_index, int n_frames,
                                    int window, int history_at, float* pixels_out, void* stream) {
  SF_TRY(check_model(m, h, w, window_frames));
  SF_CHECK(latent_frames && pixels_out, "sf_vae_decode_frames: null tensor");
  const int K = window_frames, F = n_frames;
  SF_CHECK(frame_index >= 0 && frame_index < (1 << 24), "sf_vae_decode_frames: frame_index %d (latent frames decoded since the reset)", frame_index);
  SF_CHECK(F >= 1 && F <= K - 1, "sf_vae_decode_frames: n_frames %d (1..window_frames - 1 = %d)", F, K - 1);
  SF_CHECK(frame_index > 0 || (F == 1 && window == 0 && history_at == 0),
           "sf_vae_decode_frames: the frame that follows a reset is decoded alone at window 0 (it has one output frame: vae.py:109-111)");
  SF_CHECK(window >= 0 && window + F <= K && window <= frame_index, "sf_vae_decode_frames: window %d + %d frames does not fit %d slots", window, F, K);
  SF_CHECK(history_at == window || (window == 0 && history_at >= 1 && history_at <= K && history_at <= frame_index),
           "sf_vae_decode_frames: history_at %d (== window, or the previous lap's end when the window restarts at 0)", history_at);
  const Call cl = {frame_index, F, window, history_at, (hipStream_t)stream};
  const bool first_chunk = cl.first_chunk();
  const Plan p = make_plan(m, state, scratch, h, w, K);
  SF_CHECK(state && state_bytes >= p.state_total, "sf_vae_decode_frames: state too small (%zu < %zu)", state_bytes, p.state_total);
  SF_CHECK(scratch && scratch_bytes >= p.scratch_total, "sf_vae_decode_frames: scratch too small (%zu < %zu)", scratch_bytes, p.scratch_total);
  const int C0 = m->conv1.cout;
  const int L = m->n_stages - 1;
  const int Ch = m->head_conv.cin;

  // a restarted window: every volume's two history frames move to the front BEFORE any producer writes new frames
  if (history_at != window)
    SF_TRY(for_each_volume(m, p, h, w, [&](char* buf, int Tmax, size_t frame_bytes, bool tc) { return place_history(cl, buf, Tmax, frame_bytes, tc); }));

  // un-scale + conv2 (1x1x1) -> the new frames of decoder.conv1's input volume; conv1 (vae.py:425-438)
  const size_t f_in = vol(1, h, w, m->conv1.cin);
  const int c1 = cl.off(1);
  for (int f = 0; f < F; ++f)
    SF_TRY(sf_vae_prepare_latent((const char*)latent_frames + (size_t)f * m->z_dim * h * w * 2, m->latent_mean, m->latent_std, m->conv2_w, m->conv2_b,
                                 p.c1_in + (size_t)(c1 + 2 + f) * f_in, m->z_dim, h, w, m->conv1.cin, stream));
  SF_TRY(conv(m->conv1, p.c1_in, F, h, w, 0, c1, p.xi[0], C0, 0, 0, SF_CONV_BIAS, nullptr, 0, nullptr, stream));

  // middle (vae.py:441-445): res, attention (per frame), res -- on the latent-rate frames
  SF_CHECK(res_at(m, 0, 0).conv1.cin == C0 && res_at(m, 0, 0).conv1.cout == C0, "sf_vae_decode_frames: stage 0 must keep the decoder width");
  SF_TRY(resblock(m->mid0, p.mid0, p, p.xi[0], p.x[0], cl, F, 1, h, w, stream));
  for (int f = 0; f < F; ++f) SF_TRY(attention_block(m, p, p.x[0] + (size_t)f * vol(1, h, w, C0), h * w, C0, stream));
  SF_TRY(resblock(m->mid2, p.mid2, p, p.x[0], p.x[0], cl, F, 1, h, w, stream));

  // upsample stages (vae.py:448-452).  `normed`: the next consumer's input volume already holds SiLU(RMS_norm(cur))
  int T = F;
  const char* cur = p.x[0];   // the stage's running activation
  bool normed = false;
  for (int i = 0; i < m->n_stages; ++i) {
    const int H = p.H[i], W = p.W[i], Tmax = p.Tmax[i];
    const bool has_up = i + 1 < m->n_stages;
    const bool has_tc = has_up && m->time_conv[i].w != nullptr;
    SF_CHECK(T == (first_chunk ? 1 : F * Tmax), "sf_vae_decode_frames: stage %d expects %d frames, has %d", i, first_chunk ? 1 : F * Tmax, T);
    const int c_st = cl.off(Tmax), c_tc = cl.off(Tmax, true);
    for (int j = 0; j < m->res_per_stage; ++j) {
      const sf_vae_resblock& r = res_at(m, i, j);
      char* out = p.x[i];
      if (j == m->res_per_stage - 1 && has_tc && !first_chunk)
        out = p.tc[i] + (size_t)(c_tc + 2) * vol(1, H, W, m->time_conv[i].cin);   // feeds the time conv (not the first chunk: vae.py:104-132)
      // who reads this block's output through a norm: the next block of the stage, or (last block of the last stage) the head
      NormOut next = {nullptr, nullptr, 0, 0};
      if (j + 1 < m->res_per_stage) {
        const sf_vae_resblock& rn = res_at(m, i, j + 1);
        next = {p.blk[i * m->res_per_stage + j + 1].a1, rn.gamma1, rn.conv1.cin, c_st + 2};
      } else if (!has_up) {
        next = {p.head_in, m->head_gamma, Ch, c_st + 2};
      }
      bool out_normed = false;
      SF_TRY(resblock(r, p.blk[i * m->res_per_stage + j], p, cur, out, cl, T, Tmax, H, W, stream, normed, next.dst ? &next : nullptr, &out_normed));
      normed = out_normed;
      cur = out;
    }
    if (!has_up) break;
    const sf_vae_conv& uc = m->up_conv[i];
    const char* up_in = cur;
    int Tn = T;
    if (has_tc && !first_chunk) {
      // Resample 'upsample3d' (vae.py:112-137): (3,1,1) causal conv C -> 2C, channel halves -> frames 2t, 2t+1
      const sf_vae_conv& tcv = m->time_conv[i];
      SF_CHECK(tcv.cout == 2 * tcv.cin, "sf_vae_decode_frames: time conv must double the channels");
      SF_TRY(conv(tcv, p.tc[i], T, H, W, 0, c_tc, p.ty[i], tcv.cin, 0, tcv.cin, SF_CONV_BIAS, nullptr, 0, nullptr, stream));
      up_in = p.ty[i];
      Tn = 2 * T;
    }
    // nearest 2x + Conv2d 3x3 per frame (vae.py:139-141), fused; its output feeds the next stage's first block, whose
    // norm1 rides in this convolution's epilogue when the halo kernel takes it
    const sf_vae_resblock& rn = res_at(m, i + 1, 0);
    const NormOut nn = {p.blk[(i + 1) * m->res_per_stage].a1, rn.gamma1, rn.conv1.cin, cl.off(p.Tmax[i + 1]) + 2};
    const bool fuse = rn.conv1.cin == uc.cout && can_fuse_norm(uc, 2 * H, 2 * W);
    SF_TRY(conv(uc, up_in, Tn, 2 * H, 2 * W, 1, 0, p.xi[i + 1], uc.cout, 0, 0, SF_CONV_BIAS, nullptr, 0, nullptr, stream, fuse ? &nn : nullptr));
    normed = fuse;
    cur = p.xi[i + 1];
    T = Tn;
  }

  // head (vae.py:455-471): RMS-norm, SiLU, causal conv to 3 channels; float, clamp
  const int ch = cl.off(p.Tmax[L]);
  if (!normed)
    SF_TRY(sf_rmsnorm_silu_cl(cur, m->head_gamma, p.head_in + (size_t)(ch + 2) * vol(1, p.H[L], p.W[L], Ch), (long)T * p.H[L] * p.W[L], Ch, 1, stream));
  SF_TRY(conv(m->head_conv, p.head_in, T, p.H[L], p.W[L], 0, ch, nullptr, 0, 0, 0, SF_CONV_BIAS_CLAMP_F32, nullptr, 0, pixels_out, stream));
  return 0;
}
