/* sf_hip.h -- C-ABI of the MI355X (gfx950) implementation of Self-Forcing's chunk-wise
 * autoregressive denoising hot path.
 *
 * The reference (alazarteka/Self-Forcing) is pure Python/PyTorch and has no FFI layer of
 * its own: the path sits behind `WanDiffusionWrapper.forward` (utils/wan_wrapper.py:253-349)
 * and `CausalWanModel._forward_inference` (wan/modules/causal_model.py:725-893), whose heavy
 * ops go to third-party kernels (flash_attn, cuBLAS, cuDNN).  Each entry point below replaces
 * one of those op sequences; the citation on each says which.  `INTEGRATION.md` shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - every tensor is bf16 (uint16 storage) unless stated; row-major; strides in ELEMENTS;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream), allocates nothing, never synchronises, reads no environment variables and keeps no
 *     state between calls (apart from registering a kernel's LDS size with the runtime on first use);
 *   - return 0 on success, negative on error; `sf_last_error()` gives a thread-local message.
 */
#ifndef SF_HIP_H
#define SF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_HIP_ABI_VERSION 8

int sf_abi_version(void);
const char* sf_last_error(void);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue: out[M,N] = epi(a[M,K] @ w[N,K]^T + bias[N]).
 * Replaces nn.Linear (cuBLAS) + the elementwise ops that follow it in
 * wan/modules/causal_model.py:112-114 (q/k/v), :240 (o), :277-279 (ffn), :320/:331 (gated
 * residual), model.py:172-193 (cross-attention projections), causal_model.py:366 (head).
 * fp32 accumulation on the matrix cores; one rounding to bf16 at the store. */
enum sf_epilogue {
  SF_EPI_BIAS = 0,            /* y                                             */
  SF_EPI_BIAS_GELU = 1,       /* gelu_tanh(y)                                  */
  SF_EPI_BIAS_RESID = 2,      /* resid + y                                     */
  SF_EPI_BIAS_GATE_RESID = 3, /* resid + y * (gate_mod[n] + gate_e0[group(m)][n]) */
  SF_EPI_F32 = 4              /* raw fp32 accumulators, no bias: `out` is float*, ldo in floats
                                 (attention logits of the VAE's single-head block) */
};

typedef struct sf_gemm_args {
  const void* a;       /* [M, K], row stride lda                       */
  const void* w;       /* [N, K], row stride ldw (nn.Linear.weight)    */
  const void* bias;    /* [N] or NULL                                  */
  void* out;           /* [M, N], row stride ldo                       */
  const void* resid;   /* [M, N], row stride ldr; may alias out        */
  const void* gate_mod; /* [N]   (block.modulation[:, 2 or 5])         */
  const void* gate_e0;  /* [groups, *] first element of the gate chunk */
  int64_t gate_group_stride; /* elements between consecutive groups in gate_e0 */
  int32_t rows_per_group;    /* group(m) = m / rows_per_group                   */
  int32_t M, N, K;
  int32_t lda, ldw, ldo, ldr;
  int32_t epilogue;    /* enum sf_epilogue */
  int32_t batch;       /* > 1: `batch` independent products; entry b uses a + b*a_bstride, w + b*w_bstride,
                          out + b*o_bstride, resid + b*r_bstride (elements; bias / gates shared); 0 or 1: one product */
  int64_t a_bstride, w_bstride, o_bstride, r_bstride;
  int32_t structure;   /* enum sf_gemm_structure: 0 = picked from the shape; the others force a tiling (tests, A/B timing) */
} sf_gemm_args;

enum sf_gemm_structure { SF_GEMM_AUTO = 0, SF_GEMM_T128 = 1 /* 128 x 128 tile, 4 waves, 2 workgroups / CU */,
                         SF_GEMM_PP256 = 2 /* 256 x 256 tile, 8 waves, the two waves of a SIMD half a phase apart */,
                         SF_GEMM_PP128 = 3 /* 128 x 256 tile, same structure, two phases per k-tile, 3-deep rings */,
                         SF_GEMM_PP224 = 4, SF_GEMM_PP192 = 5 /* the 256 x 256 kernel with 7 / 6 row tiles per wave: 224- and
                                                                 192-row tiles for shapes whose 256-row tiles leave a round of
                                                                 workgroups mostly empty */ };

int sf_gemm_bf16(const sf_gemm_args* args, void* stream);

/* Small-M linear layer (M <= 32), weight-bandwidth bound: out = act_out(act_in(x) @ w^T + b).
 * Replaces the time-embedding MLPs, causal_model.py:464-467, :829-832.
 * act codes: 0 none, 1 SiLU, 2 GELU-tanh. */
int sf_small_linear(const void* x, const void* w, const void* bias, void* out, int M, int N, int K,
                    int act_in, int act_out, void* stream);

/* Sinusoidal timestep embedding in float64, wan/modules/model.py:15-25.
 * t: [n] float32 (t_is_int64 = 0) or int64 (= 1); out [n, dim] bf16 = cat(cos, sin). */
int sf_sinusoid_embedding(const void* t, int t_is_int64, void* out, int n, int dim, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm (no affine, eps) + AdaLN modulate: out = LN(x) * (1 + scale) + shift with
 * scale = mod_scale[c] + e0_scale[group(row)][c], shift likewise; group(row) = row / rows_per_group.
 * Replaces norm1/norm2/head.norm + the broadcast mul/add of causal_model.py:315, :327-328, :366. */
int sf_layernorm_modulate(const void* x, void* out, int M, int C, float eps, const void* mod_shift,
                          const void* mod_scale, const void* e0_shift, const void* e0_scale,
                          int64_t e0_group_stride, int rows_per_group, void* stream);

/* LayerNorm with affine weight/bias (norm3, causal_model.py:268-270, :324). */
int sf_layernorm_affine(const void* x, const void* weight, const void* bias, void* out, int M, int C,
                        float eps, void* stream);

/* WanRMSNorm over the full channel dim (model.py:70-86): out = bf16(x * rsqrt(mean(x^2)+eps)) * w.
 * x has row stride ldx, out row stride ldo (in place allowed). */
int sf_rmsnorm(const void* x, int ldx, const void* weight, void* out, int ldo, int M, int C, float eps,
               void* stream);

/* Fused q/k RMSNorm + 3-axis RoPE + KV-cache write for the fused qkv projection output
 * (causal_model.py:112-114, :195-200, :221-229).
 *   qkv   [B*L, 3C]  (q | k | v per row), L = f*h*w tokens per sample in (f,h,w) order
 *   q_out [B*L, C]   roped, normalised queries
 *   k_cache/v_cache [B, cache_tokens, H, D]: rows [write_start, write_start+L) are overwritten
 *   rope_cos/sin: float32 [1024, D/2] tables in the reference's `freqs` column layout
 *   (time | height | width), time index offset by start_frame. */
int sf_qkv_norm_rope_cache(const void* qkv, const void* norm_q_w, const void* norm_k_w, void* q_out,
                           void* k_cache, void* v_cache, const float* rope_cos, const float* rope_sin,
                           int B, int f, int h, int w, int C, int num_heads, int64_t cache_tokens,
                           int write_start, int start_frame, float eps, void* stream);

/* Rolling-window eviction, causal_model.py:212-217: for every sample move
 * cache[sink+evict : sink+evict+keep] -> cache[sink : sink+keep] (overlap-safe). */
int sf_kv_evict(void* cache, int B, int64_t cache_tokens, int row_elems, int sink, int evict, int keep,
                void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Non-causal attention softmax(q k^T / sqrt(D)) v, D = 128.  Replaces flash_attn_varlen_func /
 * SDPA of wan/modules/attention.py:136-150, :198 for self-attention over the KV cache
 * (causal_model.py:230-234) and for T5 cross-attention (model.py:189).
 *   q   [B, Lq, H, D]  with token stride q_stride (elements) and batch stride q_bstride
 *   k,v [B, Lk, H, D]  with token stride kv_stride and batch stride kv_bstride
 *   out [B, Lq, H, D]  token stride o_stride, batch stride o_bstride */
int sf_attention(const void* q, const void* k, const void* v, void* out, int B, int H, int Lq, int Lk,
                 int64_t q_stride, int64_t q_bstride, int64_t kv_stride, int64_t kv_bstride,
                 int64_t o_stride, int64_t o_bstride, void* stream);

/* The same with the kernel structure named by the caller instead of picked from the shape (tests and A/B timing;
 * every structure computes every shape): SF_ATTN_R64 = 64 query rows per wave, one wave per SIMD, hand-scheduled
 * (256 rows per workgroup); SF_ATTN_W8 = 8-wave anti-phase workgroups of 256 rows; SF_ATTN_W4 = 4-wave workgroups of
 * 128 rows.  sf_attention == sf_attention_ex(..., SF_ATTN_AUTO, stream). */
enum sf_attn_structure { SF_ATTN_AUTO = 0, SF_ATTN_R64 = 1, SF_ATTN_W8 = 2, SF_ATTN_W4 = 3 };
int sf_attention_ex(const void* q, const void* k, const void* v, void* out, int B, int H, int Lq, int Lk,
                    int64_t q_stride, int64_t q_bstride, int64_t kv_stride, int64_t kv_bstride,
                    int64_t o_stride, int64_t o_bstride, int structure, void* stream);

/* ------------------------------------------------------------------------------------------
 * Patch gather for the (1,2,2) Conv3d patch embedding, causal_model.py:458-459, :775-781:
 * x [B, F, Cin, H, W] (the wrapper's layout) -> cols [B*F*(H/2)*(W/2), Cin*4] with column index
 * c*4 + p*2 + q, so that cols @ patch_embedding.weight.flatten(1)^T is the convolution. */
int sf_patchify(const void* x, void* cols, int B, int F, int Cin, int H, int W, void* stream);

/* Unpatchify (causal_model.py:1081-1104) fused with flow -> x0 (wan_wrapper.py:204-228):
 *   head_out [B*F*h*w, 4*Cout] (column = (p*2+q)*Cout + c) -> flow [B, F, Cout, H, W],
 *   x0 = xt - sigma(t) * flow evaluated in float64; sigma by nearest-timestep lookup in the
 *   n_table-entry float32 tables.  timestep has B*groups entries (frames_per_group = F/groups). */
int sf_unpatchify_x0(const void* head_out, const void* xt, const void* timestep, int t_is_int64,
                     const float* sigmas, const float* timesteps, int n_table, void* flow, void* x0,
                     int B, int F, int groups, int Cout, int H, int W, void* stream);

/* FlowMatchScheduler.add_noise (utils/scheduler.py:159-176): out = (1-sigma) x0 + sigma eps in
 * fp32; one timestep per leading index (n_outer), inner = C*H*W elements each. */
int sf_add_noise(const void* x0, const void* eps, const void* timestep, int t_is_int64,
                 const float* sigmas, const float* timesteps, int n_table, void* out, int n_outer,
                 int64_t inner, void* stream);

/* out[i] = sum_{k < n_terms} coefs[k] * xs[k][i]   (bf16 tensors of n elements, fp32 accumulation in the order
 * k = 0, 1, ...; one rounding to bf16 at the end; 1 <= n_terms <= SF_LINCOMB_MAX; `out` may alias any input).
 * `xs` and `coefs` are HOST arrays (device pointers / scalars), read before the call returns.
 * Replaces the tensor arithmetic of the 50-step sampler: the classifier-free-guidance blend
 * (pipeline/causal_diffusion_inference.py:423-424) and FlowUniPCMultistepScheduler's convert_model_output /
 * multistep_uni_p_bh_update / multistep_uni_c_bh_update (wan/utils/fm_solvers_unipc.py:279-347, :350-484, :486-626),
 * all of which are linear in their tensors with scalar coefficients the host evaluates. */
#define SF_LINCOMB_MAX 6
int sf_lincomb_bf16(void* out, const void* const* xs, const float* coefs, int n_terms, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * One whole denoiser pass: CausalWanModel._forward_inference + flow->x0
 * (causal_model.py:725-893, wan_wrapper.py:288-300, :340-344) as a single host call that
 * enqueues every kernel on `stream`. */
typedef struct sf_layer_weights {
  const void* modulation;                 /* [6, C] */
  const void *norm3_w, *norm3_b;          /* [C]    */
  const void *qkv_w, *qkv_b;              /* [3C, C], [3C]: self_attn q|k|v stacked */
  const void *norm_q_w, *norm_k_w;        /* [C]    */
  const void *o_w, *o_b;                  /* [C, C] */
  const void *cq_w, *cq_b;                /* cross_attn.q */
  const void *ckv_w, *ckv_b;              /* [2C, C]: cross_attn k|v stacked */
  const void *cnorm_q_w, *cnorm_k_w;
  const void *co_w, *co_b;
  const void *ffn0_w, *ffn0_b;            /* [ffn, C] */
  const void *ffn2_w, *ffn2_b;            /* [C, ffn] */
} sf_layer_weights;

typedef struct sf_model {
  int32_t dim, ffn_dim, num_heads, num_layers, in_dim, out_dim, freq_dim, text_dim, text_len;
  float eps;
  const void *patch_w, *patch_b;          /* [C, in_dim*4] */
  const void *text0_w, *text0_b, *text2_w, *text2_b;
  const void *time0_w, *time0_b, *time2_w, *time2_b;
  const void *tproj_w, *tproj_b;          /* [6C, C] */
  const void *head_w, *head_b;            /* [4*out_dim, C] */
  const void* head_mod;                   /* [2, C] */
  const void *pose_w, *pose_b;            /* optional pose_proj Linear(pose_dim, C) (causal_model.py:493-503); NULL if absent */
  int32_t pose_dim;                       /* width of add_condition; with pose_w == NULL and pose_dim == dim the projection is
                                             the reference's nn.Identity() of dim-5120 models (:500-501): x += add_condition */
  const sf_layer_weights* layers_host;    /* HOST array [num_layers] */
  const float *rope_cos, *rope_sin;       /* float32 [1024, 64] */
  const float *sched_sigmas, *sched_timesteps; /* float32 [n_table] */
  int32_t n_table;
} sf_model;

typedef struct sf_forward_args {
  int32_t batch, frames, lat_h, lat_w;    /* noisy: [B, F, in_dim, H, W] */
  int32_t groups;                         /* timestep.shape[1] (modulation groups)       */
  const void* noisy;
  const void* timestep;                   /* [B, groups] */
  int32_t t_is_int64;
  const void* prompt_embeds;              /* [B, text_len, text_dim], zero padded; read only if init_cross */
  int32_t init_cross;                     /* 1: (re)compute text embedding + cross-attn K/V caches */
  const void* add_condition;              /* optional pose tokens [B, F*h*w, pose_dim]: x += pose_proj(add_condition)
                                             after the patch embedding (causal_model.py:786-819); NULL if none */
  void* const* k_cache_host;              /* HOST arrays [num_layers] of device pointers */
  void* const* v_cache_host;              /*   each [B, cache_tokens, H, D]              */
  void* const* ck_cache_host;             /*   each [B, text_len, H, D]                  */
  void* const* cv_cache_host;
  int64_t cache_tokens;
  /* cache plan (host integers; see kv_cache_plan in self-forcing_amd/kvcache.py) */
  int32_t sink_tokens, evict, keep;       /* evict > 0: roll the window first */
  int32_t write_start;                    /* rows [write_start, write_start + F*h*w) get the new K/V */
  int32_t attn_start, attn_end;           /* attend over cache rows [attn_start, attn_end) */
  int32_t start_frame;                    /* RoPE time offset = current_start // (h*w) */
  void* evict_scratch;                    /* >= batch * keep * dim * 2 bytes when evict > 0 */
  size_t evict_scratch_bytes;
  int32_t cache_only;                     /* 1: only the KV-cache update is wanted (the context pass /
                                             warm-up passes, whose outputs the pipeline discards,
                                             causal_inference.py:143-168, :227-235): everything after
                                             the LAST layer's K/V write is skipped, outputs untouched */
  void* flow_out;                         /* [B, F, out_dim, H, W] */
  void* x0_out;                           /* [B, F, out_dim, H, W] */
  void* workspace;
  size_t workspace_bytes;
  void* kv_index_out;                     /* optional int64 [num_layers][2]: every row is set to (global_end, attn_end) at
                                             the end of the pass -- the cache dicts' "global_end_index" / "local_end_index"
                                             tensors (causal_model.py:235-236) when they are views of one buffer; NULL:
                                             the caller updates its index tensors itself */
  int64_t global_end;                     /* current_start + F*h*w (only written to kv_index_out) */
} sf_forward_args;

size_t sf_dit_workspace_bytes(const sf_model* model, int batch, int frames, int lat_h, int lat_w,
                              int groups);
int sf_dit_forward(const sf_model* model, const sf_forward_args* args, void* stream);
/* Two passes of the rollout in ONE call: the context pass of chunk k (`context_pass`, cache_only = 1: it rewrites the
 * chunk's K / V "clean", causal_inference.py:226-235) and the first denoising pass of chunk k + 1 (`next_pass`,
 * :190-205), which the reference runs back to back.  Layer l of the second needs only layer l's K / V of the first, so
 * the two run layer by layer as one batch of 2 x batch samples through every row-wise kernel and GEMM (twice the rows
 * per GEMM) and one after the other through the cache (eviction, K / V write, attention).  Results are bit-identical to
 * two sf_dit_forward calls.  Both argument structs must name the same caches, latent geometry, batch and groups, and the
 * same workspace, sized sf_dit_workspace_bytes(model, 2 * batch, ...); neither may have init_cross set;
 * `next_pass->kv_index_out` (if any) receives the final indices. */
int sf_dit_forward_pair(const sf_model* model, const sf_forward_args* context_pass, const sf_forward_args* next_pass,
                        void* stream);

/* ==========================================================================================
 * Wan VAE decode (latents -> pixels): WanVAEWrapper.decode_to_pixel -> WanVAE_.decode /
 * cached_decode -> Decoder3d.forward (utils/wan_wrapper.py:95-117, wan/modules/vae.py:556-593,
 * :423-472).  Activations are CHANNELS-LAST bf16 volumes [T][H][W][C]; the reference's
 * `feat_cache` list (two cached input frames per causal convolution, vae.py:206-216) becomes two
 * history frames kept physically in front of the new frames of each convolution's input buffer.
 * ========================================================================================== */

/* Implicit-GEMM convolution, fp32 accumulation on the matrix cores:
 *   out[(t,h,w)][n] = bias[n] + sum_{dt,dh,dw,ci} x[t+dt+t_in_offset][(h+dh-kh/2)>>up][(w+dw-kw/2)>>up][ci]
 *                                               * w[n][((dt*kh+dh)*kw+dw)*Cin + ci]
 * with zero padding in h/w.  Replaces CausalConv3d (vae.py:17-38; kernel 3x3x3, (3,1,1) or 1x1x1),
 * Resample's nearest-2x Upsample + Conv2d 3x3 (vae.py:77-83, :139-141; upsample = 1, kt = 1),
 * the residual add of ResidualBlock (vae.py:221), the channel->frame interleave after the time
 * convolution (vae.py:134-137) and the .float().clamp_(-1, 1) of decode_to_pixel (wan_wrapper.py:113). */
enum sf_conv_epilogue {
  SF_CONV_BIAS = 0,            /* bf16 out[row][n] = y                                            */
  SF_CONV_BIAS_RESID = 1,      /* bf16 out[row][n] = y + resid[row][n]; resid may alias out       */
  SF_CONV_BIAS_CLAMP_F32 = 2   /* float out_f32[t][n][h][w] = clamp(y, -1, 1) (planar, Cout small) */
};

typedef struct sf_conv_args {
  const void* x;          /* [Tin][Hin][Win][Cin], Cin % 32 == 0                                   */
  const void* w;          /* [Cout][ldw]: k = tap*Cin + ci, zero padded to ldw >= roundup(taps*Cin, 64) */
  const void* bias;       /* [Cout]                                                                */
  void* out;              /* rows of ldo channels; row = (out_frame_offset + t')*H*W + h*W + w     */
  const void* resid;      /* same row indexing, ldr channels per row                               */
  float* out_f32;         /* SF_CONV_BIAS_CLAMP_F32 only: [Tout][Cout][H][W]                       */
  int32_t Tout, H, W;     /* output volume                                                         */
  int32_t Hin, Win;       /* input frame size: (H, W), or (H/2, W/2) when upsample = 1             */
  int32_t Cin, Cout;
  int32_t kt, kh, kw;     /* 3 or 1 each; kh == kw                                                 */
  int32_t upsample;       /* 1: read the input through a nearest-neighbour 2x upsampling           */
  int32_t t_in_offset;    /* input frame of tap dt for output frame t is t + dt + t_in_offset      */
  int32_t ldw, ldo, ldr;
  int32_t out_frame_offset;
  int32_t interleave_c;   /* > 0 (= Cout/2): channel n of output frame t goes to frame 2t + n/interleave_c,
                             channel n % interleave_c (t' above)                                   */
  int32_t epilogue;       /* enum sf_conv_epilogue */
  int32_t structure;      /* enum sf_conv_structure: 0 = picked from the shape; the others force a kernel (tests, A/B timing) */
  /* Optional second output of the SF_CONV_HALO kernel with Cout = 96 or 192 (all channels in one tile): the NEXT
   * convolution's input, SiLU(RMS_norm(y) * gamma) (vae.py:41-56, :190-196), written to
   * norm_out[((norm_frame_offset + t) * H + h) * W + w][norm_ld]; `out` may then be NULL (raw result not needed). */
  void* norm_out;
  const void* norm_gamma; /* [Cout] */
  int32_t norm_ld, norm_frame_offset;
} sf_conv_args;

enum sf_conv_structure { SF_CONV_AUTO = 0, SF_CONV_IGEMM = 1 /* A tile gathered per tap (every shape) */,
                         SF_CONV_HALO = 2 /* 16 x 16 output patch, input halo staged once per (channel slice, frame):
                                             3 x 3 spatial taps, Cout % 96 == 0, bf16 bias / bias + residual */ };

int sf_conv_igemm(const sf_conv_args* args, void* stream);
int sf_conv_pick_nt(int cout);   /* column tiles (of 16) per wave the launcher will use for Cout    */

/* RMS_norm of the VAE (vae.py:41-56): out = x / max(||x||_2, 1e-12) * sqrt(C) * gamma over the C
 * channels of each row, optionally followed by SiLU (the nn.SiLU after it in ResidualBlock / head,
 * vae.py:190-196, :420-421).  x, out: [rows][C] contiguous, C % 8 == 0, C <= 512; in place allowed. */
int sf_rmsnorm_silu_cl(const void* x, const void* gamma, void* out, int64_t rows, int C, int silu,
                       void* stream);

/* Row softmax of the single-head attention block (vae.py:252-257): p[r][c] = softmax_c(scale * s[r][c])
 * for c < cols, 0 for cols <= c < cols_padded.  s float32 row stride lds, p bf16 row stride ldp. */
int sf_softmax_rows(const float* s, int64_t lds, void* p, int64_t ldp, int rows, int cols,
                    int cols_padded, float scale, void* stream);

/* Latent frame -> input of decoder.conv1: un-scale (z * std + mean, vae.py:559-563), the 1x1x1
 * conv2 (vae.py:564) and the layout change [z][h][w] -> [h][w][c_pad] (channels z..c_pad-1 zero). */
int sf_vae_prepare_latent(const void* latent, const float* mean, const float* std, const void* conv2_w,
                          const void* conv2_b, void* out, int z, int h, int w, int c_pad, void* stream);

typedef struct sf_vae_conv {
  const void* w;          /* repacked [cout][ldw] as sf_conv_args.w; NULL = layer absent           */
  const void* bias;       /* [cout] */
  int32_t cin, cout, kt, kh, kw, ldw;   /* cin already padded to a multiple of 32                  */
} sf_vae_conv;

typedef struct sf_vae_resblock {          /* ResidualBlock, vae.py:182-221 */
  const void* gamma1;     /* residual.0.gamma [in_dim]  */
  const void* gamma2;     /* residual.3.gamma [out_dim] */
  sf_vae_conv conv1;      /* residual.2 */
  sf_vae_conv conv2;      /* residual.6 */
  sf_vae_conv shortcut;   /* 1x1x1, w = NULL when in_dim == out_dim */
} sf_vae_resblock;

#define SF_VAE_MAX_STAGES 4

typedef struct sf_vae_model {             /* Decoder3d + conv2, vae.py:369-421, :503 */
  int32_t z_dim;
  int32_t n_stages;                       /* len(dim_mult) = 4                                      */
  int32_t res_per_stage;                  /* num_res_blocks + 1 = 3                                 */
  int32_t temporal_up[SF_VAE_MAX_STAGES]; /* stage i ends with upsample3d (1) / upsample2d (0); last stage: none */
  const float* latent_mean;               /* float32 [z_dim] */
  const float* latent_std;                /* float32 [z_dim] */
  const void *conv2_w, *conv2_b;          /* [z][z], [z] */
  sf_vae_conv conv1;                      /* decoder.conv1, cin padded to 32 */
  sf_vae_resblock mid0, mid2;             /* decoder.middle.0 / .2 */
  const void* attn_gamma;                 /* middle.1.norm.gamma [C]            */
  const void *attn_qk_w, *attn_qk_b;      /* to_qkv rows [0, 2C): [2C][C], [2C] */
  const void *attn_v_w, *attn_v_b;        /* to_qkv rows [2C, 3C)               */
  const void *attn_proj_w, *attn_proj_b;  /* [C][C], [C] */
  const sf_vae_resblock* res_host;        /* HOST array [n_stages * res_per_stage] */
  sf_vae_conv time_conv[SF_VAE_MAX_STAGES];   /* per stage; w = NULL where absent  */
  sf_vae_conv up_conv[SF_VAE_MAX_STAGES];     /* Resample.resample[1] (Conv2d 3x3) */
  const void* head_gamma;                 /* decoder.head.0.gamma */
  sf_vae_conv head_conv;                  /* decoder.head.2 (cout = 3) */
} sf_vae_model;

/* Per-stream persistent state = the input volumes of every cached convolution: 2 history frames + new frames, as a
 * window that slides through 2 + window_frames * T frames (T = the stage's frames per latent frame); scratch =
 * everything else, reusable by any stream that does not overlap in time.  `window_frames` (2..64) is chosen by the
 * caller once per state: a call decodes up to window_frames - 1 latent frames. */
size_t sf_vae_state_bytes(const sf_vae_model* model, int lat_h, int lat_w, int window_frames);
size_t sf_vae_scratch_bytes(const sf_vae_model* model, int lat_h, int lat_w, int window_frames);
/* WanVAE_.clear_cache (vae.py:610-617): zero every history. */
int sf_vae_reset(const sf_vae_model* model, void* state, size_t state_bytes, int lat_h, int lat_w,
                 int window_frames, void* stream);
/* n_frames iterations of the per-latent-frame loop of decode / cached_decode (vae.py:566-578) in one call:
 * latent_frames [n_frames][z][lat_h][lat_w] bf16 -> pixels [T][3][8 lat_h][8 lat_w] float32 in [-1, 1], T = 1 for
 * the frame that follows a reset (frame_index 0: decoded alone, vae.py:109-111, :134-137), else 4 n_frames.  The
 * result is bit-identical to n_frames single-frame calls (the causal convolutions see the same inputs either way); a
 * group fills the chip at the low-resolution stages.
 * The library keeps no state of its own, so the caller says where the history windows are: `frame_index` = latent
 * frames decoded into `state` since its reset; `window` = the slot (in latent frames) this call's frames take in the
 * current lap of the sliding windows, window + n_frames <= window_frames; `history_at` = where the previous call
 * ended: equal to `window` while the windows slide on, or -- when the caller restarts at window 0 because the frames
 * no longer fit -- the previous lap's end slot, from which the two history frames of every volume are first copied
 * to the front.  A caller's bookkeeping (self-forcing_amd/vae.py): slot = 0 after a reset; per call: if slot +
 * n_frames > window_frames: (window, history_at) = (0, slot) else (slot, slot); slot = window + n_frames. */
int sf_vae_decode_frames(const sf_vae_model* model, void* state, size_t state_bytes, void* scratch,
                         size_t scratch_bytes, const void* latent_frames, int lat_h, int lat_w, int window_frames,
                         int frame_index, int n_frames, int window, int history_at, float* pixels_out, void* stream);

/* ==========================================================================================
 * umT5 text encoder (prompt token ids -> prompt embeddings): WanTextEncoder.forward after its tokenizer
 * (utils/wan_wrapper.py:40-55) -> T5Encoder.forward (wan/modules/t5.py:299-312).  Runs once per prompt.
 * ========================================================================================== */

/* nn.Embedding lookup: out[t][:] = table[ids[t]][:] (ids int64; an id outside [0, vocab) is an error
 * reported by the caller -- the kernel clamps). */
int sf_embedding_gather(const int64_t* ids, const void* table, void* out, int n_tokens, int dim, int vocab,
                        void* stream);

/* Logits -> probabilities of T5Attention (t5.py:104-118) for all heads of one sample:
 *   p[h][i][j] = softmax_j( s[h][i][j] + emb[rel_bucket[j - i + L - 1]][h] + (key_mask[j] ? 0 : -inf) ),  j < L
 *   p[h][i][j] = 0 for L <= j < ld  (zero padding so that p can be the A operand of a K = ld GEMM)
 * s float32 [H][L][ld], p bf16 [H][L][ld], emb bf16 [num_buckets][H] (T5RelativeEmbedding.embedding.weight),
 * rel_bucket int32 [2L-1] (host-computed bucket of every relative position, t5.py:236-256), key_mask int64 [L]. */
int sf_t5_softmax_bias(const float* s, void* p, const void* emb, const int32_t* rel_bucket, const int64_t* key_mask,
                       int H, int L, int ld, void* stream);

/* out[i] = a[i] * b[i] (bf16; the gated-GELU product fc1(x) * gelu(gate(x)), t5.py:138); in place allowed. */
int sf_mul_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);

/* rows of x [B*L][dim] whose mask entry is 0 are set to zero (WanTextEncoder.forward, wan_wrapper.py:50-51). */
int sf_zero_masked_rows(void* x, const int64_t* mask, int rows, int dim, void* stream);

typedef struct sf_t5_layer {       /* T5SelfAttention, t5.py:146-178 */
  const void* norm1_w;             /* [dim] */
  const void* qk_w;                /* attn.q | attn.k stacked: [2*dim_attn][dim] */
  const void* v_w;                 /* [dim_attn][dim] */
  const void* o_w;                 /* [dim][dim_attn] */
  const void* norm2_w;
  const void* gate_w;              /* ffn.gate.0: [dim_ffn][dim] */
  const void* fc1_w;               /* [dim_ffn][dim] */
  const void* fc2_w;               /* [dim][dim_ffn] */
  const void* pos_emb;             /* pos_embedding.embedding.weight [num_buckets][num_heads] */
} sf_t5_layer;

typedef struct sf_t5_model {
  int32_t vocab, dim, dim_attn, dim_ffn, num_heads, num_layers, num_buckets;
  float eps;
  const void* token_embedding;     /* [vocab][dim] */
  const sf_t5_layer* layers_host;  /* HOST array [num_layers] */
  const void* final_norm_w;        /* [dim] */
} sf_t5_model;

size_t sf_t5_workspace_bytes(const sf_t5_model* model, int batch, int seq_len);
/* ids, mask: int64 [batch][seq_len] (mask 1 = token, a prefix mask); rel_bucket int32 [2 seq_len - 1];
 * out bf16 [batch][seq_len][dim], rows past each prompt's length zero. */
int sf_t5_encode(const sf_t5_model* model, const int64_t* ids, const int64_t* mask, const int32_t* rel_bucket,
                 int batch, int seq_len, void* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measured ceilings of the box ("Peaks to divide by": "re-measure on the box (a peak-MFMA micro-kernel
 * and a streaming-copy kernel) and use the measured ceilings in the fraction").  Measurement entry points for
 * bench.py (roofline.measured_peak, hbm_measured_peak); no product kernel depends on them and the reference has no
 * counterpart.
 *   sf_probe_mfma: `workgroups` x 4 waves each issue iters x 32 v_mfma_f32_32x32x16_bf16 (shape 0) or iters x 64
 *                  v_mfma_f32_16x16x32_bf16 (shape 1) from registers, 4 independent accumulators; `operands` = 8 KiB
 *                  of (random) bf16, `sink` = workgroups * 256 floats; *flops_out (HOST, optional) = flops of the launch.
 *   sf_probe_copy: streaming copy of `bytes` (multiple of 16) from src to dst, 16 bytes per lane. */
int sf_probe_mfma(int shape, int iters, int workgroups, const void* operands, float* sink, double* flops_out_host,
                  void* stream);
int sf_probe_copy(const void* src, void* dst, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SF_HIP_H */
