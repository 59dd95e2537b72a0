// 3x3 (x kt) convolution on channels-last bf16 volumes with a HALO TILE in LDS, for gfx950 (MI355X).
//
// Same operator as conv_igemm.hip (CausalConv3d 3x3x3 / per-frame Conv2d 3x3 of the Wan VAE decoder incl. the fused
// nearest 2x upsampling, bias, residual add: wan/modules/vae.py:17-38, :77-83, :139-141, :221), for the full-resolution
// stages where it is the decode's biggest cost.  conv_igemm stages a fresh 128 x 64 A tile for EVERY tap: 27 (9) times
// the same activations travel global -> LDS, and that path moves 64 B/clk/CU -- at Cout = 96 the kernel needs 73 B/clk
// of it and runs at ~650 TFLOP/s.  Here the output tile is a 16 x 16 PATCH of one frame; per 32-channel slice and input
// frame the 18 x 18 input positions around it (10 x 10 with the fused upsampling) are staged ONCE ("plane": 324 x 64 B)
// and the nine spatial taps are nine SHIFTED fragment reads of that plane; only the 32-channel weight slice of each
// tap (6-12 KiB) is streamed per tap.  LDS-DMA traffic per MFMA drops 3x.
//
// Structure: the ping-pong scheme of gemm_bf16.hip -- 8 waves as 4 (patch rows) x 2 (channel halves), 64 positions x
// 16 NT channels per wave, the two waves of a SIMD one barrier interval apart: one issues a cluster of MFMAs from
// registers (TPC taps x 4 x NT), the other reads its next fragments, retires them and requests LDS-DMA pieces; raw
// s_barrier, counted vmcnt.  Planes live in a ring of 4 slots (plane P + 2 is requested during plane P), weight taps in
// a ring of 3 clusters (cluster q + 2 is requested during cluster q); a segment waits for everything requested BEFORE
// it, so a cluster's operands have a whole cluster (two intervals) to land.
//
// LDS images: position (or weight row) r holds its 32 channels as four 16-byte chunks; chunk c sits at slot
// c ^ (((r >> 2) & 1) << 1).  With that swizzle a ds_read_b128 fragment read of 16 CONSECUTIVE positions is free of bank
// conflicts for every alignment of the first position (checked exhaustively against the lane groups of
// MI355X_MICROARCH.md): the taps' shifts cost nothing.  The swizzle is applied on the LDS-DMA's source address.
#include "sf_common.h"
#include "../../include/sf_hip.h"

namespace {

constexpr int HALO_THREADS = 512;
constexpr int PLANE_SLOT = 21 * 1024;      // 336 positions x 64 B >= 18 x 18
constexpr int PLANE_RING = 4;
constexpr unsigned OOR = 0xFFFFFFF0u;      // byte offset past every volume: the range-checked LDS-DMA writes zeros

struct HaloP {
  const bf16_t* x;
  const bf16_t* w;
  const bf16_t* bias;
  bf16_t* out;
  const bf16_t* resid;
  float* out_f32;            // SF_CONV_BIAS_CLAMP_F32 (the 3-channel head): planar [T][Cout][H][W]
  bf16_t* norm_out;          // optional second output: SiLU(RMS_norm(y)) for the next convolution (tiles_n == 1 only)
  const bf16_t* norm_gamma;
  int norm_ld, norm_frame0;
  int H, W, Hin, Win, up, Cin, Cout, cpt, kt, t_off, ldw, ldo, ldr, out_frame0;
  int patches_h, patches_w, tiles_n;
  unsigned x_bytes;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int swz16(int r, int c) { return (c ^ (((r >> 2) & 1) << 1)) << 4; }

template <int NT, int TPC, int EPI>
__global__ __launch_bounds__(HALO_THREADS) void conv_halo_kernel(HaloP p) {
  constexpr int BN = 32 * NT;
  constexpr int W_TAP = BN * 64;                        // bytes of one tap's 32-channel weight slice
  constexpr int W_CLUSTER = TPC * W_TAP;
  constexpr int W_BASE = PLANE_RING * PLANE_SLOT;
  constexpr int DUMMY = W_BASE + 3 * W_CLUSTER;         // 1 KiB per wave: destination of padding requests
  constexpr int NWP = TPC * (BN / 16);                  // weight pieces (16 rows x 64 B) per cluster
  constexpr int WPW = (NWP + 7) / 8;                    // ... per wave (padded)
  constexpr int CPP = 9 / TPC;                          // clusters per plane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;

  // tile: XCD-aware bijective remap, then patch-major order (neighbouring patches share halo rows in an XCD's L2)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = wg / p.tiles_n, tn = wg - tile_m * p.tiles_n;
  const int per_frame = p.patches_h * p.patches_w;
  const int t = tile_m / per_frame, pr = tile_m - t * per_frame;
  const int py = pr / p.patches_w, px = pr - py * p.patches_w;
  const int h0 = py * 16, w0 = px * 16, n0 = tn * BN;
  const int PW = p.up ? 10 : 18, NPOS = PW * PW, NP = (NPOS + 15) >> 4;
  const int org_h = p.up ? (h0 >> 1) - 1 : h0 - 1, org_w = p.up ? (w0 >> 1) - 1 : w0 - 1;

  // ---- LDS-DMA sources.  A plane piece = 16 positions x 64 B; wave w requests pieces w, w + 8, w + 16.
  unsigned pl_off[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = wave + 8 * i;
    const int pos = j * 16 + (lane >> 2);
    const int hr = pos / PW, wc = pos - hr * PW;
    const int hin = org_h + hr, win = org_w + wc;
    const bool valid = j < NP && pos < NPOS && (unsigned)hin < (unsigned)p.Hin && (unsigned)win < (unsigned)p.Win;
    const int kq = (lane & 3) ^ (((pos >> 2) & 1) << 1);
    pl_off[i] = valid ? (unsigned)((((long)hin * p.Win + win) * p.Cin) * 2 + kq * 16) : OOR;
  }
  u32x4 x_srd;
  {
    const unsigned long long a64 = (unsigned long long)p.x;
    x_srd[0] = __builtin_amdgcn_readfirstlane((unsigned)a64);
    x_srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(a64 >> 32) & 0xFFFFu);
    x_srd[2] = __builtin_amdgcn_readfirstlane(p.x_bytes);
    x_srd[3] = 0x00020000u;
  }
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  auto dma_x = [&](unsigned voff, unsigned lds_addr) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(x_srd) : "memory");
  };
  const unsigned frameB = (unsigned)(p.Hin * p.Win * p.Cin * 2);
  auto request_plane = [&](int P) __attribute__((always_inline)) {     // plane P = (channel slice cs, temporal tap dt)
    const int cs = P / p.kt, dt = P - cs * p.kt;
    const unsigned add = (unsigned)(t + dt + p.t_off) * frameB + (unsigned)cs * 64u;
    const unsigned slot = lds_base + (unsigned)((P & (PLANE_RING - 1)) * PLANE_SLOT);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int j = wave + 8 * i;
      const unsigned dst = __builtin_amdgcn_readfirstlane(j < NP ? slot + (unsigned)j * 1024u : lds_base + (unsigned)(DUMMY + wave * 1024));
      dma_x(pl_off[i] == OOR ? OOR : pl_off[i] + add, dst);
    }
  };
  // A weight piece = 16 rows (output channels) x 64 B of one tap; wave w requests pieces w + 8 i of the cluster
  const bf16_t* w_src[WPW];
  int w_dst[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int idx = wave + 8 * i;
    const bool real = idx < NWP;
    const int tapi = real ? idx / (BN / 16) : 0, rb = real ? idx - tapi * (BN / 16) : 0;
    const int n = rb * 16 + (lane >> 2);
    const int kq = (lane & 3) ^ (((n >> 2) & 1) << 1);
    w_src[i] = p.w + (long)min(n0 + n, p.Cout - 1) * p.ldw + tapi * p.Cin + kq * 8;
    w_dst[i] = real ? tapi * W_TAP + rb * 1024 : -1;
  }
  auto request_weights = [&](int q, int tap0, int cs) __attribute__((always_inline)) {   // cluster q starts at tap tap0 of slice cs
    const long koff = (long)tap0 * p.Cin + cs * 32;
    char* ring = smem + W_BASE + (q % 3) * W_CLUSTER;
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      char* dst = w_dst[i] >= 0 ? ring + w_dst[i] : smem + DUMMY + wave * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + koff), (lptr_t)dst, 16, 0, 0);
    }
  };

  // ---- fragment read addresses.  Everything that depends on the tap is tabulated ONCE per lane (9 taps x 4 row tiles =
  // 36 registers): computed per cluster the shifts, the upsampling's halving and the swizzle were ~200 VALU / SALU
  // instructions in front of every cluster's reads, three times the cluster's own MFMA time.
  const int i16 = lane & 15, kq = lane >> 4;
  int w_frag[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = wn * 16 * NT + nt * 16 + i16;
    w_frag[nt] = W_BASE + n * 64 + swz16(n, kq);
  }
  int a_tab[9][4];
#pragma unroll
  for (int st = 0; st < 9; ++st) {
    const int dh = st / 3, dw = st - 3 * dh;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int r = wm * 4 + mt;
      const int hr = p.up ? ((r + dh - 1) >> 1) + 1 : r + dh;
      const int wc = p.up ? ((i16 + dw - 1) >> 1) + 1 : i16 + dw;
      const int pos = hr * PW + wc;
      a_tab[st][mt] = pos * 64 + swz16(pos, kq);
    }
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[TPC][4], wf[TPC][NT];

#ifdef SF_STAMP   // diagnostic build (tools/probes/conv_stamp.py): where a patch's time goes; the shipped library executes no stamp
  unsigned long long* stamp = (EPI != SF_CONV_BIAS_CLAMP_F32 && p.out_f32) ? reinterpret_cast<unsigned long long*>(p.out_f32) + ((long)blockIdx.x * 2 + grp) * 8 : nullptr;
  if (stamp && (tid & 255) == 0) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[4] = __builtin_amdgcn_s_memrealtime(); }
#endif
  const int NPL = p.cpt * p.kt;              // planes: (channel slice, temporal tap), slice-major
  // prologue: planes 0, 1 and the weights of clusters 0, 1 (both in plane 0: a plane has CPP >= 3 clusters)
  request_plane(0);
  if (NPL > 1) request_plane(1);
  request_weights(0, 0, 0);
  request_weights(1, TPC, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef SF_STAMP
  if (stamp && (tid & 255) == 0) stamp[1] = __builtin_amdgcn_s_memtime();
#endif
  if (grp == 1) __builtin_amdgcn_s_barrier();

  // weight k-offset (elements) of a plane's first tap: (dt * 9) * Cin + cs * 32; cursors of this plane and the next
  int cs_n = 0, dt_n = 1;                    // (cs, dt) of plane P + 1
  if (dt_n == p.kt) { dt_n = 0; cs_n = 1; }
  long k_cur = 0, k_nxt = (long)dt_n * 9 * p.Cin + cs_n * 32;
  for (int P = 0; P < NPL; ++P) {
    const char* plane = smem + (P & (PLANE_RING - 1)) * PLANE_SLOT;
    const bool more_plane = P + 2 < NPL, next_plane = P + 1 < NPL;
#pragma unroll
    for (int c = 0; c < CPP; ++c) {          // cluster q = P * CPP + c; CPP % 3 == 0, so its weight ring slot is c % 3
      // ---------------- load segment
#pragma unroll
      for (int ti = 0; ti < TPC; ++ti) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[ti][mt] = *reinterpret_cast<const bf16x8*>(plane + a_tab[c * TPC + ti][mt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          wf[ti][nt] = *reinterpret_cast<const bf16x8*>(smem + (c % 3) * W_CLUSTER + ti * W_TAP + w_frag[nt]);
      }
      const bool req_plane = c == 0 && more_plane;
      if (req_plane) request_plane(P + 2);
      // weights of cluster q + 2: two clusters further in this plane, or in the next one
      const bool req_w = c + 2 < CPP || next_plane;
      if (req_w) {
        const long koff = c + 2 < CPP ? k_cur + (long)(c + 2) * TPC * p.Cin : k_nxt + (long)(c + 2 - CPP) * TPC * p.Cin;
        char* ring = smem + W_BASE + ((c + 2) % 3) * W_CLUSTER;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
          char* dst = w_dst[i] >= 0 ? ring + w_dst[i] : smem + DUMMY + wave * 1024;
          __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + koff), (lptr_t)dst, 16, 0, 0);
        }
      }
      // everything requested BEFORE this segment has landed: the next cluster's weights, the next plane
      if (req_plane) {
        if (req_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + 3) : "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      } else {
        if (req_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): this wave's fragment reads are retired
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- MFMA cluster
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ti = 0; ti < TPC; ++ti)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ti][nt], af[ti][mt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    // advance the plane cursors
    k_cur = k_nxt;
    if (++dt_n == p.kt) { dt_n = 0; ++cs_n; }
    k_nxt = (long)dt_n * 9 * p.Cin + cs_n * 32;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
#ifdef SF_STAMP
  if (stamp && (tid & 255) == 0) stamp[2] = __builtin_amdgcn_s_memtime();
#endif

  if (EPI == SF_CONV_BIAS_CLAMP_F32) {
    // the decoder's head: Cout = 3 of the 32 columns are real; float, clamp(-1, 1), planar output (the layout
    // decode_to_pixel returns: utils/wan_wrapper.py:113).  Lanes 0-15 hold channels 0-3 of their position.
    if ((lane >> 4) == 0 && wn == 0) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int h = h0 + wm * 4 + mt, w = w0 + i16;
        if (h < p.H && w < p.W) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < p.Cout) {
              const float v = acc[mt][0][j] + (float)p.bias[j];
              p.out_f32[(((long)t * p.Cout + j) * p.H + h) * p.W + w] = fminf(fmaxf(v, -1.f), 1.f);
            }
        }
      }
    }
    return;
  }
  // ---- epilogue: bias (+ residual), through LDS so that every store covers whole 16-byte channel groups of a position
  constexpr int RBP = 64 * NT + 16;          // padded row bytes of the 256 x BN staging image
  char* obuf = smem;
  const long frame_row = (long)(p.out_frame0 + t) * p.H;
  {
    int ncol[NT];
    bf16x4 bias_v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      ncol[nt] = min(n0 + wn * 16 * NT + nt * 16 + (lane >> 4) * 4, p.Cout - 4);
      bias_v[nt] = *reinterpret_cast<const bf16x4*>(p.bias + ncol[nt]);
    }
    bf16x4 rv[4][NT];
    if (EPI == SF_CONV_BIAS_RESID) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int h = min(h0 + wm * 4 + mt, p.H - 1), w = min(w0 + i16, p.W - 1);
        const long row = (frame_row + h) * p.W + w;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rv[mt][nt] = *reinterpret_cast<const bf16x4*>(p.resid + row * p.ldr + ncol[nt]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = (wm * 4 + mt) * 16 + i16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = wn * 16 * NT + nt * 16 + (lane >> 4) * 4;
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = acc[mt][nt][j] + (float)bias_v[nt][j];
        if (EPI == SF_CONV_BIAS_RESID) {
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] += (float)rv[mt][nt][j];
        }
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)y[j];
        *reinterpret_cast<bf16x4*>(obuf + row * RBP + col * 2) = o;
      }
    }
  }
  __syncthreads();
  constexpr int CPR = 4 * NT;                // 16-byte chunks per position
  // Optional fused RMS_norm + SiLU (vae.py:41-56, :190-196) of the tile for the NEXT convolution's input volume: with all
  // output channels in this tile a position's row is complete in LDS.  Same arithmetic as sf_rmsnorm_silu_cl on the
  // bf16-rounded values: x * sqrt(C) / max(||x||, 1e-12) * gamma, SiLU, one rounding.
  float* rinv = reinterpret_cast<float*>(smem + 256 * RBP);
  if (p.norm_out) {
    const int row = tid >> 1, half = tid & 1;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CPR / 2; ++i) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(obuf + row * RBP + (half * (CPR / 2) + i) * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
    }
    ss += __shfl_xor(ss, 1, 64);
    if (half == 0) rinv[row] = sqrtf((float)BN) / fmaxf(sqrtf(ss), 1e-12f);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < (256 * CPR) / HALO_THREADS; ++i) {
    const int id = i * HALO_THREADS + tid;
    const int row = id / CPR, ch = id - row * CPR;
    const int h = h0 + (row >> 4), w = w0 + (row & 15), n = n0 + ch * 8;
    if (h < p.H && w < p.W && n < p.Cout) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(obuf + row * RBP + ch * 16);
      if (p.out) *reinterpret_cast<bf16x8*>(p.out + ((frame_row + h) * p.W + w) * p.ldo + n) = v;
      if (p.norm_out) {
        const float inv = rinv[row];
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(p.norm_gamma + n);
        bf16x8 o8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = (bf16_t)silu_fast_f((float)v[j] * inv * (float)g[j]);
        *reinterpret_cast<bf16x8*>(p.norm_out + (((long)(p.norm_frame0 + t) * p.H + h) * p.W + w) * p.norm_ld + n) = o8;
      }
    }
  }
#ifdef SF_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stamp && (tid & 255) == 0) { stamp[3] = __builtin_amdgcn_s_memtime(); stamp[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

template <int NT, int TPC>
int launch_halo(const HaloP& p, int tiles, int epi, hipStream_t s) {
  constexpr int LDS = PLANE_RING * PLANE_SLOT + 3 * TPC * 32 * NT * 64 + 8 * 1024;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(256 * (64 * NT + 16) + 1024 <= LDS, "epilogue staging + norm factors fit the k-loop's LDS");
  static_assert(NT == 3 || NT == 6, "bf16 epilogues: 96 or 192 columns per tile");
  static bool done = false;   // one-time registration of the kernels' LDS size (idempotent)
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<NT, TPC, SF_CONV_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<NT, TPC, SF_CONV_BIAS_RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    done = true;
  }
  if (epi == SF_CONV_BIAS)
    hipLaunchKernelGGL((conv_halo_kernel<NT, TPC, SF_CONV_BIAS>), dim3(tiles), dim3(HALO_THREADS), LDS, s, p);
  else
    hipLaunchKernelGGL((conv_halo_kernel<NT, TPC, SF_CONV_BIAS_RESID>), dim3(tiles), dim3(HALO_THREADS), LDS, s, p);
  return 0;
}

int launch_halo_head(const HaloP& p, int tiles, hipStream_t s) {   // NT = 1: 32 columns, Cout <= 4 of them real
  constexpr int LDS = PLANE_RING * PLANE_SLOT + 3 * 3 * 32 * 64 + 8 * 1024;
  static bool done = false;
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<1, 3, SF_CONV_BIAS_CLAMP_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    done = true;
  }
  hipLaunchKernelGGL((conv_halo_kernel<1, 3, SF_CONV_BIAS_CLAMP_F32>), dim3(tiles), dim3(HALO_THREADS), LDS, s, p);
  return 0;
}

}  // namespace

// Returns 1 when the problem is outside this kernel's domain (the caller then uses conv_igemm; with a fused norm
// output that is an error the caller reports), 0 after a launch, negative on error.  Domain: 3 x 3 spatial taps (kt 1 or 3), Cin % 32 == 0, Cout a multiple of 96 or of 192, bf16
// output with bias / bias + residual, no channel -> frame interleave.
__attribute__((visibility("hidden"))) int sf_conv_halo_launch(const sf_conv_args* a, void* stream) {
  if (!(a->kh == 3 && a->kw == 3) || a->interleave_c != 0 || a->Cin % 32 != 0) return 1;
  if (a->H < 16 || a->W < 16) return 1;
  const bool head = a->epilogue == SF_CONV_BIAS_CLAMP_F32 && a->Cout <= 4 && a->out_f32 && !a->norm_out;
  if (head) {
    HaloP p;
    p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w; p.bias = (const bf16_t*)a->bias;
    p.out = nullptr; p.resid = nullptr; p.out_f32 = a->out_f32; p.norm_out = nullptr; p.norm_gamma = nullptr; p.norm_ld = 0; p.norm_frame0 = 0;
    p.H = a->H; p.W = a->W; p.Hin = a->Hin; p.Win = a->Win; p.up = a->upsample;
    p.Cin = a->Cin; p.Cout = a->Cout; p.cpt = a->Cin / 32; p.kt = a->kt; p.t_off = a->t_in_offset;
    p.ldw = a->ldw; p.ldo = 0; p.ldr = 0; p.out_frame0 = 0;
    p.patches_h = (a->H + 15) / 16; p.patches_w = (a->W + 15) / 16; p.tiles_n = 1;
    const long xbh = (long)(a->t_in_offset + a->Tout + a->kt - 1) * a->Hin * a->Win * a->Cin * 2;
    if (xbh >= 0xFFFFFF00L) return 1;
    p.x_bytes = (unsigned)xbh;
    const long th = (long)a->Tout * p.patches_h * p.patches_w;
    if (th >= (1L << 30)) return 1;
    launch_halo_head(p, (int)th, (hipStream_t)stream);
    return 0;
  }
  if (a->epilogue != SF_CONV_BIAS && a->epilogue != SF_CONV_BIAS_RESID) return 1;
  if ((a->Cout & 7) != 0 || (a->out && (a->ldo & 7) != 0)) return 1;
  if (a->norm_out && !((a->Cout == 96 || a->Cout == 192) && a->norm_gamma && a->norm_ld >= a->Cout && (a->norm_ld & 7) == 0)) return 1;
  if (!a->out && !a->norm_out) return 1;
  int nt = a->Cout % 192 == 0 ? 6 : a->Cout % 96 == 0 ? 3 : 0;
  if (nt == 0) return 1;
  // small images (the 60 x 104 stages at one frame per call: 28 patches x 2 column tiles): the 96-column tile doubles the
  // number of workgroups; same arithmetic per output element (k order unchanged), so the result is bit-identical
  if (nt == 6 && !a->norm_out && (long)a->Tout * ((a->H + 15) / 16) * ((a->W + 15) / 16) * (a->Cout / 192) < 160) nt = 3;
  HaloP p;
  p.out_f32 = nullptr;
#ifdef SF_STAMP
  p.out_f32 = a->out_f32;
#endif
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w; p.bias = (const bf16_t*)a->bias;
  p.out = (bf16_t*)a->out; p.resid = (const bf16_t*)a->resid;
  p.norm_out = (bf16_t*)a->norm_out; p.norm_gamma = (const bf16_t*)a->norm_gamma; p.norm_ld = a->norm_ld; p.norm_frame0 = a->norm_frame_offset;
  p.H = a->H; p.W = a->W; p.Hin = a->Hin; p.Win = a->Win; p.up = a->upsample;
  p.Cin = a->Cin; p.Cout = a->Cout; p.cpt = a->Cin / 32; p.kt = a->kt; p.t_off = a->t_in_offset;
  p.ldw = a->ldw; p.ldo = a->ldo; p.ldr = a->ldr; p.out_frame0 = a->out_frame_offset;
  p.patches_h = (a->H + 15) / 16; p.patches_w = (a->W + 15) / 16;
  p.tiles_n = a->Cout / (32 * nt);
  const long xb = (long)(a->t_in_offset + a->Tout + a->kt - 1) * a->Hin * a->Win * a->Cin * 2;
  if (xb >= 0xFFFFFF00L) return 1;
  p.x_bytes = (unsigned)xb;
  const long tiles = (long)a->Tout * p.patches_h * p.patches_w * p.tiles_n;
  if (tiles >= (1L << 30)) return 1;
  if (nt == 3) launch_halo<3, 3>(p, (int)tiles, a->epilogue, (hipStream_t)stream);
  else launch_halo<6, 1>(p, (int)tiles, a->epilogue, (hipStream_t)stream);
  return 0;
}
