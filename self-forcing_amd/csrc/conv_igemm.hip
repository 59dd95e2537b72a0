// Implicit-GEMM convolution on channels-last bf16 volumes for gfx950 (MI355X).
//
// Replaces, on the VAE decode path, CausalConv3d (3x3x3, (3,1,1) and 1x1x1: wan/modules/vae.py:17-38),
// the per-frame nn.Conv2d 3x3 of Resample INCLUDING the nearest-neighbour 2x upsampling in front of it
// (vae.py:77-83, :139-141: the 4x larger tensor is never materialised), the bias add, the residual add
// of ResidualBlock (vae.py:221), the channel->frame interleave after the time convolution
// (vae.py:134-137) and the final float / clamp(-1, 1) of decode_to_pixel (utils/wan_wrapper.py:113).
//
//   out[(t,h,w)][n] = bias[n] + sum over taps (dt,dh,dw), ci of
//                     x[t + dt + t_off][(h + dh - kh/2) >> up][(w + dw - kw/2) >> up][ci] * wk[n][tap*Cin + ci]
//
// i.e. a GEMM with M = Tout*H*W output positions, N = Cout, K = taps*Cin whose A operand is GATHERED:
// K is walked in 32-channel slices (Cin % 32 == 0), two slices per 64-deep k-step, every slice lies
// inside one tap, and a lane's 16-byte piece of an A row comes from the tap-shifted position or -- for
// the zero padding in h/w, rows past M and the padding slice of an odd slice count -- from an offset past the
// end of the volume, which the range-checked LDS-DMA (buffer_load ... lds) turns into zeros.  Causality costs nothing here: the input volume holds the two history frames physically in
// front of the new ones (the caller keeps them there), so t + dt never leaves the buffer.
//
// Everything after the gather is the GEMM of gemm_bf16.hip: 128 x (32 NT) output tile per 256-thread
// workgroup, 4 waves as 2x2, 64 x (16 NT) per wave in 16x16x32 bf16 MFMAs, A and W tiles by LDS-DMA
// into two stages, 128-byte LDS rows with the chunk swizzle c ^ ((r>>1)&7) applied on the source side
// and on the ds_read_b128, operands swapped so that a lane owns 4 consecutive output channels.
// NT in {1,2,3,4,6} covers Cout = 3 (head) ... 96, 192, 384, 768 without padding waste.
#include <cstdlib>
#include "sf_common.h"
#include "../../include/sf_hip.h"

namespace {

constexpr int CBM = 128, CBK = 64;
constexpr int CONV_THREADS = 256;
constexpr int A_TILE_BYTES = CBM * CBK * 2;   // 16 KiB

struct ConvP {
  const bf16_t* x;
  const bf16_t* w;
  const bf16_t* bias;
  bf16_t* out;
  const bf16_t* resid;
  float* out_f32;
  int M, HW, H, W;
  int Hin, Win, up;
  int Cin, Cout, cpt, ntaps, khw, kw, ph, pw;
  int t_off, nk, ldw, ldo, ldr, out_frame0, inter_c, Tout;
  int tiles_m, tiles_n;
  unsigned x_bytes;   // size of the input volume the gather may touch (range check of the LDS-DMA)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <int NT, int EPI>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_igemm_kernel(ConvP p) {
  constexpr int BN = 32 * NT;
  constexpr int W_TILE_BYTES = BN * CBK * 2;
  constexpr int STAGE = A_TILE_BYTES + W_TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware bijective remap, then row-tile-major order: consecutive workgroups of one XCD work on
  // neighbouring output positions, whose gathered inputs overlap (taps) and share that XCD's L2
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
  const int m0 = tm * CBM, n0 = tn * BN;

  // ---- the four A pieces of this lane: row r of the tile, 16-byte chunk c of the 128-byte LDS row (chunks 0-3 hold
  // the first 32-channel slice of the k-step, 4-7 the second).  Everything that does not depend on the tap is
  // computed ONCE: the byte offset of the piece at tap (0, 0, 0) and a 9-bit mask of the spatial taps that fall
  // inside the image.  Per k-step a piece then costs a handful of VALU instructions: offset = base + D(slice) with D
  // wave-uniform, validity = one bit of the mask, and an invalid piece (zero padding, rows past M, the padding slice
  // of an odd slice count) gets an out-of-range offset: the range-checked buffer load writes zeros to LDS.
  unsigned abase[4], vmask[4], parh[4], parw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int m = m0 + r;
    const bool valid = m < p.M;
    const int mm = min(m, p.M - 1);
    const int t = mm / p.HW, hw = mm - t * p.HW;
    const int h = hw / p.W;
    const int hh0 = h - p.ph, ww0 = hw - h * p.W - p.pw;
    abase[i] = (unsigned)(((((long)(t + p.t_off) * p.Hin + (hh0 >> p.up)) * p.Win + (ww0 >> p.up)) * p.Cin + (c & 3) * 8) * 2);
    unsigned vm = 0;
    const int kk = p.khw == 9 ? 3 : 1;
    for (int dh = 0; dh < kk; ++dh)
      for (int dw = 0; dw < kk; ++dw)
        if (valid && (unsigned)(hh0 + dh) < (unsigned)p.H && (unsigned)(ww0 + dw) < (unsigned)p.W) vm |= 1u << (dh * 3 + dw);
    vmask[i] = vm;
    parh[i] = (unsigned)hh0 & 1u;
    parw[i] = (unsigned)ww0 & 1u;
  }
  // piece i lies in slice ((lane >> 2) & 1) ^ (i & 1) of the k-step (from the chunk swizzle above)
  const bool lane_hi = ((lane >> 2) & 1) != 0;
  const bf16_t* w_src[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int r = (wave * NT + j) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int n = min(n0 + r, p.Cout - 1);
    w_src[j] = p.w + (long)n * p.ldw + c * 8;
  }
  // range-checked view of the input volume (offsets past x_bytes read as zero)
  u32x4 x_srd;
  {
    const unsigned long long a64 = (unsigned long long)p.x;
    x_srd[0] = __builtin_amdgcn_readfirstlane((unsigned)a64);
    x_srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(a64 >> 32) & 0xFFFFu);
    x_srd[2] = __builtin_amdgcn_readfirstlane(p.x_bytes);
    x_srd[3] = 0x00020000u;
  }
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  auto dma_a = [&](unsigned voff, unsigned lds_addr) __attribute__((always_inline)) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(x_srd) : "memory");
  };

  // slice cursor of the NEXT stage to issue: slice 2 kt = (tap, cc), cc counting 32-channel groups
  int tap = 0, cc = 0;
  const unsigned rowB = (unsigned)(p.Win * p.Cin * 2), colB = (unsigned)(p.Cin * 2), frameB = (unsigned)(p.Hin * p.Win * p.Cin * 2);
  // byte offsets of this lane's four A pieces of the stage the cursor points at; advances the cursor
  auto gather_offsets = [&](unsigned (&voff)[4]) __attribute__((always_inline)) {
    int tap1 = tap, cc1 = cc + 1;
    if (cc1 >= p.cpt) { cc1 -= p.cpt; ++tap1; }
    int dt0, dh0, dw0, dt1, dh1, dw1;   // tap -> (dt, dh, dw): khw is 9 or 1, kw 3 or 1
    if (p.khw == 9) {
      dt0 = (tap * 57) >> 9; const int r0 = tap - 9 * dt0; dh0 = (r0 * 11) >> 5; dw0 = r0 - 3 * dh0;
      dt1 = (tap1 * 57) >> 9; const int r1 = tap1 - 9 * dt1; dh1 = (r1 * 11) >> 5; dw1 = r1 - 3 * dh1;
    } else {
      dt0 = tap; dh0 = dw0 = 0; dt1 = tap1; dh1 = dw1 = 0;
    }
    const unsigned sh0 = tap < p.ntaps ? (unsigned)(dh0 * 3 + dw0) : 31u, sh1 = tap1 < p.ntaps ? (unsigned)(dh1 * 3 + dw1) : 31u;
    const unsigned base0 = (unsigned)dt0 * frameB + (unsigned)cc * 64u, base1 = (unsigned)dt1 * frameB + (unsigned)cc1 * 64u;
    // the lane's two slices: pieces 0, 2 use slice `lane_hi`, pieces 1, 3 the other one
    const unsigned shA = lane_hi ? sh1 : sh0, shB = lane_hi ? sh0 : sh1;
    if (p.up == 0) {
      const unsigned d0 = base0 + (unsigned)dh0 * rowB + (unsigned)dw0 * colB, d1 = base1 + (unsigned)dh1 * rowB + (unsigned)dw1 * colB;
      const unsigned dA = lane_hi ? d1 : d0, dB = lane_hi ? d0 : d1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned bad = ((vmask[i] >> ((i & 1) ? shB : shA)) & 1u) - 1u;      // 0 if the tap is inside, ~0 if not
        voff[i] = (abase[i] + ((i & 1) ? dB : dA)) | (bad & 0xFFFFFFF0u);          // (no select: keeps the code branch-free)
      }
    } else {   // fused nearest 2x upsample: the input row / column of a tap depends on the parity of the output position
      const unsigned tA = lane_hi ? base1 : base0, tB = lane_hi ? base0 : base1;
      const unsigned dhA = lane_hi ? dh1 : dh0, dhB = lane_hi ? dh0 : dh1, dwA = lane_hi ? dw1 : dw0, dwB = lane_hi ? dw0 : dw1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned bad = ((vmask[i] >> ((i & 1) ? shB : shA)) & 1u) - 1u;
        const unsigned rdh = (((i & 1) ? dhB : dhA) + parh[i]) >> 1, rdw = (((i & 1) ? dwB : dwA) + parw[i]) >> 1;
        voff[i] = (abase[i] + ((i & 1) ? tB : tA) + rdh * rowB + rdw * colB) | (bad & 0xFFFFFFF0u);
      }
    }
    cc += 2;                                   // advance the cursor by two slices
    if (cc >= p.cpt) { cc -= p.cpt; ++tap; }
    if (cc >= p.cpt) { cc -= p.cpt; ++tap; }
  };

  // ---- fragment read addresses
  const int wr = wave >> 1, wc = wave & 1;
  const int i16 = lane & 15, kq = lane >> 4;
  const int swz = (i16 >> 1) & 7;
  const int x_row_off = (wr * 64 + i16) * 128;                              // + t*2048
  const int w_row_off = A_TILE_BYTES + (wc * (16 * NT) + i16) * 128;       // + nt*2048
  const int coff0 = ((0 + kq) ^ swz) << 4;
  const int coff1 = ((4 + kq) ^ swz) << 4;

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned voff[4];                         // offsets of the stage that the NEXT k-step requests
  {   // prologue: stage 0 requested, the offsets of stage 1 computed
    gather_offsets(voff);
    const unsigned abase_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * 4096));
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_a(voff[i], abase_lds + i * 1024);
    char* wbase = smem + A_TILE_BYTES + wave * (NT * 1024);
#pragma unroll
    for (int j = 0; j < NT; ++j) glds16(w_src[j], wbase + j * 1024);
    if (p.nk > 1) {
      gather_offsets(voff);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) voff[i] = 0xFFFFFFF0u;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();

  for (int kt = 0; kt < p.nk; ++kt) {
    const int cur = kt & 1;
    const char* buf = smem + cur * STAGE;
    // The k-step as 4 + NT pinned slices of {MFMAs of the first 32-deep sub-step, one fragment read of the second,
    // ONE LDS-DMA request of the next stage} (see gemm_bf16.hip), then the second sub-step's MFMAs, under which the
    // gather offsets of the stage after next are computed.  The last k-step re-requests its own W pieces and
    // all-invalid A pieces into the idle buffer so that the body stays branch-free.
    const unsigned abase_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((cur ^ 1) * STAGE + wave * 4096));
    char* wbase = smem + (cur ^ 1) * STAGE + A_TILE_BYTES + wave * (NT * 1024);
    const int kn = min(kt + 1, p.nk - 1) * CBK;
    bf16x8 xf0[4], xf1[4], wf0[NT], wf1[NT];
#pragma unroll
    for (int t = 0; t < 4; ++t) xf0[t] = *reinterpret_cast<const bf16x8*>(buf + x_row_off + t * 2048 + coff0);
#pragma unroll
    for (int t = 0; t < NT; ++t) wf0[t] = *reinterpret_cast<const bf16x8*>(buf + w_row_off + t * 2048 + coff0);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NS = 4 + NT;               // slices
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
#pragma unroll
      for (int q = (4 * NT * sl) / NS; q < (4 * NT * (sl + 1)) / NS; ++q) {
        const int mt = q / NT, nt = q - mt * NT;
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt], xf0[mt], acc[mt][nt], 0, 0, 0);
      }
      if (sl < 4) {
        xf1[sl] = *reinterpret_cast<const bf16x8*>(buf + x_row_off + sl * 2048 + coff1);
        dma_a(voff[sl], abase_lds + sl * 1024);
      } else {
        wf1[sl - 4] = *reinterpret_cast<const bf16x8*>(buf + w_row_off + (sl - 4) * 2048 + coff1);
        glds16(w_src[sl - 4] + kn, wbase + (sl - 4) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 2 < p.nk) {
      gather_offsets(voff);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) voff[i] = 0xFFFFFFF0u;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], xf1[mt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // drains the in-flight LDS-DMA (vmcnt(0)) and orders the stage swap
  }

  // ---- epilogue: the lane holds out[m][n .. n+3] for (mt, nt); m = mrow + 16 mt, n = ncol + 16 nt
  const int mrow = m0 + wr * 64 + (lane & 15);
  const int ncol = n0 + wc * (16 * NT) + (lane >> 4) * 4;
  if (EPI != SF_CONV_BIAS_CLAMP_F32 && p.inter_c == 0 && (p.Cout & 7) == 0 && (p.ldo & 7) == 0) {
    // Through LDS: the 128 x BN tile is assembled as bf16 rows (padded by 16 B against bank conflicts) and
    // written back in 16-byte pieces along the rows -- with all channels in one tile that is one contiguous
    // region of the output volume.  The direct form stores 8-byte pieces of 16 different positions per
    // instruction (32-byte partial lines): a quarter of HBM's write efficiency on the 300 MB volumes.
    constexpr int RBP = 64 * NT + 16;    // padded row bytes
    char* obuf = smem;                   // (the k-loop's last __syncthreads has released the stages)
    // bias and residual are requested for the whole wave tile first and consumed afterwards (one memory round trip
    // instead of 4 NT dependent ones, see gemm_epilogue_lds)
    int ncol[NT];
    bf16x4 bias_v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      ncol[nt] = min(n0 + wc * (16 * NT) + nt * 16 + (lane >> 4) * 4, p.Cout - 4);
      bias_v[nt] = *reinterpret_cast<const bf16x4*>(p.bias + ncol[nt]);
    }
    bf16x4 rv[4][NT];
    if (EPI == SF_CONV_BIAS_RESID) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int m = min(m0 + wr * 64 + mt * 16 + (lane & 15), p.M - 1);
        const long grow = (long)p.out_frame0 * p.HW + m;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rv[mt][nt] = *reinterpret_cast<const bf16x4*>(p.resid + grow * p.ldr + ncol[nt]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = wr * 64 + mt * 16 + (lane & 15);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = wc * (16 * NT) + nt * 16 + (lane >> 4) * 4;
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
    __syncthreads();
    constexpr int CPR = 4 * NT;          // 16-byte chunks per row
#pragma unroll
    for (int i = 0; i < (CBM * CPR) / CONV_THREADS; ++i) {
      const int id = i * CONV_THREADS + tid;
      const int row = id / CPR, ch = id - row * CPR;
      const int m = m0 + row, n = n0 + ch * 8;
      if (m < p.M && n < p.Cout) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(obuf + row * RBP + ch * 16);
        *reinterpret_cast<bf16x8*>(p.out + ((long)p.out_frame0 * p.HW + m) * p.ldo + n) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = mrow + mt * 16;
    if (m >= p.M) continue;
    const int t = m / p.HW, hw = m - t * p.HW;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = ncol + nt * 16;
      if (n >= p.Cout) continue;
      float y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = acc[mt][nt][j];
      if (EPI == SF_CONV_BIAS_CLAMP_F32) {
        // Cout is tiny (3): per-element guards, planar float output [Tout][Cout][H][W]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (n + j < p.Cout) {
            const float v = y[j] + (float)p.bias[n + j];
            p.out_f32[((long)t * p.Cout + n + j) * p.HW + hw] = fminf(fmaxf(v, -1.f), 1.f);
          }
        }
        continue;
      }
      const bf16x4 b = *reinterpret_cast<const bf16x4*>(p.bias + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] += (float)b[j];
      // channel -> frame interleave of the time convolution: channels [0,C) are frame 2t, [C,2C) frame 2t+1
      int tf = t, nn = n;
      if (p.inter_c > 0) {
        const int sel = n >= p.inter_c ? 1 : 0;
        tf = 2 * t + sel;
        nn = n - sel * p.inter_c;
      }
      const long row = (long)(p.out_frame0 + tf) * p.HW + hw;
      if (EPI == SF_CONV_BIAS_RESID) {
        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(p.resid + row * p.ldr + nn);
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] += (float)rv[j];
      }
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16_t)y[j];
      *reinterpret_cast<bf16x4*>(p.out + row * p.ldo + nn) = o;
    }
  }
}

template <int NT>
int launch_nt(const ConvP& p, int epi, hipStream_t s) {
  constexpr int LDS = 2 * (A_TILE_BYTES + 32 * NT * CBK * 2);
  const dim3 grid(p.tiles_m * p.tiles_n), block(CONV_THREADS);
  if (LDS > 64 * 1024) {   // above the default dynamic-LDS limit: opt in once per kernel
    static bool done = false;
    if (!done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<NT, SF_CONV_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<NT, SF_CONV_BIAS_RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<NT, SF_CONV_BIAS_CLAMP_F32>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      done = true;
    }
  }
  switch (epi) {
    case SF_CONV_BIAS: hipLaunchKernelGGL((conv_igemm_kernel<NT, SF_CONV_BIAS>), grid, block, LDS, s, p); break;
    case SF_CONV_BIAS_RESID: hipLaunchKernelGGL((conv_igemm_kernel<NT, SF_CONV_BIAS_RESID>), grid, block, LDS, s, p); break;
    case SF_CONV_BIAS_CLAMP_F32: hipLaunchKernelGGL((conv_igemm_kernel<NT, SF_CONV_BIAS_CLAMP_F32>), grid, block, LDS, s, p); break;
    default: return -1;
  }
  return 0;
}

}  // namespace

extern "C" int sf_conv_pick_nt(int cout) {
  // the per-wave column count (16 NT) that pads Cout least; ties go to the larger tile
  const int cand[5] = {6, 4, 3, 2, 1};
  int best = 1;
  long best_cost = -1;
  for (int i = 0; i < 5; ++i) {
    const int bn = 32 * cand[i];
    const long cost = (long)((cout + bn - 1) / bn) * bn;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cand[i]; }
  }
  return best;
}

int sf_conv_halo_launch(const sf_conv_args* a, void* stream);   // conv_halo.hip: 1 = outside its domain

extern "C" int sf_conv_igemm(const sf_conv_args* a, void* stream) {
  SF_CHECK(a != nullptr, "sf_conv_igemm: null args");
  SF_CHECK(a->structure >= SF_CONV_AUTO && a->structure <= SF_CONV_HALO, "sf_conv_igemm: unknown structure %d", a->structure);
  SF_CHECK(a->x && a->w && a->bias, "sf_conv_igemm: null tensor");
  SF_CHECK(!a->norm_out || a->structure != SF_CONV_IGEMM, "sf_conv_igemm: the fused norm output exists in the halo kernel only");
  SF_CHECK(a->Tout > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0, "sf_conv_igemm: empty problem");
  SF_CHECK(a->Cin % 32 == 0, "sf_conv_igemm: Cin=%d must be a multiple of 32 (pad the channels)", a->Cin);
  SF_CHECK((a->kh == 3 && a->kw == 3) || (a->kh == 1 && a->kw == 1), "sf_conv_igemm: spatial taps must be 3x3 or 1x1");
  SF_CHECK(a->kt == 1 || a->kt == 3, "sf_conv_igemm: kt must be 1 or 3");
  SF_CHECK(a->upsample == 0 || a->upsample == 1, "sf_conv_igemm: upsample must be 0 or 1");
  SF_CHECK(a->Hin == (a->upsample ? a->H / 2 : a->H) && a->Win == (a->upsample ? a->W / 2 : a->W) &&
           (!a->upsample || (a->H % 2 == 0 && a->W % 2 == 0)), "sf_conv_igemm: input size %dx%d does not match output %dx%d", a->Hin, a->Win, a->H, a->W);
  const int taps = a->kt * a->kh * a->kw;
  const int slices = taps * (a->Cin / 32);
  const int nk = (slices + 1) / 2;
  SF_CHECK(a->ldw >= nk * 64 && a->ldw % 8 == 0, "sf_conv_igemm: weight row stride %d < padded K %d", a->ldw, nk * 64);
  SF_CHECK((long)a->Tout * a->H * a->W < (1L << 31), "sf_conv_igemm: too many output positions");
  SF_CHECK(a->t_in_offset >= 0, "sf_conv_igemm: negative input frame offset");
  SF_CHECK(((uintptr_t)a->x % 16 == 0) && ((uintptr_t)a->w % 16 == 0) && ((uintptr_t)a->bias % 8 == 0), "sf_conv_igemm: misaligned tensor");
  if (a->epilogue == SF_CONV_BIAS_CLAMP_F32) {
    SF_CHECK(a->out_f32 != nullptr && a->interleave_c == 0, "sf_conv_igemm: float epilogue needs out_f32 and no interleave");
  } else {
    SF_CHECK((a->out != nullptr || a->norm_out != nullptr) && a->Cout % 4 == 0, "sf_conv_igemm: bf16 output needs out (or norm_out) and Cout %% 4 == 0");
    SF_CHECK(!a->out || (a->ldo % 4 == 0 && (uintptr_t)a->out % 8 == 0), "sf_conv_igemm: misaligned output / ldo %% 4 != 0");
    SF_CHECK(!a->norm_out || ((uintptr_t)a->norm_out % 16 == 0 && (uintptr_t)a->norm_gamma % 16 == 0), "sf_conv_igemm: misaligned norm output");
    SF_CHECK(a->interleave_c == 0 || (a->interleave_c * 2 == a->Cout && a->interleave_c % 4 == 0), "sf_conv_igemm: interleave_c must be Cout/2");
    SF_CHECK(!a->out || a->ldo >= (a->interleave_c ? a->interleave_c : a->Cout), "sf_conv_igemm: ldo too small");
    if (a->epilogue == SF_CONV_BIAS_RESID)
      SF_CHECK(a->resid != nullptr && a->ldr % 4 == 0 && a->ldr >= a->Cout && a->interleave_c == 0, "sf_conv_igemm: residual epilogue needs resid/ldr");
  }
  if (a->structure != SF_CONV_IGEMM) {
    // 3 x 3 convolutions with 96 k / 192 k output channels: the halo-tile kernel (the A operand staged once per
    // (channel slice, frame) instead of once per tap), conv_halo.hip
    const int rc = sf_conv_halo_launch(a, stream);
    SF_CHECK(rc >= 0, "sf_conv_igemm: halo kernel failed");
    SF_CHECK(rc == 0 || (a->structure == SF_CONV_AUTO && !a->norm_out), "sf_conv_igemm: the halo structure needs 3x3 spatial taps, "
             "Cout %% 96 == 0 (a fused norm output: Cout 96 or 192), H, W >= 16, a bf16 bias / bias + residual epilogue and no interleave");
    if (rc == 0) {
      SF_HIP_LAUNCH_CHECK("sf_conv_igemm");
      return 0;
    }
  }
  ConvP p;
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w; p.bias = (const bf16_t*)a->bias;
  p.out = (bf16_t*)a->out; p.resid = (const bf16_t*)a->resid; p.out_f32 = a->out_f32;
  p.HW = a->H * a->W; p.M = a->Tout * p.HW; p.H = a->H; p.W = a->W; p.Tout = a->Tout;
  p.Hin = a->Hin; p.Win = a->Win; p.up = a->upsample;
  p.Cin = a->Cin; p.Cout = a->Cout; p.cpt = a->Cin / 32; p.ntaps = taps; p.khw = a->kh * a->kw; p.kw = a->kw;
  p.ph = a->kh / 2; p.pw = a->kw / 2;
  p.t_off = a->t_in_offset; p.nk = nk; p.ldw = a->ldw; p.ldo = a->ldo; p.ldr = a->ldr;
  p.out_frame0 = a->out_frame_offset; p.inter_c = a->interleave_c;
  {   // frames [0, t_in_offset + Tout + kt - 1) of the input volume can be gathered from
    const long xb = (long)(a->t_in_offset + a->Tout + a->kt - 1) * a->Hin * a->Win * a->Cin * 2;
    SF_CHECK(xb < 0xFFFFFF00L, "sf_conv_igemm: input volume of %ld bytes exceeds the 4 GiB the gather's 32-bit offsets cover", xb);
    p.x_bytes = (unsigned)xb;
  }
  const int nt = sf_conv_pick_nt(a->Cout);
  p.tiles_m = (p.M + CBM - 1) / CBM;
  p.tiles_n = (a->Cout + 32 * nt - 1) / (32 * nt);
  hipStream_t s = (hipStream_t)stream;
  int rc = 0;
  switch (nt) {
    case 1: rc = launch_nt<1>(p, a->epilogue, s); break;
    case 2: rc = launch_nt<2>(p, a->epilogue, s); break;
    case 3: rc = launch_nt<3>(p, a->epilogue, s); break;
    case 4: rc = launch_nt<4>(p, a->epilogue, s); break;
    default: rc = launch_nt<6>(p, a->epilogue, s); break;
  }
  SF_CHECK(rc == 0, "sf_conv_igemm: unknown epilogue %d", a->epilogue);
  SF_HIP_LAUNCH_CHECK("sf_conv_igemm");
  return 0;
}
