// Non-causal flash attention forward for gfx950 (MI355X), head_dim = 128, bf16 in/out.
//
// Replaces flash_attn_varlen_func / SDPA on the reference's hot path
// (wan/modules/attention.py:136-150, :198) for self-attention over the growing KV cache
// (wan/modules/causal_model.py:230-234; Lk = 1560 ... 32760) and T5 cross-attention
// (wan/modules/model.py:189; Lk = 512).  No mask: causality comes only from what is in the cache.
//
// Structure (v1).  One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns
// 32 query rows and walks the keys in tiles of 64.  Per tile and wave:
//   S^T[key][q]  = K_tile . Q^T        16 x v_mfma_f32_32x32x16_bf16   (A = K rows from LDS,
//                                                                       B = Q fragments in VGPRs)
//   online softmax on S^T: the query index is the LANE (col = lane & 31), so the running max /
//   sum of a query row live in one lane pair (l, l+32) -- one cross-half shuffle per tile, no LDS;
//   O^T[d][q]   += V_tile^T . P^T      16 x v_mfma_f32_32x32x16_bf16   (A = V^T via
//                   ds_read_b64_tr_b16 transposing reads, B = the S^T accumulator registers
//                   converted to bf16 in place: an accumulator tile is directly the next MFMA's
//                   B operand when the product sums over its row index).
// K and V tiles are register-staged HBM -> VGPR -> LDS (loads for tile t+1 are issued before the
// MFMAs of tile t and written to the other LDS buffer afterwards; one barrier per tile).
// LDS image of a [64 keys][128 d] bf16 tile: 256-byte rows, the 16-byte chunk `ch` of row `row`
// lives at chunk ch ^ (((row&3)<<2) | ((row>>2)&3)); this one image serves the row reads
// (ds_read_b128, K) and the transposed reads (V) without bank conflicts.
#include <type_traits>
#include "sf_common.h"
#include "../../include/sf_hip.h"
#ifndef SF_R64_INC
#define SF_R64_INC "attention_r64_asm.inc"   // timing-only ablation builds substitute their own (tools/gen_attention_r64.py --abl)
#endif
#include SF_R64_INC

namespace {

constexpr int HD = 128;          // head dim
constexpr int QT = 128;          // query rows per workgroup
constexpr int KT = 64;           // keys per tile
constexpr int ATT_THREADS = 256;
constexpr int TILE_B = KT * HD * 2;          // 16 KiB
constexpr int ATT_LDS = 2 * 2 * TILE_B;      // {K,V} x 2 buffers = 64 KiB

struct AttP {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;
  int B, H, Lq, Lk;
  long q_stride, q_bstride, kv_stride, kv_bstride, o_stride, o_bstride;
  int q_tiles;
  float scale_log2;  // (1/sqrt(D)) * log2(e)
};

__device__ __forceinline__ int lds_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

__device__ __forceinline__ bf16x4 lds_tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
}

// KIND only gives self-attention (0: keys = the KV cache) and cross-attention (1: keys = the text
// context) distinct symbols, so per-kernel profiles do not mix a 32760-key launch with a 512-key one.
template <int KIND>
__global__ __launch_bounds__(ATT_THREADS, 2) void attention_kernel(AttP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware bijective remap so that workgroups on one XCD share (batch, head) -> K/V hit in L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int bh = wg / p.q_tiles, qt = wg - bh * p.q_tiles;
  const int b = bh / p.H, head = bh - b * p.H;

  const bf16_t* qbase = p.q + (long)b * p.q_bstride + head * HD;
  const bf16_t* kbase = p.k + (long)b * p.kv_bstride + head * HD;
  const bf16_t* vbase = p.v + (long)b * p.kv_bstride + head * HD;
  bf16_t* obase = p.o + (long)b * p.o_bstride + head * HD;

  const int r32 = lane & 31, hh = lane >> 5;
  const int qrow = qt * QT + wave * 32 + r32;
  const int qrow_c = min(qrow, p.Lq - 1);

  // ---- Q fragments (B operand of S^T = K . Q^T): lane holds Q[q][16 s + 8 hh + j]
  bf16x8 qf[8];
  {
    const bf16_t* qp = qbase + (long)qrow_c * p.q_stride + 8 * hh;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }

  // ---- staging geometry: thread t moves chunks idx = t + 256 i, i = 0..3 of each 16 KiB tile
  const int st_row = tid >> 4;        // + 16 i
  const int st_ch = tid & 15;
  int st_lds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) st_lds[i] = lds_off(st_row + 16 * i, st_ch);

  // K/V tiles come in through buffer loads: the per-lane byte offset is loop invariant, the tile
  // advances through the scalar offset, and rows past Lk fall outside the descriptor's range and
  // read as zero (no clamping, no per-tile address arithmetic).  The range of one (batch, head)
  // slab is (Lk-1) * stride + 128 elements < 2^31 bytes (checked on the host).
  const unsigned kv_bytes = (unsigned)(((long)(p.Lk - 1) * p.kv_stride + HD) * 2);
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kbase), 0, kv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vbase), 0, kv_bytes, 0x00020000);
  unsigned st_goff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) st_goff[i] = (unsigned)(((long)(st_row + 16 * i) * p.kv_stride + st_ch * 8) * 2);
  const unsigned tile_bytes = (unsigned)((long)KT * p.kv_stride * 2);
  u32x4 kreg[4], vreg[4];
  auto load_tile = [&](int t) {
    const unsigned soff = (unsigned)t * tile_bytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, st_goff[i], soff, 0);
      vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, st_goff[i], soff, 0);
    }
  };
  auto write_tile = [&](int buf) {
    char* kb = smem + buf * (2 * TILE_B);
    char* vb = kb + TILE_B;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(kb + st_lds[i]) = kreg[i];
      *reinterpret_cast<u32x4*>(vb + st_lds[i]) = vreg[i];
    }
  };

  // ---- fragment read offsets: per-lane bases, everything else is a compile-time immediate.
  // K rows: row = kb*32 + r32, chunk (2s + hh) ^ swz(r32) = (2s) ^ x with x = hh ^ swz(r32).
  int k_addr[8];
  {
    const int x = hh ^ (((r32 & 3) << 2) | ((r32 >> 2) & 3));
#pragma unroll
    for (int s = 0; s < 8; ++s) k_addr[s] = 256 * r32 + 16 * ((2 * s) ^ x);
  }
  // V transposed reads (ds_read_b64_tr_b16): 16-lane group g covers d = db*32 + 16 (g&1) .. +15 and
  // keys r0 .. r0+3 with r0 = 16 ks + 4 hh (elements 0-3) or + 8 (elements 4-7); lane 4q+p of the
  // group addresses row r0+q, columns 4p..4p+3.  With y = 2 (g&1) + (p>>1):
  //   lo: 256 (4hh + q)     + 16 (4 (db ^ q) + (y ^ hh))       + 8 (p&1) + 4096 ks
  //   hi: 256 (8 + 4hh + q) + 16 (4 (db ^ q) + (y ^ (hh + 2))) + 8 (p&1) + 4096 ks
  int v_lo[4], v_hi[4];
  {
    const int g = lane >> 4, i16 = lane & 15;
    const int tq = i16 >> 2, tp = i16 & 3;
    const int y = 2 * (g & 1) + (tp >> 1);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      v_lo[db] = 256 * (4 * hh + tq) + 16 * (4 * (db ^ tq) + (y ^ hh)) + 8 * (tp & 1);
      v_hi[db] = 256 * (8 + 4 * hh + tq) + 16 * (4 * (db ^ tq) + (y ^ (hh + 2))) + 8 * (tp & 1);
    }
  }

  f32x16 o_acc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.scale_log2;

  const int ntiles = (p.Lk + KT - 1) / KT;
  load_tile(0);
  write_tile(0);
  __syncthreads();

  auto process_tile = [&](int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int cur = t & 1;
    const char* kb_lds = smem + cur * (2 * TILE_B);
    const char* vb_lds = kb_lds + TILE_B;

    // ---- S^T = K . Q^T : all 16 K fragments are requested up front, the MFMA chains drain them
    bf16x8 kf[2][8];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 8; ++s) kf[kb][s] = *reinterpret_cast<const bf16x8*>(kb_lds + kb * 8192 + k_addr[s]);
    // prefetch the next K/V tile into registers (the last iteration re-loads its own tile: harmless,
    // and keeping the loop body branch-free keeps the compiler's s_waitcnt bookkeeping exact)
    load_tile(min(t + 1, ntiles - 1));
    f32x16 st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][s], qf[s], st[kb], 0, 0, 0);
    }

    // ---- first half of the V^T fragments: requested now, they land while the softmax runs
    bf16x4 vlo[2][4], vhi[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        vlo[ks][db] = lds_tr_read(vb_lds + ks * 4096 + v_lo[db]);
        vhi[ks][db] = lds_tr_read(vb_lds + ks * 4096 + v_hi[db]);
      }

    // ---- mask the tail keys of the last tile
    if (MASK) {
      const int key0 = t * KT + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= p.Lk) st[kb][r] = -1e30f;
        }
    }

    // ---- online softmax (per query = per lane pair l, l+32)
    float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(st[0][r], st[1][r]));
    {
      const unsigned mi = __float_as_uint(mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mi, mi, false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (__any(mx > m_run)) {  // wave-uniform: some row's running max grew -> rescale what is at the old max
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
    }
    const float mc = m_run * c;
    float lsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(st[kb][r] * c - mc);
        st[kb][r] = pv;
        lsum += pv;
      }
    l_run += lsum;

    // ---- P^T fragments (B operand): k-step ks <- registers 8 (ks&1) .. +7 of st[ks>>1]
    bf16x8 pf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[ks][j] = (bf16_t)st[ks >> 1][8 * (ks & 1) + j];

    // ---- O^T += V^T . P^T
    auto pv_step = [&](int ks, const bf16x4 (&lo)[4], const bf16x4 (&hi)[4]) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        bf16x8 vf;
        vf[0] = lo[db][0]; vf[1] = lo[db][1]; vf[2] = lo[db][2]; vf[3] = lo[db][3];
        vf[4] = hi[db][0]; vf[5] = hi[db][1]; vf[6] = hi[db][2]; vf[7] = hi[db][3];
        o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], o_acc[db], 0, 0, 0);
      }
    };
    bf16x4 wlo[2][4], whi[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        wlo[ks][db] = lds_tr_read(vb_lds + (ks + 2) * 4096 + v_lo[db]);
        whi[ks][db] = lds_tr_read(vb_lds + (ks + 2) * 4096 + v_hi[db]);
      }
    pv_step(0, vlo[0], vhi[0]);
    pv_step(1, vlo[1], vhi[1]);
    pv_step(2, wlo[0], whi[0]);
    pv_step(3, wlo[1], whi[1]);

    write_tile(cur ^ 1);
    __syncthreads();
  };
  const bool tail = (p.Lk & (KT - 1)) != 0;
  const int nfull = tail ? ntiles - 1 : ntiles;
  for (int t = 0; t < nfull; ++t) process_tile(t, std::false_type{});
  if (tail) process_tile(ntiles - 1, std::true_type{});

  // ---- epilogue: O[q][d] = O^T[d][q] / l
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qrow < p.Lq) {
    bf16_t* op = obase + (long)qrow * p.o_stride + 4 * hh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (bf16_t)(o_acc[db][4 * rg + j] * inv);
        *reinterpret_cast<bf16x4*>(op + db * 32 + rg * 8) = w;
      }
  }
}


// ------------------------------------------------------------------------------------------
// 8-wave structure (used when it fills the chip): ONE workgroup per CU = 256 query rows of one
// (batch, head).  Waves 0-3 (half A) and 4-7 (half B) share the four SIMDs pairwise (w, w+4).  The
// per-tile work of a wave is cut into a MATRIX segment and a VECTOR segment,
//     matrix(t):  O^T += V^T(t-1) . P^T(t-1)   then   S^T(t) = K(t) . Q^T      (32 MFMAs, ~0 VALU)
//     vector(t):  online softmax of S^T(t) -> P^T(t) in bf16                   (~130 VALU, 0 MFMA)
// and the halves run them in anti-phase, one s_barrier per global step g:
//     g = 2t   : A matrix(t)   | B vector(t-1)
//     g = 2t+1 : A vector(t)   | B matrix(t)
// so every SIMD always has one wave feeding the matrix pipe and its partner issuing VALU
// (an MFMA blocks the SIMD's vector issue for only 8 of its 32 cycles).  Two independent 4-wave
// workgroups per CU drift into lockstep instead (measured: 45 % MFMA-busy, 35 % co-execution), and
// pairing "QK^T + softmax" with "PV" does not help either: the two MFMA streams share the pipe,
// finish together and leave the softmax alone.
// (A 64-rows-per-wave variant -- every K/V fragment feeding two MFMAs, 19 instead of 34 LDS bytes
// per kflop, K/V by LDS-DMA into a 3-deep ring, the softmax software-pipelined over 32-key units
// inside the single wave per SIMD -- was built twice this round and passes the same tests, but does
// not ship: plain C++ gives 557 TFLOP/s (hipcc parks accumulators in AGPRs and pays hundreds of
// v_accvgpr moves per tile); with inline-asm MFMAs pinning O to AGPRs it reaches 645-750, and the
// counters say why: 556 VALU instructions per tile and wave (~170 of them register shuffles) make it
// VALU-issue bound.  It needs a hand-allocated instruction stream; left for a later round.)
// LDS: K and V double-buffered separately (64 KiB).  K(t) and V(t-1) are read in steps 2t (A) and
// 2t+1 (B); K(t+1) and V(t) are requested by LDS-DMA in step 2t into the buffers of K(t-1) / V(t-2),
// both dead since step 2t-1, and must have landed by the barrier that ends step 2t+1.
// Measured shares at Lk = 32760 (compile-time ablation builds, -DSF_ABL_*): softmax 18 %, K/V
// staging 10 % with register staging (VGPR + ds_write_b128) -> 7 % with LDS-DMA, the rest is the
// MFMA + fragment-read skeleton.
constexpr int QT8 = 256;
constexpr int ATT8_THREADS = 512;
constexpr int ATT8_LDS = 4 * TILE_B;  // K0 K1 V0 V1 = 64 KiB

template <int KIND>
__global__ __launch_bounds__(ATT8_THREADS, 2) void attention_w8_kernel(AttP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2;  // 0 = A, 1 = B

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int bh = wg / p.q_tiles, qt = wg - bh * p.q_tiles;
  const int b = bh / p.H, head = bh - b * p.H;

  const bf16_t* qbase = p.q + (long)b * p.q_bstride + head * HD;
  const bf16_t* kbase = p.k + (long)b * p.kv_bstride + head * HD;
  const bf16_t* vbase = p.v + (long)b * p.kv_bstride + head * HD;
  bf16_t* obase = p.o + (long)b * p.o_bstride + head * HD;

  const int r32 = lane & 31, hh = lane >> 5;
  const int qrow = qt * QT8 + wave * 32 + r32;
  const int qrow_c = min(qrow, p.Lq - 1);

  bf16x8 qf[8];
  {
    const bf16_t* qp = qbase + (long)qrow_c * p.q_stride + 8 * hh;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }

  // staging by LDS-DMA (`buffer_load_dwordx4 ... offen lds`, range-checked: rows past Lk read zero):
  // a K or V tile is 16 pieces of 1 KiB (4 rows x 256 B); wave w issues pieces 2w, 2w+1 of each.
  // Lane l of a piece lands at byte 16 l (lane-linear) = row l>>4, slot l&15, so it FETCHES chunk
  // slot ^ swz(row): the swizzle sits on the source side.  Issued from inline asm: with the builtin
  // hipcc puts `s_waitcnt vmcnt(0)` in front of every later ds_read (it cannot tell the buffers apart).
  // No staging registers, no ds_write; completion = one vmcnt(0) before the barrier of the next step.
  const unsigned kv_bytes = (unsigned)(((long)(p.Lk - 1) * p.kv_stride + HD) * 2);
  auto make_srd = [&](const bf16_t* base) {
    const unsigned long long a64 = (unsigned long long)base;
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((unsigned)a64);
    d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a64 >> 32) & 0xFFFFu);
    d[2] = __builtin_amdgcn_readfirstlane(kv_bytes);
    d[3] = 0x00020000u;
    return d;
  };
  const u32x4 k_srd = make_srd(kbase), v_srd = make_srd(vbase);
  unsigned dma_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    dma_off[i] = (unsigned)(((long)row * p.kv_stride + chunk * 8) * 2);
  }
  const unsigned tile_bytes = (unsigned)((long)KT * p.kv_stride * 2);
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  auto dma16 = [&](const u32x4& srd, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff) : "memory");
  };
  auto dma_k = [&](int t) {   // K(t) -> buffer t & 1
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((t & 1) * TILE_B + wave * 2048));
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)t * tile_bytes);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(k_srd, dma_off[i], soff, base + i * 1024);
  };
  auto dma_v = [&](int t) {   // V(t) -> buffer 2 + (t & 1)
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((2 + (t & 1)) * TILE_B + wave * 2048));
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)t * tile_bytes);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(v_srd, dma_off[i], soff, base + i * 1024);
  };

  int k_addr[8];
  {
    const int x = hh ^ (((r32 & 3) << 2) | ((r32 >> 2) & 3));
#pragma unroll
    for (int s = 0; s < 8; ++s) k_addr[s] = 256 * r32 + 16 * ((2 * s) ^ x);
  }
  int v_lo[4], v_hi[4];
  {
    const int g16 = lane >> 4, i16 = lane & 15;
    const int tq = i16 >> 2, tp = i16 & 3;
    const int y = 2 * (g16 & 1) + (tp >> 1);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      v_lo[db] = 256 * (4 * hh + tq) + 16 * (4 * (db ^ tq) + (y ^ hh)) + 8 * (tp & 1);
      v_hi[db] = 256 * (8 + 4 * hh + tq) + 16 * (4 * (db ^ tq) + (y ^ (hh + 2))) + 8 * (tp & 1);
    }
  }

  f32x16 o_acc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.scale_log2;
  bf16x8 pf[4];
  f32x16 st[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[ks][j] = (bf16_t)0.f;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;

  const int ntiles = (p.Lk + KT - 1) / KT;
  const bool tail = (p.Lk & (KT - 1)) != 0;

  // ---- matrix segment
  auto seg_pv = [&](int t) {   // O^T += V^T(t) . P^T
    const char* vb_lds = smem + (2 + (t & 1)) * TILE_B;
    bf16x4 vlo[4][4], vhi[4][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
#ifdef SF_ABL_NOTR   // timing-only: one 16-byte row read instead of two transposed 8-byte reads
        const bf16x8 w8 = *reinterpret_cast<const bf16x8*>(vb_lds + ks * 4096 + (v_lo[db] & ~15));
        vlo[ks][db][0] = w8[0]; vlo[ks][db][1] = w8[1]; vlo[ks][db][2] = w8[2]; vlo[ks][db][3] = w8[3];
        vhi[ks][db][0] = w8[4]; vhi[ks][db][1] = w8[5]; vhi[ks][db][2] = w8[6]; vhi[ks][db][3] = w8[7];
#else
        vlo[ks][db] = lds_tr_read(vb_lds + ks * 4096 + v_lo[db]);
        vhi[ks][db] = lds_tr_read(vb_lds + ks * 4096 + v_hi[db]);
#endif
      }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        bf16x8 vf;
        vf[0] = vlo[ks][db][0]; vf[1] = vlo[ks][db][1]; vf[2] = vlo[ks][db][2]; vf[3] = vlo[ks][db][3];
        vf[4] = vhi[ks][db][0]; vf[5] = vhi[ks][db][1]; vf[6] = vhi[ks][db][2]; vf[7] = vhi[ks][db][3];
        o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], o_acc[db], 0, 0, 0);
      }
    // pin the issue order (hipcc otherwise sinks every read to just before its MFMA and each MFMA
    // pays an LDS round trip): 8 transposed reads ahead, then 1 MFMA : 2 reads, then the last MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  };
  auto seg_qk = [&](int t) {   // S^T(t) = K(t) . Q^T
    const char* kb_lds = smem + (t & 1) * TILE_B;
    bf16x8 kf[2][8];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 8; ++s) kf[kb][s] = *reinterpret_cast<const bf16x8*>(kb_lds + kb * 8192 + k_addr[s]);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kb][s], qf[s], st[kb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // 6 K fragments ahead, then 1 MFMA : 1 read
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
  };
  // ---- vector segment: online softmax of st -> pf
  auto seg_softmax = [&](int t) {
#ifdef SF_ABL_NOSOFTMAX   // timing-only ablation build: convert S^T straight to bf16
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) pf[2 * kb + (r >> 3)][r & 7] = (bf16_t)st[kb][r];
    l_run += 1.f;
    return;
#endif
    if (tail && t == ntiles - 1) {
      asm volatile("" ::: "memory");   // keep this a real (wave-uniform) branch, not 32 selects per tile
      const int key0 = t * KT + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= p.Lk) st[kb][r] = -1e30f;
        }
    }
    float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(st[0][r], st[1][r]));
    {
      const unsigned mi = __float_as_uint(mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mi, mi, false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (__any(mx > m_run)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
    }
    const float mc = m_run * c;
    float lsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(st[kb][r] * c - mc);
        pf[2 * kb + (r >> 3)][r & 7] = (bf16_t)pv;   // k-step ks = 2 kb + (r>>3), element r & 7
        lsum += pv;
      }
    l_run += lsum;
  };

  // rows past Lk are never fetched (out of the descriptor's range): make sure they hold zeros and not
  // whatever bit patterns a previous kernel left in LDS (0 * NaN = NaN in the P.V product)
#pragma unroll
  for (int i = 0; i < ATT8_LDS / (ATT8_THREADS * 16); ++i)
    *reinterpret_cast<u32x4*>(smem + (i * ATT8_THREADS + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  dma_k(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // Global steps g = 0 .. 2 ntiles + 1, one barrier after each but the last.  In even steps g = 2 tt
  // every wave requests K(tt+1) and V(tt) by LDS-DMA: their buffers (those of K(tt-1), V(tt-2)) were
  // last read in step 2 tt - 1; the data is first read in step 2 tt + 2, so the DMA has two whole steps
  // to land and is waited for (vmcnt(0), then the barrier) only at the end of the odd step 2 tt + 1.
  // The halves run the same segments one step apart, written out per half so that every register's
  // live range is static (st: matrix -> vector of the same tile; pf: vector -> next matrix).
  auto step_barrier = [&](bool dma_must_land) {
#ifndef SF_ABL_EVENBAR
    if (!dma_must_land) return;
#endif
    if (dma_must_land) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto fetch = [&](int tt) {
#ifdef SF_ABL_NOSTAGE
    return;
#endif
    if (tt + 1 < ntiles) dma_k(tt + 1);
    if (tt < ntiles) dma_v(tt);
  };
  if (half == 0) {
    fetch(0);                         // g = 0
    seg_qk(0);
    step_barrier(false);
    seg_softmax(0);                   // g = 1
    step_barrier(true);
    for (int t = 1; t < ntiles; ++t) {
      fetch(t);                       // g = 2t
      seg_pv(t - 1);
      seg_qk(t);
      step_barrier(false);
      seg_softmax(t);                 // g = 2t + 1
      step_barrier(true);
    }
    seg_pv(ntiles - 1);   // g = 2 ntiles
    step_barrier(false);
  } else {
    fetch(0);                         // g = 0
    step_barrier(false);
    seg_qk(0);     // g = 1
    step_barrier(true);
    fetch(1);                         // g = 2
    seg_softmax(0);
    step_barrier(false);
    for (int t = 1; t < ntiles; ++t) {
      seg_pv(t - 1);   // g = 2t + 1
      seg_qk(t);
      step_barrier(true);
      fetch(t + 1);                   // g = 2t + 2
      seg_softmax(t);
      step_barrier(false);
    }
    seg_pv(ntiles - 1);   // g = 2 ntiles + 1
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  // The lane pair (l, l + 32) holds the two 8-byte halves of 16 contiguous output bytes (d = 8 g + 4 hh + 0..3).  With
  // 16-byte aligned output rows one v_permlane32_swap per dword regroups them: lane l stores the whole even 16-byte group,
  // lane l + 32 the odd one -- 8 x dwordx4 per lane instead of 16 x dwordx2 (the store tail of an attention epilogue is
  // issue-bound: cdna_hip_programming.md T21).  Validity is per query row, i.e. the same for both lanes of a pair.
  const bool wide = (((uintptr_t)obase | (uintptr_t)(p.o_stride * 2)) & 15) == 0;
  if (wide) {
    bf16_t* op = obase + (long)min(qrow, p.Lq - 1) * p.o_stride + 8 * hh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        unsigned wd[2][2];                      // [group 2 pr + g][dword]
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const bf16x2 pk = {(bf16_t)(o_acc[db][4 * (2 * pr + g) + 2 * d] * inv), (bf16_t)(o_acc[db][4 * (2 * pr + g) + 2 * d + 1] * inv)};
            wd[g][d] = __builtin_bit_cast(unsigned, pk);
          }
        const auto s0 = __builtin_amdgcn_permlane32_swap(wd[0][0], wd[1][0], false, false);   // {[A0.lo, B0.lo], [A0.hi, B0.hi]}
        const auto s1 = __builtin_amdgcn_permlane32_swap(wd[0][1], wd[1][1], false, false);
        const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
        if (qrow < p.Lq) *reinterpret_cast<u32x4*>(op + db * 32 + pr * 16) = v;
      }
  } else if (qrow < p.Lq) {
    bf16_t* op = obase + (long)qrow * p.o_stride + 4 * hh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        bf16x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (bf16_t)(o_acc[db][4 * rg + j] * inv);
        *reinterpret_cast<bf16x4*>(op + db * 32 + rg * 8) = w;
      }
  }
}



// ------------------------------------------------------------------------------------------
// 64 query rows per wave, one wave per SIMD, hand-scheduled: the C++ below only computes the per-lane
// addresses (same layouts and formulas as the kernels above) and hands them to the generated assembly
// body through LDS; tools/gen_attention_r64.py documents the register map and the pipeline.
constexpr int QT64 = 256;                      // 4 waves x 64 rows
constexpr int ATT64_LDS = 7 * TILE_B;          // K ring of 4 + V ring of 3 = 112 KiB

template <int KIND>   // 0: keys = the KV cache (self-attention), 1: short key sequences (cross-attention): distinct symbols for profiles
__global__ __launch_bounds__(256) void attention_r64_kernel(AttP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Workgroup -> (batch * head, query tile), XCD-aware (placement affects speed only): workgroups bid and bid + 8 share an
  // XCD and its L2.  Every query tile of a (batch, head) pair streams the same K / V slab, so a pair's tiles should sit
  // on ONE XCD: each XCD first takes f = floor(slots / q_tiles) whole pairs; the pairs left over fill the XCDs'
  // remaining slots in the order 0,4,1,5,2,6,3,7 -- the round-robin gives the first (nwg & 7) XCDs one slot more, so
  // this order pairs a larger remainder with a smaller one and a split pair spans two XCDs, not three.  At the
  // rollout's shape (12 heads x 19 tiles = 228 workgroups: 28.5 slots per XCD) 8 heads are read by one XCD and 4 by
  // two: 1.33x the K / V bytes instead of 1.47x with contiguous chunks; with two samples per launch (456 workgroups =
  // 57 slots = exactly 3 pairs per XCD) every slab is fetched once.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, jx = bid >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int Q = p.q_tiles, f = q8 / Q;
  int bh, qt;
  if (jx < f * Q) {
    bh = xcd * f + jx / Q;
    qt = jx - (jx / Q) * Q;
  } else {
    int g = jx - f * Q;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int px = (i >> 1) + 4 * (i & 1);                  // 0,4,1,5,2,6,3,7
      const bool before = ((xcd & 3) * 2 + (xcd >> 2)) > i;     // XCD px comes earlier in that order than this one
      if (before) g += q8 + (px < r8 ? 1 : 0) - f * Q;
    }
    bh = 8 * f + g / Q;
    qt = g - (g / Q) * Q;
  }
  const int b = bh / p.H, head = bh - b * p.H;
  const bf16_t* qbase = p.q + (long)b * p.q_bstride + head * HD;
  const bf16_t* kbase = p.k + (long)b * p.kv_bstride + head * HD;
  const bf16_t* vbase = p.v + (long)b * p.kv_bstride + head * HD;
  bf16_t* obase = p.o + (long)b * p.o_bstride + head * HD;

  const int r32 = lane & 31, hh = lane >> 5;
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  unsigned prm[SF_R64_N_PARAM];
  {
    const int x = hh ^ (((r32 & 3) << 2) | ((r32 >> 2) & 3));
#pragma unroll
    for (int s = 0; s < 8; ++s) prm[s] = lds_base + 256 * r32 + 16 * ((2 * s) ^ x);
    const int g16 = lane >> 4, i16 = lane & 15;
    const int tq = i16 >> 2, tp = i16 & 3;
    const int y = 2 * (g16 & 1) + (tp >> 1);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      prm[8 + db] = lds_base + 256 * (4 * hh + tq) + 16 * (4 * (db ^ tq) + (y ^ hh)) + 8 * (tp & 1);
      prm[12 + db] = lds_base + 256 * (8 + 4 * hh + tq) + 16 * (4 * (db ^ tq) + (y ^ (hh + 2))) + 8 * (tp & 1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // this wave's LDS-DMA pieces 4 wave + i: 4 rows x 256 B each, swizzle on the source
      const int row = (wave * 4 + i) * 4 + (lane >> 4);
      const int chunk = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
      // (minus the i KiB the instruction's immediate offset adds back: piece i's LDS address is M0 + i KiB, set once per group)
      prm[16 + i] = (unsigned)(((long)row * p.kv_stride + chunk * 8) * 2 - i * 1024);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qrow = qt * QT64 + wave * 64 + qb * 32 + r32;
      const unsigned long long qa = (unsigned long long)(qbase + (long)min(qrow, p.Lq - 1) * p.q_stride + 8 * hh);
      // (the epilogue regroups a lane pair's two 8-byte halves into 16-byte stores: lane l writes the even 16-byte group,
      // lane l + 32 the odd one)
      const unsigned long long oa = (unsigned long long)(obase + (long)min(qrow, p.Lq - 1) * p.o_stride + 8 * hh);
      prm[20 + 2 * qb] = (unsigned)qa; prm[21 + 2 * qb] = (unsigned)(qa >> 32);
      prm[24 + 2 * qb] = (unsigned)oa; prm[25 + 2 * qb] = (unsigned)(oa >> 32);
      prm[29 + qb] = qrow < p.Lq ? 1u : 0u;
    }
    prm[28] = 4 * hh;
  }
  unsigned* pl = reinterpret_cast<unsigned*>(smem);
#pragma unroll
  for (int j = 0; j < SF_R64_N_PARAM; ++j) pl[j * 256 + tid] = prm[j];

  const unsigned kv_bytes = (unsigned)(((long)(p.Lk - 1) * p.kv_stride + HD) * 2);
  auto make_srd = [&](const bf16_t* base) {
    const unsigned long long a64 = (unsigned long long)base;
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((unsigned)a64);
    d[1] = __builtin_amdgcn_readfirstlane((unsigned)(a64 >> 32) & 0xFFFFu);
    d[2] = __builtin_amdgcn_readfirstlane(kv_bytes);
    d[3] = 0x00020000u;
    return d;
  };
  const u32x4 k_srd = make_srd(kbase), v_srd = make_srd(vbase);
  const unsigned tile_bytes = __builtin_amdgcn_readfirstlane((unsigned)((long)KT * p.kv_stride * 2));
  const int ntiles = __builtin_amdgcn_readfirstlane((p.Lk + KT - 1) / KT);
  const int lk = __builtin_amdgcn_readfirstlane(p.Lk);
  const unsigned cbits = __builtin_amdgcn_readfirstlane(__float_as_uint(p.scale_log2));
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 4096u);
  const unsigned lds_b = __builtin_amdgcn_readfirstlane(lds_base);
  const unsigned tid4 = lds_base + 4u * (unsigned)tid;
  asm volatile(SF_R64_ASM_BODY
               :
               : "s"(k_srd), "s"(v_srd), "s"(tile_bytes), "s"(ntiles), "s"(lk), "s"(cbits), "s"(lds_wave), "v"(tid4), "s"(lds_b)
               : SF_R64_CLOBBERS);
}



}  // namespace

extern "C" int sf_attention_ex(const void* q, const void* k, const void* v, void* out, int B, int H, int Lq,
                               int Lk, int64_t q_stride, int64_t q_bstride, int64_t kv_stride,
                               int64_t kv_bstride, int64_t o_stride, int64_t o_bstride, int structure, void* stream) {
  SF_CHECK(q && k && v && out, "sf_attention: null tensor");
  SF_CHECK(B > 0 && H > 0 && Lq > 0 && Lk > 0, "sf_attention: empty problem B=%d H=%d Lq=%d Lk=%d", B, H, Lq, Lk);
  SF_CHECK(q_stride % 8 == 0 && kv_stride % 8 == 0 && o_stride % 4 == 0, "sf_attention: strides must keep 16-byte row alignment");
  SF_CHECK(q_bstride % 8 == 0 && kv_bstride % 8 == 0 && o_bstride % 4 == 0, "sf_attention: batch strides must keep alignment");
  SF_CHECK(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)out % 8 == 0),
           "sf_attention: misaligned tensor");
  SF_CHECK(((long)(Lk - 1) * kv_stride + 128) * 2 < (1L << 31), "sf_attention: one (batch, head) K/V slab must span < 2 GiB");
  SF_CHECK(structure >= SF_ATTN_AUTO && structure <= SF_ATTN_W4, "sf_attention: unknown structure %d", structure);
  AttP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)out;
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk;
  p.q_stride = q_stride; p.q_bstride = q_bstride; p.kv_stride = kv_stride; p.kv_bstride = kv_bstride;
  p.o_stride = o_stride; p.o_bstride = o_bstride;
  p.q_tiles = (Lq + QT - 1) / QT;
  p.scale_log2 = 1.4426950408889634f / sqrtf((float)HD);
  // Structure (SF_ATTN_AUTO): long key sequences that fill the chip run the hand-scheduled 64-rows-per-wave
  // kernel; short ones that fill it (cross-attention: 8 key tiles) the 8-wave anti-phase kernel (256 query rows
  // per workgroup); small problems the 4-wave one (128 rows, two workgroups per CU).  Every structure is correct
  // for every shape; the explicit values exist for tests and A/B timing.
  const long nwg64 = (long)((Lq + QT64 - 1) / QT64) * H * B;
  const long nwg8 = (long)((Lq + QT8 - 1) / QT8) * H * B;
  // (the 64-row kernel stores 16 bytes per lane: it needs 16-byte aligned output rows)
  const bool out16 = ((uintptr_t)out % 16 == 0) && o_stride % 8 == 0 && o_bstride % 8 == 0;
  // (round 3: with its shorter prologue / epilogue the 64-row kernel is ahead of the 8-wave one from 512 keys on --
  // cross-attention, Lk = 512: 21.2 vs 22.1 us at one prompt, 41.7 vs 43.1 at two; Lk = 256: 15.8 vs 14.8)
  if (structure == SF_ATTN_AUTO)
    structure = (nwg64 >= 192 && Lk >= 512 && out16) ? SF_ATTN_R64 : nwg8 >= 192 ? SF_ATTN_W8 : SF_ATTN_W4;
  if (structure == SF_ATTN_R64) {
    SF_CHECK(out16, "sf_attention: the r64 structure needs 16-byte aligned output rows (out %% 16, o_stride %% 8, o_bstride %% 8)");
    p.q_tiles = (Lq + QT64 - 1) / QT64;
    static bool attr = false;   // one-time registration of the kernels' LDS size (idempotent; no other state is kept)
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_r64_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, ATT64_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_r64_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, ATT64_LDS);
      attr = true;
    }
    if (Lk > 1024)
      hipLaunchKernelGGL(attention_r64_kernel<0>, dim3((unsigned)nwg64), dim3(256), ATT64_LDS, (hipStream_t)stream, p);
    else
      hipLaunchKernelGGL(attention_r64_kernel<1>, dim3((unsigned)nwg64), dim3(256), ATT64_LDS, (hipStream_t)stream, p);
    SF_HIP_LAUNCH_CHECK("sf_attention");
    return 0;
  }
  if (structure == SF_ATTN_W8) {
    p.q_tiles = (Lq + QT8 - 1) / QT8;
    if (Lk > 1024)
      hipLaunchKernelGGL(attention_w8_kernel<0>, dim3((unsigned)nwg8), dim3(ATT8_THREADS), ATT8_LDS, (hipStream_t)stream, p);
    else
      hipLaunchKernelGGL(attention_w8_kernel<1>, dim3((unsigned)nwg8), dim3(ATT8_THREADS), ATT8_LDS, (hipStream_t)stream, p);
    SF_HIP_LAUNCH_CHECK("sf_attention");
    return 0;
  }
  const long nwg = (long)p.q_tiles * H * B;
  SF_CHECK(nwg < (1L << 30), "sf_attention: grid too large");
  if (Lk > 1024)
    hipLaunchKernelGGL(attention_kernel<0>, dim3((unsigned)nwg), dim3(ATT_THREADS), ATT_LDS, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(attention_kernel<1>, dim3((unsigned)nwg), dim3(ATT_THREADS), ATT_LDS, (hipStream_t)stream, p);
  SF_HIP_LAUNCH_CHECK("sf_attention");
  return 0;
}

extern "C" int sf_attention(const void* q, const void* k, const void* v, void* out, int B, int H, int Lq,
                            int Lk, int64_t q_stride, int64_t q_bstride, int64_t kv_stride,
                            int64_t kv_bstride, int64_t o_stride, int64_t o_bstride, void* stream) {
  return sf_attention_ex(q, k, v, out, B, H, Lq, Lk, q_stride, q_bstride, kv_stride, kv_bstride, o_stride, o_bstride,
                         SF_ATTN_AUTO, stream);
}
