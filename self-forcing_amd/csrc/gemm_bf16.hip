// bf16 GEMM with fused epilogues for gfx950 (MI355X):  out[M,N] = epi(A[M,K] @ W[N,K]^T + bias).
//
// Replaces nn.Linear (+ the elementwise ops after it) on the reference's hot path
// (wan/modules/causal_model.py:112-114, :240, :277-279, :320, :331, :366; model.py:172-193).
//
// Structure: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave,
// 4x4 MFMA 16x16x32 bf16 tiles), BK = 64, A and W tiles staged HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4) into two buffers so that tile k+1 is in flight while tile k feeds the
// matrix cores.  Both operands are K-contiguous (activations [M,K], nn.Linear weights [N,K]), so
// every MFMA fragment is one 16-byte ds_read_b128.  LDS rows are 128 B; the 16-byte chunk c of row
// r is stored at chunk c ^ ((r>>1)&7) -- applied on the global SOURCE address (the LDS-DMA
// destination is lane-linear) and again on the read -- which makes the ds_read_b128 fragment reads
// bank-conflict free.  The MFMA is issued as D = W_frag x X_frag (operands swapped) so that each
// lane ends up with 4 CONSECUTIVE output columns of one row: 8-byte bf16x4 stores/loads in the
// epilogue instead of 2-byte scattered ones.
#include <type_traits>
#include "sf_common.h"
#include "../../include/sf_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;       // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A tile + W tile
constexpr int GEMM_LDS = 2 * STAGE_BYTES;     // double buffered: 64 KiB

struct GemmP {
  const bf16_t* a;
  const bf16_t* w;
  const bf16_t* bias;
  bf16_t* out;
  const bf16_t* resid;
  const bf16_t* gate_mod;
  const bf16_t* gate_e0;
  long gate_group_stride;
  int rows_per_group;
  int M, N, K, lda, ldw, ldo, ldr;
  int tiles_m, tiles_n;
  long a_bs, w_bs, o_bs, r_bs;   // batch strides (elements; o_bs in floats for SF_EPI_F32); batch index = blockIdx.y
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// Epilogue arithmetic on the 4 consecutive output columns a lane owns of row m: bias, GELU, gate, residual.
template <int EPI>
__device__ __forceinline__ bf16x4 epi_apply(const GemmP& p, const f32x4& a4, int m, int n, const bf16_t* e0row) {
  float y[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = a4[j];
  if (p.bias) {
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(p.bias + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] += (float)b[j];
  }
  if (EPI == SF_EPI_BIAS_GELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(y[j]);
  }
  if (EPI == SF_EPI_BIAS_GATE_RESID) {
    const bf16x4 gm = *reinterpret_cast<const bf16x4*>(p.gate_mod + n);
    const bf16x4 ge = *reinterpret_cast<const bf16x4*>(e0row + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] *= (float)(bf16_t)((float)gm[j] + (float)ge[j]);
  }
  if (EPI == SF_EPI_BIAS_RESID || EPI == SF_EPI_BIAS_GATE_RESID) {
    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(p.resid + (long)m * p.ldr + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] += (float)rv[j];
  }
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16_t)y[j];
  return o;
}

// Direct epilogue: the lane holds out[m][n .. n+3] for each (mt, nt) of its wave's sub-tile
// (m = mrow + 16 mt, n = ncol + 16 nt) and stores them as they are (8-byte pieces).
template <int EPI, int MT>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[MT][4], int mrow, int ncol) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mrow + mt * 16;
    if (m >= p.M) continue;
    const bf16_t* e0row = nullptr;
    if (EPI == SF_EPI_BIAS_GATE_RESID) e0row = p.gate_e0 + (long)(m / p.rows_per_group) * p.gate_group_stride;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = ncol + nt * 16;
      if (n >= p.N) continue;
      if (EPI == SF_EPI_F32) {   // raw accumulators (attention logits): `out` is float*, ldo in floats
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (long)m * p.ldo + n) = acc[mt][nt];
        continue;
      }
      *reinterpret_cast<bf16x4*>(p.out + (long)m * p.ldo + n) = epi_apply<EPI>(p, acc[mt][nt], m, n, e0row);
    }
  }
}

// Epilogue arithmetic on operands that are already in registers (see gemm_epilogue_lds).
template <int EPI>
__device__ __forceinline__ bf16x4 epi_math(const f32x4& a4, const bf16x4& b, const bf16x4& gm, const bf16x4& ge, const bf16x4& rv) {
  float y[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = a4[j] + (float)b[j];
  if (EPI == SF_EPI_BIAS_GELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = gelu_tanh_f(y[j]);
  }
  if (EPI == SF_EPI_BIAS_GATE_RESID) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] *= (float)(bf16_t)((float)gm[j] + (float)ge[j]);
  }
  if (EPI == SF_EPI_BIAS_RESID || EPI == SF_EPI_BIAS_GATE_RESID) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] += (float)rv[j];
  }
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16_t)y[j];
  return o;
}

// Epilogue through LDS: the wave's (16 MT) x 64 sub-tile is written to its own LDS region as bf16 rows of
// 128 B (16-byte chunk c of row r at c ^ (r & 7): 2-way conflicts at most on the 8-byte writes, none on the
// reads) and read back row-wise, so that every global store instruction covers 8 whole 128-byte rows.
// The direct form writes 32-byte pieces of 16 different rows per instruction: for the 84 MB ffn.0 output
// that is a quarter of HBM's write efficiency and cost 12-20 us per 256 x 256 tile.
// All global operands of the epilogue (bias and gate vectors once per column block, residual and per-group gate
// rows for four row blocks at a time) are REQUESTED FIRST and consumed afterwards: written piece by piece
// (load bias, wait, load residual, wait, compute, 16-32 times per lane) the epilogue was a chain of dependent memory
// round trips -- 8 us of a 36 us o-projection, 15 us per 256 x 256 tile.
template <int EPI, int MT>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmP& p, f32x4 (&acc)[MT][4], int m_base, int n_base, char* wbuf, int lane) {
  const int r16 = lane & 15, cg = lane >> 4;
  constexpr bool GATE = EPI == SF_EPI_BIAS_GATE_RESID;
  constexpr bool RESID = EPI == SF_EPI_BIAS_RESID || EPI == SF_EPI_BIAS_GATE_RESID;
  const bf16x4 zero4 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  int ncol[4];
  bf16x4 bias_v[4], gm_v[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    ncol[nt] = min(n_base + nt * 16 + cg * 4, p.N - 4);
    bias_v[nt] = zero4;
    gm_v[nt] = zero4;
  }
  if (p.bias) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias_v[nt] = *reinterpret_cast<const bf16x4*>(p.bias + ncol[nt]);
  }
  if (GATE) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) gm_v[nt] = *reinterpret_cast<const bf16x4*>(p.gate_mod + ncol[nt]);
  }
  constexpr int MB = 4;                                      // row blocks whose operands are in flight together
#pragma unroll
  for (int m0 = 0; m0 < MT; m0 += MB) {
    bf16x4 rv[MB][4], ge[MB][4];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      if (m0 + i >= MT) break;
      const int m = min(m_base + (m0 + i) * 16 + r16, p.M - 1);
      const bf16_t* e0row = GATE ? p.gate_e0 + (long)(m / p.rows_per_group) * p.gate_group_stride : nullptr;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        rv[i][nt] = RESID ? *reinterpret_cast<const bf16x4*>(p.resid + (long)m * p.ldr + ncol[nt]) : zero4;
        ge[i][nt] = GATE ? *reinterpret_cast<const bf16x4*>(e0row + ncol[nt]) : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      if (m0 + i >= MT) break;
      const int row = (m0 + i) * 16 + r16;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int col = nt * 16 + cg * 4;                     // 0..63 within the wave's columns
        const bf16x4 o = epi_math<EPI>(acc[m0 + i][nt], bias_v[nt], gm_v[nt], ge[i][nt], rv[i][nt]);
        const int chunk = (col >> 3) ^ (row & 7);
        *reinterpret_cast<bf16x4*>(wbuf + row * 128 + (chunk << 4) + (col & 4) * 2) = o;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes (the region is private to the wave)
  const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
  for (int i = 0; i < 2 * MT; ++i) {
    const int row = i * 8 + rr;
    const int m = m_base + row, n = n_base + ch * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(wbuf + row * 128 + ((ch ^ (row & 7)) << 4));
    if (m < p.M && n < p.N) *reinterpret_cast<bf16x8*>(p.out + (long)m * p.ldo + n) = v;
  }
}

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch), so give
  // each XCD a contiguous range of tiles -> neighbouring tiles (same A rows / same W panel)
  // share one L2.  Placement affects speed only.
  if (blockIdx.y) {   // batched launch (per-head products of the text encoder): same tile grid per batch entry
    const long bz = blockIdx.y;
    p.a += bz * p.a_bs; p.w += bz * p.w_bs;
    if (EPI == SF_EPI_F32) p.out = reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(p.out) + bz * p.o_bs); else p.out += bz * p.o_bs;
    if (p.resid) p.resid += bz * p.r_bs;
  }
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  // Grouped order on top of the XCD remap: consecutive workgroups (which run concurrently on one
  // XCD and advance through K in step) cover GROUP_M row tiles x a run of column tiles, so both the
  // A panel and the W tiles they stream are shared in that XCD's L2.  Row-major tile order made
  // every sweep of a row re-read the whole weight matrix: 1.05 GB fetched for the 1.3B ffn.0 GEMM
  // (FETCH_SIZE x 2) against 126 MB of operands + output.
  constexpr int GROUP_M = 8;
  const int width = GROUP_M * p.tiles_n;
  const int group = wg / width, first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * width;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging addresses: wave w issues 4 A pieces + 4 W pieces of 1 KiB (8 rows x 128 B) each
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int am = min(m0 + r, p.M - 1);
    const int wn = min(n0 + r, p.N - 1);
    a_src[i] = p.a + (long)am * p.lda + c * 8;
    w_src[i] = p.w + (long)wn * p.ldw + c * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES + wave * 4096;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + k0, base + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w_src[i] + k0, base + TILE_BYTES + i * 1024);
  };

  // ---- fragment read addresses
  const int wr = wave >> 1, wc = wave & 1;
  const int i16 = lane & 15, kq = lane >> 4;
  const int swz = (i16 >> 1) & 7;
  const int x_row_off = (wr * 64 + i16) * 128;                // + t*2048
  const int w_row_off = TILE_BYTES + (wc * 64 + i16) * 128;   // + t*2048
  const int coff0 = ((0 + kq) ^ swz) << 4;
  const int coff1 = ((4 + kq) ^ swz) << 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): LDS-DMA landed
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* buf = smem + cur * STAGE_BYTES;
    // The k-step as eight pinned slices of {2 MFMAs of the first 32-deep sub-step, one fragment read of the
    // second sub-step, ONE LDS-DMA request of the next tile}, then the second sub-step's 16 MFMAs.  Left to the
    // scheduler the 8 requests go out as a burst, which blocks the wave for ~220 ns per k-step (the TCP->LDS
    // path moves 64 B/clk/CU); one request per two MFMAs costs ~20 ns (tools/probes/dma_probe.hip).  The last
    // k-step re-requests its own tile into the idle buffer so that the loop body stays branch-free.
    const int kn = min(kt + 1, nk - 1) * BK;
    char* sbase = smem + (cur ^ 1) * STAGE_BYTES + wave * 4096;
    bf16x8 xf0[4], wf0[4], xf1[4], wf1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      xf0[t] = *reinterpret_cast<const bf16x8*>(buf + x_row_off + t * 2048 + coff0);
      wf0[t] = *reinterpret_cast<const bf16x8*>(buf + w_row_off + t * 2048 + coff0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int mt = i >> 1, nt0 = (i & 1) * 2;
      acc[mt][nt0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt0], xf0[mt], acc[mt][nt0], 0, 0, 0);
      acc[mt][nt0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt0 + 1], xf0[mt], acc[mt][nt0 + 1], 0, 0, 0);
      if (i < 4) {
        xf1[i] = *reinterpret_cast<const bf16x8*>(buf + x_row_off + i * 2048 + coff1);
        glds16(a_src[i] + kn, sbase + i * 1024);
      } else {
        wf1[i - 4] = *reinterpret_cast<const bf16x8*>(buf + w_row_off + (i - 4) * 2048 + coff1);
        glds16(w_src[i - 4] + kn, sbase + TILE_BYTES + (i - 4) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], xf1[mt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);   // keep the wait + barrier BEHIND the MFMAs: the requests need the time to land
    __syncthreads();  // drains the in-flight LDS-DMA (vmcnt(0)) and orders the buffer swap
  }

  // ---- epilogue: lane holds out[m][n .. n+3] for (mt, nt)
  const int mrow = m0 + wr * 64 + (lane & 15);
  const int ncol = n0 + wc * 64 + (lane >> 4) * 4;
  if (EPI != SF_EPI_F32 && (p.N & 7) == 0 && (p.ldo & 7) == 0)   // (the k-loop's last __syncthreads has released the stages)
    gemm_epilogue_lds<EPI, 4>(p, acc, m0 + wr * 64, n0 + wc * 64, smem + wave * 8192, lane);
  else
    gemm_epilogue<EPI, 4>(p, acc, mrow, ncol);
}


// ------------------------------------------------------------------------------------------
// Ping-pong structure ("pp"): (32 MT) x 256 output tile per 512-thread workgroup, 8 waves as 2 (M) x 4 (N), one
// workgroup per CU.  The two waves that share a SIMD (wave w and w + 4: the two M halves) run ONE BARRIER INTERVAL
// APART: while one issues its MFMA cluster (a quarter of its sub-tile x one 64-deep k-tile, from registers) the other
// reads its next fragments from LDS and requests LDS-DMA pieces -- the matrix pipe of every SIMD always has a cluster to
// run, and neither wave's LDS / address work sits in front of its own MFMAs.  The kernels above run both waves of a
// SIMD in lockstep (MFMAs together, loads together) and drain the LDS-DMA (vmcnt(0)) once per k-tile; here the
// requests stay in flight across barriers and are waited for ONCE per k-tile with a counted vmcnt that leaves the
// youngest two half-tiles outstanding (raw s_barrier: __syncthreads() would drain them).
//
// Per k-tile t (4 phases j = 0..3; buffer t & 1; A fragments of the current row half + B fragments of all four
// column tiles live in registers):
//     phase   fragments read from LDS (tile t)        cluster                LDS-DMA requested
//     j = 0   A rows q0 (MT/2 tiles), B cols 0,1      (q0, cols 0,1)         A half 0 of tile t+1
//     j = 1   B cols 2,3                              (q0, cols 2,3)         A half 1 of tile t+1
//     j = 2   A rows q1                               (q1, cols 2,3)         B half 0 of tile t+2
//     j = 3   --                                      (q1, cols 0,1)         B half 1 of tile t+2; vmcnt(4)
// Write-after-read: B of tile t is in registers after j = 1, A half h after j = 2, and every read is retired
// (lgkmcnt(0)) before the barrier that ends its interval, so the regions are free when the requests above are issued
// (A(t+1) goes to the OTHER buffer, last read during tile t-1).  Read-after-write: tile t+1's first reads come after
// the barrier behind the vmcnt(4) of phase (t, 3), executed by every wave: everything but B(t+2) has landed.
//
// Slot-by-slot (global barrier intervals I; group 0's load segment of (t, j) is I = 8t + 2j, its cluster 8t + 2j + 1,
// group 1 runs one interval later; group g reads only A half g, column group wc reads only its 64 rows of B):
//   A half 0 of buffer (t+1)&1: last read by group 0 at I = 8t - 4 (tile t-1, phase 2), rewritten by requests issued at
//     I = 8t (group 0) / 8t + 1 (group 1); first read again at I = 8t + 8, behind every wave's vmcnt(4) (I = 8t + 6 / 8t + 7)
//     and the barrier that ends I = 8t + 7.  A half 1: the same, one phase later (read until 8t - 3, rewritten from 8t + 2).
//   B halves of buffer t&1: last read at I = 8t + 3 (group 1, phase 1), rewritten from I = 8t + 4 / 8t + 6; first read at
//     I = 8t + 16, behind the vmcnt(4) of tile t + 1 (whose four youngest requests are B(t+3)).
//   Epilogue scratch (wave w: [w, w + 1) x MT x 2 KiB of the stages): entered only after the last tile's vmcnt(0)
//     (executed by every wave before a barrier that precedes every epilogue) -- no request is in flight, none is issued
//     afterwards, every fragment read was retired >= 3 intervals earlier.
//
// THE RACE OF THE PERSISTENT EXPERIMENT (git ba94dce, removed in 77df59c; analysed in round 3 from the diff).  That
// kernel continued the pipeline across OUTPUT tiles and ran each tile's epilogue without a workgroup barrier, group 0's
// four waves transposing through 8 KiB of scratch each inside buffer 1's A region (dead between the last k-tile's reads
// and the next output tile's A(1) requests).  The argument written there -- "A(1) is requested by each wave after its own
// epilogue, and by the other group after theirs" -- covers a wave's OWN scratch only: an LDS-DMA piece of A half 0 is
// 1 KiB at offset w x 1 KiB (+ 8 KiB) for EVERY wave w, i.e. the four waves' first requests of the next tile (issued in
// the same barrier interval as the epilogue, right behind their own, with no barrier in between) land in [0, 4 K) and
// [8 K, 12 K): the scratch of waves 0 and 1 -- so wave 2 or 3 (or wave 1 into wave 0's, wave 0 into wave 1's) can
// overwrite rows a sibling is still transposing (an epilogue's length varies with its residual / gate loads).  Group 1's requests were safe (they follow the
// barrier group 0 reaches only after all four epilogues); the intra-group order was the missing wait.  It showed on
// the one test shape with several tiles per workgroup AND long epilogues (10800 x 5120 x 5120: 860 tiles on 256
// workgroups), rarely.  A fix would have been a group-local rendezvous (LDS counter of the four waves) between the
// epilogues and the first A(1) request, or per-wave scratch outside every LDS-DMA target (4 KiB x 8 in the 32 KiB above
// the stages, four passes per sub-tile).  The kernels shipped here have no such window: their scratch is touched only
// after the last request has landed everywhere and nothing is requested after it (the third bullet above).
constexpr int PP_THREADS = 512;

template <int MT>   // M tiles of 16 rows per wave: 8 (256-row workgroup tile), 7 (224), 6 (192)
struct PPCfg {
  static constexpr int BM = 32 * MT;
  static constexpr int A_BYTES = BM * 128;          // one k-tile of A: BM rows x 64 bf16
  static constexpr int B_BYTES = 256 * 128;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int LDS = 2 * STAGE;             // 128 KiB (MT = 8)
  static constexpr int PA = (2 * MT + 7) / 8;       // LDS-DMA pieces (8 rows x 128 B) per wave per A half-tile (2 MT pieces over 8 waves)
};

template <int EPI, int MT>
__global__ __launch_bounds__(PP_THREADS) void gemm_pp_kernel(GemmP p) {
  using Cfg = PPCfg<MT>;
  constexpr int BMp = Cfg::BM, PA = Cfg::PA;
  constexpr int HQA = (MT + 1) / 2, HQB = MT / 2;    // row tiles of the wave's two "quarters" (4 + 3 at MT = 7)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;          // grp: M half of the tile = which of the two staggered wave groups

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GROUP_M = MT >= 6 ? 4 : 8;
  const int width = GROUP_M * p.tiles_n;
  const int group = wg / width, first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * width;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int m0 = tm * BMp, n0 = tn * 256;

  // ---- LDS-DMA source addresses: a piece = 8 rows x 128 B; half-tile h of A = rows [h BM/2, (h+1) BM/2), of B =
  // rows (= output columns) [128 h, 128 h + 128); wave w requests pieces w (+ 8 i) of every half-tile.
  const bf16_t* a_src[2][PA];
  const bf16_t* w_src[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int r = h * (BMp / 2) + (wave + 8 * i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      a_src[h][i] = p.a + (long)min(m0 + r, p.M - 1) * p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = h * 128 + (wave + 8 * i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      w_src[h][i] = p.w + (long)min(n0 + r, p.N - 1) * p.ldw + c * 8;
    }
  }
  auto dma_a = [&](int h, int kt) {
    char* base = smem + (kt & 1) * Cfg::STAGE + h * (Cfg::A_BYTES / 2) + wave * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      if (wave + 8 * i < 2 * MT) glds16(a_src[h][i] + kt * BK, base + i * 8192);   // (wave-uniform; the youngest requests of a k-tile are B's either way)
  };
  auto dma_b = [&](int h, int kt) {
    char* base = smem + (kt & 1) * Cfg::STAGE + Cfg::A_BYTES + h * (Cfg::B_BYTES / 2) + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(w_src[h][i] + kt * BK, base + i * 8192);
  };

  // ---- fragment read addresses (16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7))
  const int i16 = lane & 15, kq = lane >> 4;
  const int swz = (i16 >> 1) & 7;
  const int x_row_off = (grp * (BMp / 2) + i16) * 128;               // + 2048 per M tile
  const int w_row_off = Cfg::A_BYTES + (wc * 64 + i16) * 128;        // + 2048 per N tile
  const int coff[2] = {((0 + kq) ^ swz) << 4, ((4 + kq) ^ swz) << 4};

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[HQA][2], bfr[4][2];

#ifdef SF_STAMP   // diagnostic build (tools/probes/gemm_stamp.py): where a tile's time goes; stamps go to a buffer of their own
  unsigned long long* stamp = (EPI == SF_EPI_BIAS && p.gate_e0) ? reinterpret_cast<unsigned long long*>(const_cast<bf16_t*>(p.gate_e0)) + ((long)bid * 2 + grp) * 8 : nullptr;
  if (stamp && (tid & 255) == 0) { stamp[0] = __builtin_amdgcn_s_memtime(); stamp[4] = __builtin_amdgcn_s_memrealtime(); }
#endif
  const int nk = p.K / BK;
  // prologue: A(0), B(0), then B(1) which may still be in flight when tile 0 starts (same count as in the loop)
  dma_a(0, 0); dma_a(1, 0); dma_b(0, 0); dma_b(1, 0);
  if (nk > 1) { dma_b(0, 1); dma_b(1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef SF_STAMP
  if (stamp && (tid & 255) == 0) stamp[1] = __builtin_amdgcn_s_memtime();
#endif
  if (grp == 1) __builtin_amdgcn_s_barrier();       // the second wave group runs one interval behind the first

  auto end_load_segment = [&]() {
    __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): this wave's fragment reads are retired
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto cluster = [&](auto qc, int c0) {             // one quarter: HQA / HQB row tiles x 2 column tiles x 2 sub-steps of 32
    constexpr int q = decltype(qc)::value, NQ = q == 0 ? HQA : HQB;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < NQ; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[q * HQA + mt][c0 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[c0 + nt][ks], af[mt][ks], acc[q * HQA + mt][c0 + nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto read_a = [&](const char* buf, auto qc) {
    constexpr int q = decltype(qc)::value, NQ = q == 0 ? HQA : HQB;
#pragma unroll
    for (int mt = 0; mt < NQ; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        af[mt][ks] = *reinterpret_cast<const bf16x8*>(buf + x_row_off + (q * HQA + mt) * 2048 + coff[ks]);
  };
  auto read_b = [&](const char* buf, int c0) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        bfr[c0 + nt][ks] = *reinterpret_cast<const bf16x8*>(buf + w_row_off + (c0 + nt) * 2048 + coff[ks]);
  };

  constexpr std::integral_constant<int, 0> Q0{};
  constexpr std::integral_constant<int, 1> Q1{};
  for (int kt = 0; kt < nk; ++kt) {
    const char* buf = smem + (kt & 1) * Cfg::STAGE;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    // phase 0
    read_a(buf, Q0);
    read_b(buf, 0);
    if (more1) dma_a(0, kt + 1);
    end_load_segment();
    cluster(Q0, 0);
    // phase 1
    read_b(buf, 2);
    if (more1) dma_a(1, kt + 1);
    end_load_segment();
    cluster(Q0, 2);
    // phase 2
    read_a(buf, Q1);
    if (more2) dma_b(0, kt + 2);
    end_load_segment();
    cluster(Q1, 2);
    // phase 3: everything but the two half-tiles of B(kt + 2) must have landed before the next tile's first reads
    if (more2) {
      dma_b(1, kt + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    end_load_segment();
    cluster(Q1, 0);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();       // pairs with the extra barrier of the second group: all clusters done
#ifdef SF_STAMP
  if (stamp && (tid & 255) == 0) stamp[2] = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue (no LDS-DMA is in flight, every fragment read is retired: the stages are free)
  const int m_base = m0 + grp * (BMp / 2), n_base = n0 + wc * 64;
  if ((p.N & 7) == 0 && (p.ldo & 7) == 0)
    gemm_epilogue_lds<EPI, MT>(p, acc, m_base, n_base, smem + wave * (MT * 2048), lane);
  else
    gemm_epilogue<EPI, MT>(p, acc, m_base + (lane & 15), n_base + (lane >> 4) * 4);
#ifdef SF_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stamp && (tid & 255) == 0) { stamp[3] = __builtin_amdgcn_s_memtime(); stamp[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// ------------------------------------------------------------------------------------------
// 128 x 256 tile in the same ping-pong structure, TWO phases per k-tile.  With 64 x 64 per wave a quarter-tile cluster
// is only 8 MFMAs (128 cycles): shorter than the partner's load segment (fragment reads -> lgkmcnt(0) -> barrier is
// ~200 cycles of latency), so the 4-phase schedule above ran latency-bound at this tile (measured: 62 % of the
// matrix rate where the 256-row tile reaches 98 %).  Here a cluster is HALF the sub-tile (all 4 row tiles x 2 column
// tiles x 2 sub-steps = 16 MFMAs = 256 cycles), and A and B each get a ring of THREE k-tiles (48 + 96 = 144 KiB), so
// every LDS-DMA request is issued two k-tiles ahead of its first read:
//     phase 0: read A (8 fragments) + B cols 0,1 (4); request A(t+2);              cluster (rows, cols 0,1)
//     phase 1: read B cols 2,3 (4); request B(t+2); vmcnt(6) = all but tile t+2;   cluster (rows, cols 2,3)
// Ring slot (t+2) % 3 was last read in tile t-1 (A: its phase 0, B: its phase 1), retired before that interval's barrier.
constexpr int PP2_A = 128 * 128, PP2_B = 256 * 128;      // bytes per k-tile
constexpr int PP2_LDS = 3 * (PP2_A + PP2_B);             // 144 KiB

template <int EPI>
__global__ __launch_bounds__(PP_THREADS) void gemm_pp2_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  constexpr int GROUP_M = 8;
  const int width = GROUP_M * p.tiles_n;
  const int group = wg / width, first_m = group * GROUP_M;
  const int gsz = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = wg - group * width;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;
  const int m0 = tm * 128, n0 = tn * 256;

  const bf16_t* a_src[2];
  const bf16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave + 8 * i) * 8 + (lane >> 3);
    a_src[i] = p.a + (long)min(m0 + r, p.M - 1) * p.lda + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 8 * i) * 8 + (lane >> 3);
    w_src[i] = p.w + (long)min(n0 + r, p.N - 1) * p.ldw + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  }
  auto dma_a = [&](int slot, int kt) {
    char* base = smem + slot * PP2_A + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(a_src[i] + kt * BK, base + i * 8192);
  };
  auto dma_b = [&](int slot, int kt) {
    char* base = smem + 3 * PP2_A + slot * PP2_B + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w_src[i] + kt * BK, base + i * 8192);
  };

  const int i16 = lane & 15, kq = lane >> 4;
  const int swz = (i16 >> 1) & 7;
  const int x_row_off = (grp * 64 + i16) * 128;
  const int w_row_off = 3 * PP2_A + (wc * 64 + i16) * 128;
  const int coff[2] = {((0 + kq) ^ swz) << 4, ((4 + kq) ^ swz) << 4};

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bfr[4][2];

  const int nk = p.K / BK;
  dma_a(0, 0); dma_b(0, 0);
  if (nk > 1) { dma_a(1, 1); dma_b(1, 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();

  auto end_load_segment = [&]() {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto cluster = [&](int c0) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][c0 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[c0 + nt][ks], af[mt][ks], acc[mt][c0 + nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  int slot = 0;                       // kt % 3
  for (int kt = 0; kt < nk; ++kt) {
    const char* abuf = smem + slot * PP2_A;
    const char* bbuf = smem + slot * PP2_B;
    const int nslot = slot == 0 ? 2 : slot - 1;    // (kt + 2) % 3
    const bool more2 = kt + 2 < nk;
    // phase 0
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[mt][ks] = *reinterpret_cast<const bf16x8*>(abuf + x_row_off + mt * 2048 + coff[ks]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[nt][ks] = *reinterpret_cast<const bf16x8*>(bbuf + w_row_off + nt * 2048 + coff[ks]);
    if (more2) dma_a(nslot, kt + 2);
    end_load_segment();
    cluster(0);
    // phase 1
#pragma unroll
    for (int nt = 2; nt < 4; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[nt][ks] = *reinterpret_cast<const bf16x8*>(bbuf + w_row_off + nt * 2048 + coff[ks]);
    if (more2) {
      dma_b(nslot, kt + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // everything but tile kt + 2 has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    end_load_segment();
    cluster(2);
    slot = slot == 2 ? 0 : slot + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();

  const int m_base = m0 + grp * 64, n_base = n0 + wc * 64;
  if ((p.N & 7) == 0 && (p.ldo & 7) == 0)
    gemm_epilogue_lds<EPI, 4>(p, acc, m_base, n_base, smem + wave * 8192, lane);
  else
    gemm_epilogue<EPI, 4>(p, acc, m_base + (lane & 15), n_base + (lane >> 4) * 4);
}

template <int EPI>
int launch_pp2(GemmP& p, hipStream_t s) {
  static bool attr = false;   // one-time registration of the kernel's LDS size (idempotent)
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp2_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, PP2_LDS);
    attr = true;
  }
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + 255) / 256;
  hipLaunchKernelGGL((gemm_pp2_kernel<EPI>), dim3(p.tiles_m * p.tiles_n), dim3(PP_THREADS), PP2_LDS, s, p);
  return 0;
}

int launch_pp2_epi(GemmP& p, int epilogue, hipStream_t s) {
  switch (epilogue) {
    case SF_EPI_BIAS: return launch_pp2<SF_EPI_BIAS>(p, s);
    case SF_EPI_BIAS_GELU: return launch_pp2<SF_EPI_BIAS_GELU>(p, s);
    case SF_EPI_BIAS_RESID: return launch_pp2<SF_EPI_BIAS_RESID>(p, s);
    case SF_EPI_BIAS_GATE_RESID: return launch_pp2<SF_EPI_BIAS_GATE_RESID>(p, s);
    default: return -1;
  }
}

template <int EPI, int MT>
int launch_pp(GemmP& p, hipStream_t s) {
  using Cfg = PPCfg<MT>;
  static bool attr = false;   // one-time registration of the kernel's LDS size (idempotent)
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<EPI, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    attr = true;
  }
  p.tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM;
  p.tiles_n = (p.N + 255) / 256;
  hipLaunchKernelGGL((gemm_pp_kernel<EPI, MT>), dim3(p.tiles_m * p.tiles_n), dim3(PP_THREADS), Cfg::LDS, s, p);
  return 0;
}

template <int MT>
int launch_pp_epi(GemmP& p, int epilogue, hipStream_t s) {
  switch (epilogue) {
    case SF_EPI_BIAS: return launch_pp<SF_EPI_BIAS, MT>(p, s);
    case SF_EPI_BIAS_GELU: return launch_pp<SF_EPI_BIAS_GELU, MT>(p, s);
    case SF_EPI_BIAS_RESID: return launch_pp<SF_EPI_BIAS_RESID, MT>(p, s);
    case SF_EPI_BIAS_GATE_RESID: return launch_pp<SF_EPI_BIAS_GATE_RESID, MT>(p, s);
    default: return -1;
  }
}

}  // namespace

extern "C" int sf_gemm_bf16(const sf_gemm_args* a, void* stream) {
  SF_CHECK(a != nullptr, "sf_gemm_bf16: null args");
  SF_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "sf_gemm_bf16: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  SF_CHECK(a->K % BK == 0, "sf_gemm_bf16: K=%d must be a multiple of %d", a->K, BK);
  SF_CHECK(a->N % 4 == 0, "sf_gemm_bf16: N=%d must be a multiple of 4", a->N);
  SF_CHECK(a->lda % 8 == 0 && a->ldw % 8 == 0 && a->ldo % 4 == 0, "sf_gemm_bf16: lda/ldw must be multiples of 8, ldo of 4");
  SF_CHECK(a->lda >= a->K && a->ldw >= a->K && a->ldo >= a->N, "sf_gemm_bf16: leading dimension too small");
  SF_CHECK(a->a && a->w && a->out, "sf_gemm_bf16: null tensor");
  SF_CHECK(((uintptr_t)a->a % 16 == 0) && ((uintptr_t)a->w % 16 == 0) && ((uintptr_t)a->out % 8 == 0),
           "sf_gemm_bf16: misaligned tensor");
  if (a->epilogue == SF_EPI_BIAS_RESID || a->epilogue == SF_EPI_BIAS_GATE_RESID) {
    SF_CHECK(a->resid != nullptr && a->ldr >= a->N && a->ldr % 4 == 0, "sf_gemm_bf16: residual epilogue needs resid/ldr");
  }
  if (a->epilogue == SF_EPI_BIAS_GATE_RESID) {
    SF_CHECK(a->gate_mod && a->gate_e0 && a->rows_per_group > 0, "sf_gemm_bf16: gate epilogue needs gate_mod/gate_e0/rows_per_group");
  }
  GemmP p;
  p.a = (const bf16_t*)a->a; p.w = (const bf16_t*)a->w; p.bias = (const bf16_t*)a->bias;
  p.out = (bf16_t*)a->out; p.resid = (const bf16_t*)a->resid;
  p.gate_mod = (const bf16_t*)a->gate_mod; p.gate_e0 = (const bf16_t*)a->gate_e0;
  p.gate_group_stride = a->gate_group_stride; p.rows_per_group = a->rows_per_group;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo; p.ldr = a->ldr;
  p.tiles_n = (a->N + BN - 1) / BN;
  hipStream_t s = (hipStream_t)stream;
  p.tiles_m = (a->M + BM - 1) / BM;
  SF_CHECK(a->structure >= SF_GEMM_AUTO && a->structure <= SF_GEMM_PP192, "sf_gemm_bf16: unknown structure %d", a->structure);
  {
    int st = a->structure;
    const bool pp_ok = a->batch <= 1 && a->epilogue != SF_EPI_F32 && a->K >= 2 * BK;
    if (st == SF_GEMM_AUTO) {
      // Large problems take the ping-pong structure with the row tile (256 / 224 / 192 / 128 rows x 256 columns, one
      // workgroup per CU) that needs the least time for its ROUNDS of 256 workgroups: rounds x (a tile's k-loop, which
      // scales with its rows, slightly worse for the smaller register tiles, + a fixed prologue share).  Everything
      // else: 128 x 128 tiles, two per CU.
      st = SF_GEMM_T128;
      if (pp_ok && a->M >= 1024 && a->N >= 1024 && a->K >= 512) {
        static const struct { int st, rows; double rel; } cand[] = {
            {SF_GEMM_PP256, 256, 1.00}, {SF_GEMM_PP224, 224, 1.02}, {SF_GEMM_PP192, 192, 1.05}, {SF_GEMM_PP128, 128, 1.14}};
        const long tn = (a->N + 255) / 256;
        double best = 1e30;
        for (const auto& c : cand) {
          const long tiles = (long)((a->M + c.rows - 1) / c.rows) * tn;
          const double cost = (double)((tiles + 255) / 256) * (c.rows / 256.0 * c.rel + 0.04);
          if (cost < best - 1e-9) { best = cost; st = c.st; }
        }
      }
    }
    const bool is_pp = st == SF_GEMM_PP256 || st == SF_GEMM_PP224 || st == SF_GEMM_PP192 || st == SF_GEMM_PP128;
    if (is_pp && !pp_ok) st = SF_GEMM_T128;
    else if (is_pp) {
      const int rc = st == SF_GEMM_PP256 ? launch_pp_epi<8>(p, a->epilogue, s)
                   : st == SF_GEMM_PP224 ? launch_pp_epi<7>(p, a->epilogue, s)
                   : st == SF_GEMM_PP192 ? launch_pp_epi<6>(p, a->epilogue, s) : launch_pp2_epi(p, a->epilogue, s);
      SF_CHECK(rc == 0, "sf_gemm_bf16: unknown epilogue %d", a->epilogue);
      SF_HIP_LAUNCH_CHECK("sf_gemm_bf16");
      return 0;
    }
  }
  const int batch = a->batch > 1 ? a->batch : 1;
  p.a_bs = a->a_bstride; p.w_bs = a->w_bstride; p.o_bs = a->o_bstride; p.r_bs = a->r_bstride;
  SF_CHECK(batch == 1 || (a->a_bstride % 8 == 0 && a->w_bstride % 8 == 0 && a->o_bstride % 4 == 0 && a->r_bstride % 4 == 0),
           "sf_gemm_bf16: batch strides must keep 16-byte (operands) / 8-byte (output) alignment");
  const dim3 grid(p.tiles_m * p.tiles_n, batch), block(GEMM_THREADS);
  switch (a->epilogue) {
    case SF_EPI_BIAS: hipLaunchKernelGGL(gemm_bf16_kernel<SF_EPI_BIAS>, grid, block, GEMM_LDS, s, p); break;
    case SF_EPI_BIAS_GELU: hipLaunchKernelGGL(gemm_bf16_kernel<SF_EPI_BIAS_GELU>, grid, block, GEMM_LDS, s, p); break;
    case SF_EPI_BIAS_RESID: hipLaunchKernelGGL(gemm_bf16_kernel<SF_EPI_BIAS_RESID>, grid, block, GEMM_LDS, s, p); break;
    case SF_EPI_BIAS_GATE_RESID: hipLaunchKernelGGL(gemm_bf16_kernel<SF_EPI_BIAS_GATE_RESID>, grid, block, GEMM_LDS, s, p); break;
    case SF_EPI_F32:
      SF_CHECK((uintptr_t)a->out % 16 == 0, "sf_gemm_bf16: fp32 output must be 16-byte aligned");
      hipLaunchKernelGGL(gemm_bf16_kernel<SF_EPI_F32>, grid, block, GEMM_LDS, s, p);
      break;
    default: SF_CHECK(false, "sf_gemm_bf16: unknown epilogue %d", a->epilogue);
  }
  SF_HIP_LAUNCH_CHECK("sf_gemm_bf16");
  return 0;
}
