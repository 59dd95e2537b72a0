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
#include <cstdlib>
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
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// Epilogue shared by both structures: the lane holds out[m][n .. n+3] for each (mt, nt) of its
// wave's 64 x 64 sub-tile; m = mrow + 16 mt, n = ncol + 16 nt.
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
      float y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = acc[mt][nt][j];
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
      *reinterpret_cast<bf16x4*>(p.out + (long)m * p.ldo + n) = o;
    }
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
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* buf = smem + cur * STAGE_BYTES;
    // fragments of both 32-deep sub-steps: the second set is requested while the first set's MFMAs
    // run (issue order pinned below; left alone, hipcc reads just in time and every group of MFMAs
    // waits out an LDS round trip)
    bf16x8 xf0[4], wf0[4], xf1[4], wf1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      xf0[t] = *reinterpret_cast<const bf16x8*>(buf + x_row_off + t * 2048 + coff0);
      wf0[t] = *reinterpret_cast<const bf16x8*>(buf + w_row_off + t * 2048 + coff0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      xf1[t] = *reinterpret_cast<const bf16x8*>(buf + x_row_off + t * 2048 + coff1);
      wf1[t] = *reinterpret_cast<const bf16x8*>(buf + w_row_off + t * 2048 + coff1);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt], xf0[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], xf1[mt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    __syncthreads();  // drains the in-flight LDS-DMA (vmcnt(0)) and orders the buffer swap
  }

  // ---- epilogue: lane holds out[m][n .. n+3] for (mt, nt)
  const int mrow = m0 + wr * 64 + (lane & 15);
  const int ncol = n0 + wc * 64 + (lane >> 4) * 4;
  gemm_epilogue<EPI, 4>(p, acc, mrow, ncol);
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
  const dim3 grid(p.tiles_m * p.tiles_n), block(GEMM_THREADS);
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
