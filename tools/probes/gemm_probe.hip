// GEMM inner-loop ceilings on gfx950 (random bf16): where does a 64x64-per-wave 16x16x32 register
// tile fed from LDS stop scaling?  Build: hipcc --offload-arch=gfx950 -O3 gemm_probe.hip -o gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// TM x TN MFMA tiles per wave (each 16x16), BK = 64 per "k-step" (2 sub-steps of 32).
// MODE bit0: s_barrier per k-step; bit1: LDS-DMA staging traffic (per wave: (TM_ROWS+TN_ROWS)/... pieces);
template <int TM, int TN, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS, 2) void probe(const bf16x8* __restrict__ in, float* __restrict__ out, int iters,
                                                    const char* __restrict__ gsrc, int dma_pieces) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 3072; i += blockDim.x) reinterpret_cast<bf16x8*>(smem)[i] = in[i & 511];   // 48 KiB
  __syncthreads();
  const int i16 = lane & 15, kq = lane >> 4, swz = (i16 >> 1) & 7;
  const int row_off = i16 * 128;
  const int coff0 = ((0 + kq) ^ swz) << 4, coff1 = ((4 + kq) ^ swz) << 4;
  f32x4 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const char* gp = gsrc + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 2) {
      for (int i = 0; i < dma_pieces; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(gp + ((it * 8 + i) & 63) * 1024), (lptr_t)(smem + 49152 + ((it % 3) * 16 + wave * 2 + (i & 1)) * 1024), 16, 0, 0);
    }
    const char* buf = smem + (it & 1) * 8192;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int coff = s2 ? coff1 : coff0;
      bf16x8 xf[TM], wf[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) xf[t] = *reinterpret_cast<const bf16x8*>(buf + row_off + t * 2048 + coff);
#pragma unroll
      for (int t = 0; t < TN; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(buf + 16384 + row_off + t * 2048 + coff);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
    }
    if (MODE & 2) { if (dma_pieces == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    if (MODE & 1) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * blockDim.x + tid] = s;
}

static char* g_src;
template <int TM, int TN, int MODE, int THREADS>
void run(const char* name, int blocks, int iters, const bf16x8* in, float* out, int dma_pieces = 6) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = 49152 + 3 * 16 * 1024;
  hipLaunchKernelGGL((probe<TM, TN, MODE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, in, out, 16, g_src, dma_pieces);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<TM, TN, MODE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, in, out, iters, g_src, dma_pieces);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double flops = (double)blocks * (THREADS / 64) * iters * (2.0 * TM * TN) * (16.0 * 16 * 32 * 2);
  printf("%-78s %3d thr x %4d blk: %8.3f ms  %8.1f TFLOP/s\n", name, THREADS, blocks, ms, flops / ms / 1e9);
  fflush(stdout);
}

int main() {
  std::vector<unsigned short> h(512 * 8);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8* in; float* out;
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 4096 * 512 * 4);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&g_src, 64 * 65536 + 65536); hipMemset(g_src, 1, 64 * 65536 + 65536);
  const int it = 3000;
  run<4, 4, 0, 256>("4x4 tiles/wave (64x64), LDS-fed, no barrier, 1 wave/SIMD", 256, it, in, out);
  run<4, 4, 0, 512>("4x4 tiles/wave (64x64), LDS-fed, no barrier, 2 waves/SIMD", 256, it, in, out);
  run<4, 4, 1, 512>("4x4 + s_barrier per k-step, 2 waves/SIMD", 256, it, in, out);
  run<4, 4, 3, 512>("4x4 + barrier + LDS-DMA 6 pieces/wave/k-step (48 KiB per WG, L2 resident)", 256, it, in, out);
  run<8, 4, 0, 256>("8x4 tiles/wave (128x64), LDS-fed, no barrier, 1 wave/SIMD", 256, it, in, out);
  run<8, 4, 0, 512>("8x4 tiles/wave (128x64), LDS-fed, no barrier, 2 waves/SIMD", 256, it, in, out);
  run<8, 4, 1, 512>("8x4 + s_barrier per k-step, 2 waves/SIMD", 256, it, in, out);
  run<8, 4, 3, 512>("8x4 + barrier + LDS-DMA 8 pieces/wave/k-step (64 KiB per WG)", 256, it, in, out, 8);
  run<8, 8, 0, 256>("8x8 tiles/wave (128x128), LDS-fed, no barrier, 1 wave/SIMD", 256, it, in, out);
  return 0;
}
