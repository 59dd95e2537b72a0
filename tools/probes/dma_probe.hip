// Issue cost of LDS-DMA (global -> LDS without registers) on gfx950, one wave per SIMD:
//   buffer_load_dwordx4 ... offen lds   (1 KiB per instruction) in a burst of 8, spread between MFMAs, with
//   linear / row-swizzled lane addresses, and the dword (256 B per instruction) form for comparison.
//   hipcc -O3 --offload-arch=gfx950 dma_probe.hip -o dma_probe && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2000;
#define MFMA2 "v_mfma_f32_32x32x16_bf16 v[0:15], v[64:67], v[68:71], v[0:15]\n v_mfma_f32_32x32x16_bf16 v[16:31], v[64:67], v[68:71], v[16:31]\n"
#define DMA4(off) "s_add_u32 m0, %2, " #off "\n s_nop 1\n buffer_load_dwordx4 %0, %1, %3 offen lds\n"
#define DMA1(off) "s_add_u32 m0, %2, " #off "\n s_nop 1\n buffer_load_dword %0, %1, %3 offen lds\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v64","v65","v66","v67","v68","v69","v70","v71","memory"

template <int KIND>
__global__ __launch_bounds__(64) void k(const char* src, float* out, int swz) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x;
  const unsigned long long a64 = (unsigned long long)src + (size_t)(blockIdx.x & 255) * 262144;
  u32x4 srd;
  srd[0] = __builtin_amdgcn_readfirstlane((unsigned)a64);
  srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(a64 >> 32) & 0xFFFFu);
  srd[2] = 262144u;
  srd[3] = 0x00020000u;
  const unsigned lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  // linear: lane l reads bytes 16 l; "rows": 4 rows of 3072 B apart (the K/V tile pattern: 16 lanes per row, chunk swizzled)
  unsigned voff = swz ? (unsigned)((lane >> 4) * 3072 + (((lane & 15) ^ ((lane >> 4) * 5)) & 15) * 16) : (unsigned)lane * 16;
  for (int it = 0; it < ITER; ++it) {
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((it & 15) * 12288));
    if (KIND == 0) asm volatile(DMA4(0) DMA4(1024) DMA4(2048) DMA4(3072) DMA4(4096) DMA4(5120) DMA4(6144) DMA4(7168) "s_waitcnt vmcnt(0)\n" :: "v"(voff), "s"(srd), "s"(lds), "s"(soff) : CLOB);
    if (KIND == 1) asm volatile(MFMA2 DMA4(0) MFMA2 DMA4(1024) MFMA2 DMA4(2048) MFMA2 DMA4(3072) MFMA2 DMA4(4096) MFMA2 DMA4(5120) MFMA2 DMA4(6144) MFMA2 DMA4(7168) "s_waitcnt vmcnt(0)\n" :: "v"(voff), "s"(srd), "s"(lds), "s"(soff) : CLOB);
    if (KIND == 2) asm volatile(MFMA2 MFMA2 MFMA2 MFMA2 MFMA2 MFMA2 MFMA2 MFMA2 :: "v"(voff), "s"(srd), "s"(lds), "s"(soff) : CLOB);
    if (KIND == 3) asm volatile(DMA1(0) DMA1(256) DMA1(512) DMA1(768) DMA1(1024) DMA1(1280) DMA1(1536) DMA1(1792) "s_waitcnt vmcnt(0)\n" :: "v"(voff), "s"(srd), "s"(lds), "s"(soff) : CLOB);
    if (KIND == 4) asm volatile(DMA4(0) DMA4(1024) DMA4(2048) DMA4(3072) DMA4(4096) DMA4(5120) DMA4(6144) DMA4(7168) :: "v"(voff), "s"(srd), "s"(lds), "s"(soff) : CLOB);   // no wait inside the loop
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 64 + lane] = smem[lane];
}
template <int KIND> float run(const char* src, float* d, int swz) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(64), 16384, 0, src, d, swz);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(64), 16384, 0, src, d, swz);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  char* src; float* d;
  hipMalloc(&src, 256 * 262144 + 65536); hipMemset(src, 1, 256 * 262144 + 65536);
  hipMalloc(&d, 1024 * 64 * 4);
  const char* names[] = {"8 x dwordx4 LDS-DMA burst + wait", "8 x (2 MFMA + dwordx4 LDS-DMA) + wait", "16 MFMA", "8 x dword LDS-DMA burst + wait", "8 x dwordx4 LDS-DMA, no wait in the loop"};
  for (int swz = 0; swz < 2; ++swz) {
    float t[5] = {run<0>(src, d, swz), run<1>(src, d, swz), run<2>(src, d, swz), run<3>(src, d, swz), run<4>(src, d, swz)};
    for (int i = 0; i < 5; ++i) printf("%s %-44s %8.3f ms  %7.1f ns per iteration (8 DMA)\n", swz ? "rows  " : "linear", names[i], t[i], 1e6 * t[i] / ITER);
  }
  return 0;
}
