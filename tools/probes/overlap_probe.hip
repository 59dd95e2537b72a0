// Does a wave's own VALU work run in the shadow of its MFMAs?  One wave per SIMD (1024 waves), loop of
// 16 x v_mfma_f32_32x32x16_bf16 (two alternating accumulators) with N independent VALU instructions
// after each MFMA.  Variants: accumulators in VGPRs / AGPRs, VALU = v_fma_f32 or v_exp_f32.
//   hipcc -O3 --offload-arch=gfx950 overlap_probe.hip -o overlap_probe && ./overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 1500;
#define STR2(x) #x
#define STR(x) STR2(x)
#define VALU6 "v_fma_f32 v100, v100, v116, v116\n v_fma_f32 v101, v101, v116, v116\n v_fma_f32 v102, v102, v116, v116\n" \
              "v_fma_f32 v103, v103, v116, v116\n v_fma_f32 v104, v104, v116, v116\n v_fma_f32 v105, v105, v116, v116\n"
#define PKADD6 "v_pk_add_f32 v[100:101], v[100:101], v[116:117]\n v_pk_add_f32 v[102:103], v[102:103], v[116:117]\n v_pk_add_f32 v[104:105], v[104:105], v[116:117]\n v_pk_add_f32 v[106:107], v[106:107], v[116:117]\n v_pk_add_f32 v[108:109], v[108:109], v[116:117]\n v_pk_add_f32 v[110:111], v[110:111], v[116:117]\n"
#define CVT6 "v_cvt_pk_bf16_f32 v100, v100, v116\n v_cvt_pk_bf16_f32 v101, v101, v116\n v_cvt_pk_bf16_f32 v102, v102, v116\n v_cvt_pk_bf16_f32 v103, v103, v116\n v_cvt_pk_bf16_f32 v104, v104, v116\n v_cvt_pk_bf16_f32 v105, v105, v116\n"
#define MAX6 "v_max3_f32 v100, v100, v116, v116\n v_max3_f32 v101, v101, v116, v116\n v_max3_f32 v102, v102, v116, v116\n v_max3_f32 v103, v103, v116, v116\n v_max3_f32 v104, v104, v116, v116\n v_max3_f32 v105, v105, v116, v116\n"
#define ADD6 "v_add_f32 v100, v100, v116\n v_add_f32 v101, v101, v116\n v_add_f32 v102, v102, v116\n v_add_f32 v103, v103, v116\n v_add_f32 v104, v104, v116\n v_add_f32 v105, v105, v116\n"
#define EXP6  "v_exp_f32 v100, v100\n v_exp_f32 v101, v101\n v_exp_f32 v102, v102\n v_exp_f32 v103, v103\n v_exp_f32 v104, v104\n v_exp_f32 v105, v105\n"
#define MFMA_V(a) "v_mfma_f32_32x32x16_bf16 v[" a "], v[64:67], v[68:71], v[" a "]\n"
#define MFMA_A(a) "v_mfma_f32_32x32x16_bf16 a[" a "], v[64:67], v[68:71], a[" a "]\n"
#define MFMA_SA(a) "v_mfma_f32_32x32x16_bf16 v[" a "], a[40:43], a[44:47], v[" a "]\n"
#define MFMA_AA(a) "v_mfma_f32_32x32x16_bf16 a[" a "], a[40:43], v[68:71], a[" a "]\n"
#define LDS2 "ds_read_b128 v[72:75], v116 offset:0\n ds_read_b64_tr_b16 v[76:77], v116 offset:4096\n"
#define LDS2A "ds_read_b128 a[48:51], v116 offset:0\n ds_read_b64_tr_b16 a[52:53], v116 offset:4096\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
  "v64","v65","v66","v67","v68","v69","v70","v71","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v116","v117", \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","v72","v73","v74","v75","v76","v77"

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out) {
  __shared__ char lds_buf[8192];
  if (threadIdx.x == 1000) lds_buf[0] = 1;
  unsigned long long r0 = __builtin_readcyclecounter();          // s_memtime: constant 100 MHz
  unsigned long long c0 = clock64();
  unsigned cyc0 = __builtin_amdgcn_s_getreg((29 << 0) | (0 << 6) | (19 << 11));   // HW_REG_SHADER_CYCLES, 20 bits
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) asm volatile(MFMA_V("0:15") MFMA_V("16:31") MFMA_V("0:15") MFMA_V("16:31") MFMA_V("0:15") MFMA_V("16:31") MFMA_V("0:15") MFMA_V("16:31") ::: CLOB);
    if (KIND == 1) asm volatile(MFMA_V("0:15") VALU6 MFMA_V("16:31") VALU6 MFMA_V("0:15") VALU6 MFMA_V("16:31") VALU6 MFMA_V("0:15") VALU6 MFMA_V("16:31") VALU6 MFMA_V("0:15") VALU6 MFMA_V("16:31") VALU6 ::: CLOB);
    if (KIND == 2) asm volatile(VALU6 VALU6 VALU6 VALU6 VALU6 VALU6 VALU6 VALU6 ::: CLOB);
    if (KIND == 3) asm volatile(MFMA_A("0:15") VALU6 MFMA_A("16:31") VALU6 MFMA_A("0:15") VALU6 MFMA_A("16:31") VALU6 MFMA_A("0:15") VALU6 MFMA_A("16:31") VALU6 MFMA_A("0:15") VALU6 MFMA_A("16:31") VALU6 ::: CLOB);
    if (KIND == 4) asm volatile(MFMA_A("0:15") MFMA_A("16:31") MFMA_A("0:15") MFMA_A("16:31") MFMA_A("0:15") MFMA_A("16:31") MFMA_A("0:15") MFMA_A("16:31") ::: CLOB);
    if (KIND == 5) asm volatile(MFMA_V("0:15") EXP6 MFMA_V("16:31") EXP6 MFMA_V("0:15") EXP6 MFMA_V("16:31") EXP6 MFMA_V("0:15") EXP6 MFMA_V("16:31") EXP6 MFMA_V("0:15") EXP6 MFMA_V("16:31") EXP6 ::: CLOB);
    if (KIND == 6) asm volatile(EXP6 EXP6 EXP6 EXP6 EXP6 EXP6 EXP6 EXP6 ::: CLOB);
    if (KIND == 8) asm volatile(MFMA_SA("0:15") VALU6 MFMA_SA("16:31") VALU6 MFMA_SA("0:15") VALU6 MFMA_SA("16:31") VALU6 MFMA_SA("0:15") VALU6 MFMA_SA("16:31") VALU6 MFMA_SA("0:15") VALU6 MFMA_SA("16:31") VALU6 ::: CLOB);
    if (KIND == 9) asm volatile(MFMA_AA("0:15") VALU6 MFMA_AA("16:31") VALU6 MFMA_AA("0:15") VALU6 MFMA_AA("16:31") VALU6 MFMA_AA("0:15") VALU6 MFMA_AA("16:31") VALU6 MFMA_AA("0:15") VALU6 MFMA_AA("16:31") VALU6 ::: CLOB);
    if (KIND == 10) asm volatile("v_mov_b32 v116, 0\n" MFMA_V("0:15") LDS2 MFMA_V("16:31") LDS2 MFMA_V("0:15") LDS2 MFMA_V("16:31") LDS2 MFMA_V("0:15") LDS2 MFMA_V("16:31") LDS2 MFMA_V("0:15") LDS2 MFMA_V("16:31") LDS2 "s_waitcnt lgkmcnt(0)\n" ::: CLOB);
    if (KIND == 11) asm volatile("v_mov_b32 v116, 0\n" MFMA_V("0:15") LDS2A MFMA_V("16:31") LDS2A MFMA_V("0:15") LDS2A MFMA_V("16:31") LDS2A MFMA_V("0:15") LDS2A MFMA_V("16:31") LDS2A MFMA_V("0:15") LDS2A MFMA_V("16:31") LDS2A "s_waitcnt lgkmcnt(0)\n" ::: CLOB);
    if (KIND == 12) asm volatile("v_mov_b32 v116, 0\n" LDS2 LDS2 LDS2 LDS2 LDS2 LDS2 LDS2 LDS2 "s_waitcnt lgkmcnt(0)\n" ::: CLOB);
    if (KIND == 13) asm volatile(MFMA_V("0:15") PKADD6 MFMA_V("16:31") PKADD6 MFMA_V("0:15") PKADD6 MFMA_V("16:31") PKADD6 MFMA_V("0:15") PKADD6 MFMA_V("16:31") PKADD6 MFMA_V("0:15") PKADD6 MFMA_V("16:31") PKADD6 ::: CLOB);
    if (KIND == 14) asm volatile(MFMA_V("0:15") CVT6 MFMA_V("16:31") CVT6 MFMA_V("0:15") CVT6 MFMA_V("16:31") CVT6 MFMA_V("0:15") CVT6 MFMA_V("16:31") CVT6 MFMA_V("0:15") CVT6 MFMA_V("16:31") CVT6 ::: CLOB);
    if (KIND == 15) asm volatile(MFMA_V("0:15") MAX6 MFMA_V("16:31") MAX6 MFMA_V("0:15") MAX6 MFMA_V("16:31") MAX6 MFMA_V("0:15") MAX6 MFMA_V("16:31") MAX6 MFMA_V("0:15") MAX6 MFMA_V("16:31") MAX6 ::: CLOB);
    if (KIND == 16) asm volatile(MFMA_V("0:15") ADD6 MFMA_V("16:31") ADD6 MFMA_V("0:15") ADD6 MFMA_V("16:31") ADD6 MFMA_V("0:15") ADD6 MFMA_V("16:31") ADD6 MFMA_V("0:15") ADD6 MFMA_V("16:31") ADD6 ::: CLOB);
    if (KIND == 17) asm volatile(PKADD6 PKADD6 PKADD6 PKADD6 PKADD6 PKADD6 PKADD6 PKADD6 ::: CLOB);
    if (KIND == 7) asm volatile(MFMA_V("0:15") VALU6 VALU6 MFMA_V("16:31") VALU6 VALU6 MFMA_V("0:15") VALU6 VALU6 MFMA_V("16:31") VALU6 VALU6 MFMA_V("0:15") VALU6 VALU6 MFMA_V("16:31") VALU6 VALU6 MFMA_V("0:15") VALU6 VALU6 MFMA_V("16:31") VALU6 VALU6 ::: CLOB);
  }
  unsigned cyc1 = __builtin_amdgcn_s_getreg((29 << 0) | (0 << 6) | (19 << 11));
  unsigned long long r1 = __builtin_readcyclecounter();
  unsigned long long c1 = clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = (float)((cyc1 - cyc0) & 0xFFFFF); out[1] = (float)(r1 - r0); out[2] = (float)(c1 - c0);
  } else if (blockIdx.x > 0) out[blockIdx.x * 64 + threadIdx.x] = 0.f;
}
template <int KIND> float run(float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(64), 0, 0, d);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(64), 0, 0, d);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("   [shader cycles (20-bit) %.0f, s_memtime ticks %.0f, clock64 %.0f over the wave -> %.3f ms]\n", h[0], h[1], h[2], ms);
  return ms;
}
int main() {
  float* d; hipMalloc(&d, 1024 * 64 * 4);
  const char* names[] = {"8 MFMA (VGPR acc)", "8 MFMA + 48 v_fma interleaved", "48 v_fma", "8 MFMA (AGPR acc) + 48 v_fma", "8 MFMA (AGPR acc)",
                         "8 MFMA + 48 v_exp", "48 v_exp", "8 MFMA + 96 v_fma", "8 MFMA (A,B from AGPR) + 48 v_fma", "8 MFMA (A from AGPR, AGPR acc) + 48 v_fma",
                         "8 MFMA + 16 LDS reads -> VGPR", "8 MFMA + 16 LDS reads -> AGPR", "16 LDS reads", "8 MFMA + 48 v_pk_add_f32", "8 MFMA + 48 v_cvt_pk_bf16_f32", "8 MFMA + 48 v_max3_f32", "8 MFMA + 48 v_add_f32", "48 v_pk_add_f32"};
  float t[18] = {run<0>(d), run<1>(d), run<2>(d), run<3>(d), run<4>(d), run<5>(d), run<6>(d), run<7>(d), run<8>(d), run<9>(d), run<10>(d), run<11>(d), run<12>(d), run<13>(d), run<14>(d), run<15>(d), run<16>(d), run<17>(d)};
  for (int i = 0; i < 18; ++i) printf("%-48s %8.3f ms  %7.1f ns per group of 8 MFMA slots\n", names[i], t[i], 1e6 * t[i] / ITER);
  return 0;
}
