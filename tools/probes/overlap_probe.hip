// Does a wave's own VALU work run in the shadow of its MFMAs?  One wave per SIMD (1024 waves), loop of
// 16 x v_mfma_f32_32x32x16_bf16 (two alternating accumulators) with N independent VALU instructions
// after each MFMA.  Variants: accumulators in VGPRs / AGPRs, VALU = v_fma_f32 or v_exp_f32.
//   hipcc -O3 --offload-arch=gfx950 overlap_probe.hip -o overlap_probe && ./overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 4000;
#define STR2(x) #x
#define STR(x) STR2(x)
#define VALU6 "v_fma_f32 v100, v100, v116, v116\n v_fma_f32 v101, v101, v116, v116\n v_fma_f32 v102, v102, v116, v116\n" \
              "v_fma_f32 v103, v103, v116, v116\n v_fma_f32 v104, v104, v116, v116\n v_fma_f32 v105, v105, v116, v116\n"
#define EXP6  "v_exp_f32 v100, v100\n v_exp_f32 v101, v101\n v_exp_f32 v102, v102\n v_exp_f32 v103, v103\n v_exp_f32 v104, v104\n v_exp_f32 v105, v105\n"
#define MFMA_V(a) "v_mfma_f32_32x32x16_bf16 v[" a "], v[64:67], v[68:71], v[" a "]\n"
#define MFMA_A(a) "v_mfma_f32_32x32x16_bf16 a[" a "], v[64:67], v[68:71], a[" a "]\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
  "v64","v65","v66","v67","v68","v69","v70","v71","v100","v101","v102","v103","v104","v105","v116", \
  "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out) {
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) asm volatile(MFMA_V("0:15") MFMA_V("16:31") MFMA_V("0:15") MFMA_V("16:31") MFMA_V("0:15") MFMA_V("16:31") MFMA_V("0:15") MFMA_V("16:31") ::: CLOB);
    if (KIND == 1) asm volatile(MFMA_V("0:15") VALU6 MFMA_V("16:31") VALU6 MFMA_V("0:15") VALU6 MFMA_V("16:31") VALU6 MFMA_V("0:15") VALU6 MFMA_V("16:31") VALU6 MFMA_V("0:15") VALU6 MFMA_V("16:31") VALU6 ::: CLOB);
    if (KIND == 2) asm volatile(VALU6 VALU6 VALU6 VALU6 VALU6 VALU6 VALU6 VALU6 ::: CLOB);
    if (KIND == 3) asm volatile(MFMA_A("0:15") VALU6 MFMA_A("16:31") VALU6 MFMA_A("0:15") VALU6 MFMA_A("16:31") VALU6 MFMA_A("0:15") VALU6 MFMA_A("16:31") VALU6 MFMA_A("0:15") VALU6 MFMA_A("16:31") VALU6 ::: CLOB);
    if (KIND == 4) asm volatile(MFMA_A("0:15") MFMA_A("16:31") MFMA_A("0:15") MFMA_A("16:31") MFMA_A("0:15") MFMA_A("16:31") MFMA_A("0:15") MFMA_A("16:31") ::: CLOB);
    if (KIND == 5) asm volatile(MFMA_V("0:15") EXP6 MFMA_V("16:31") EXP6 MFMA_V("0:15") EXP6 MFMA_V("16:31") EXP6 MFMA_V("0:15") EXP6 MFMA_V("16:31") EXP6 MFMA_V("0:15") EXP6 MFMA_V("16:31") EXP6 ::: CLOB);
    if (KIND == 6) asm volatile(EXP6 EXP6 EXP6 EXP6 EXP6 EXP6 EXP6 EXP6 ::: CLOB);
    if (KIND == 7) asm volatile(MFMA_V("0:15") VALU6 VALU6 MFMA_V("16:31") VALU6 VALU6 MFMA_V("0:15") VALU6 VALU6 MFMA_V("16:31") VALU6 VALU6 MFMA_V("0:15") VALU6 VALU6 MFMA_V("16:31") VALU6 VALU6 MFMA_V("0:15") VALU6 VALU6 MFMA_V("16:31") VALU6 VALU6 ::: CLOB);
  }
  out[blockIdx.x * 64 + threadIdx.x] = 0.f;
}
template <int KIND> float run(float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(64), 0, 0, d);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(64), 0, 0, d);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; hipMalloc(&d, 1024 * 64 * 4);
  const char* names[] = {"8 MFMA (VGPR acc)", "8 MFMA + 48 v_fma interleaved", "48 v_fma", "8 MFMA (AGPR acc) + 48 v_fma", "8 MFMA (AGPR acc)",
                         "8 MFMA + 48 v_exp", "48 v_exp", "8 MFMA + 96 v_fma"};
  float t[8] = {run<0>(d), run<1>(d), run<2>(d), run<3>(d), run<4>(d), run<5>(d), run<6>(d), run<7>(d)};
  for (int i = 0; i < 8; ++i) printf("%-34s %8.3f ms  %7.1f ns per group of 8 MFMA slots\n", names[i], t[i], 1e6 * t[i] / ITER);
  return 0;
}
