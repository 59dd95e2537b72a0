// Issue-rate probe for the VALU instructions the attention softmax is made of (gfx950).
// One wave per SIMD (256 CUs x 4), each runs ITER x 16 independent instructions of one kind;
// the rate relative to v_fma_f32 (one 64-lane instruction per 4 cycles) gives cycles/instruction.
//   hipcc -O3 --offload-arch=gfx950 valu_probe.hip -o valu_probe && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 20000;
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, float seed) {
  float a[16]; f32x2 b[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; b[i] = f32x2{a[i], a[i] + 1.f}; }
  const float c = seed * 0.5f; const f32x2 c2 = {c, c};
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 1) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      REP16(X)
#undef X
    } else if (KIND == 2) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(c2));
      REP16(X)
#undef X
    } else if (KIND == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
      REP16(X)
#undef X
    } else if (KIND == 4) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 5) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 6) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 7) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
      REP16(X)
#undef X
    } else if (KIND == 8) {   // exp and fma alternating: does the transcendental unit overlap the main ALU?
#define X(i) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %2" : "+v"(a[i]), "+v"(b[i][0]) : "v"(c));
      REP16(X)
#undef X
    } else if (KIND == 9) {   // exp + 3 fma
#define X(i) asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %3, %3\n\tv_fma_f32 %2, %2, %3, %3\n\tv_add_f32 %1, %1, %3" : "+v"(a[i]), "+v"(b[i][0]), "+v"(b[i][1]) : "v"(c));
      REP16(X)
#undef X
    }
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + b[i][0] + b[i][1];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int KIND> float run(float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(64), 0, 0, d, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(64), 0, 0, d, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; hipMalloc(&d, 1024 * 64 * 4);
  const char* names[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_add_f32", "v_pk_mul_f32", "exp+fma pair", "exp+3 valu"};
  float t[10] = {run<0>(d), run<1>(d), run<2>(d), run<3>(d), run<4>(d), run<5>(d), run<6>(d), run<7>(d), run<8>(d), run<9>(d)};
  for (int i = 0; i < 10; ++i)
    printf("%-20s %8.3f ms  %6.2f cycles per instr-group (v_fma_f32 = 4)\n", names[i], t[i], 4.0 * t[i] / t[0]);
  return 0;
}
