// Does a packed fp32 VALU instruction give wrong results when waves of another kernel share the CU?
// (The q/k RoPE kernel did, beside the VAE convolution: DESIGN.md section 7.)  Victim kernel: every lane evaluates
// a few packed-fp32 forms next to their scalar equivalents on changing data and counts the mismatches per lane;
// the co-runner is launched from Python on another stream (tools/probes/pk_probe.py).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC pk_probe.hip -o pk_probe.so
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void pk_victim(unsigned* __restrict__ counts, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  float a0 = seed + 0.001f * threadIdx.x, a1 = 1.5f - 0.002f * threadIdx.x, b0 = 0.75f + 0.0001f * blockIdx.x, b1 = -1.25f, c0 = 0.3f, c1 = -0.7f;
  unsigned bad[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x2 a = {a0, a1}, b = {b0, b1}, c = {c0, c1}, r;
    // (0) plain packed fma
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    float s0 = __builtin_fmaf(a0, b0, c0), s1 = __builtin_fmaf(a1, b1, c1);
    bad[0] += (r[0] != s0) + (r[1] != s1);
    // (1) packed fma, src0 low half for both lanes (op_sel_hi:[0,1,1])
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    s0 = __builtin_fmaf(a0, b0, c0); s1 = __builtin_fmaf(a0, b1, c1);
    bad[1] += (r[0] != s0) + (r[1] != s1);
    // (2) packed mul with op_sel:[0,1] op_sel_hi:[0,0]: lo = a0 * b1, hi = a0 * b0
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    s0 = a0 * b1; s1 = a0 * b0;
    bad[2] += (r[0] != s0) + (r[1] != s1);
    // (3) packed fma with negated addend (neg_lo / neg_hi on src2)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    s0 = __builtin_fmaf(a0, b0, -c0); s1 = __builtin_fmaf(a1, b1, -c1);
    bad[3] += (r[0] != s0) + (r[1] != s1);
    // (4) control: two scalar fmas compared with themselves through a register copy
    float t0, t1;
    asm volatile("v_fma_f32 %0, %2, %3, %4\n\tv_fma_f32 %1, %5, %6, %7" : "=&v"(t0), "=&v"(t1) : "v"(a0), "v"(b0), "v"(c0), "v"(a1), "v"(b1), "v"(c1));
    bad[4] += (t0 != __builtin_fmaf(a0, b0, c0)) + (t1 != __builtin_fmaf(a1, b1, c1));
    // (5) plain packed mul
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    bad[5] += (r[0] != a0 * b0) + (r[1] != a1 * b1);
    // (6) packed mul, op_sel:[1,0] op_sel_hi:[1,1]: lo = a1 * b0, hi = a1 * b1
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    bad[6] += (r[0] != a1 * b0) + (r[1] != a1 * b1);
    // (7) packed add with op_sel:[0,1] op_sel_hi:[0,0]: lo = a0 + b1, hi = a0 + b0
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    bad[7] += (r[0] != a0 + b1) + (r[1] != a0 + b0);
    // (8) packed mul, op_sel:[0,1] op_sel_hi:[1,0]: lo = a0 * b1, hi = a1 * b0 (a plain swap of src1's halves)
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    bad[8] += (r[0] != a0 * b1) + (r[1] != a1 * b0);
    // new data every iteration (bounded)
    a0 = a0 * 0.999f + 0.01f; a1 = a1 * 0.998f - 0.02f; b0 = b0 * 1.0001f; c0 += 0.125f; if (c0 > 8.f) c0 -= 8.f; c1 = -c1 * 0.99f;
  }
#pragma unroll
  for (int v = 0; v < 9; ++v)
    if (bad[v]) atomicAdd(&counts[v * 64 + lane], bad[v]);
}

extern "C" int pk_probe_launch(void* counts, int blocks, int iters, float seed, void* stream) {
  hipLaunchKernelGGL(pk_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)counts, iters, seed);
  return (int)hipGetLastError();
}
