// Ceiling probes for gfx950: what the matrix pipe delivers on THIS box under controlled conditions
// (random bf16 operands).  Build: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 4 independent 32x32x16 accumulators, operands in registers
// MODE 1: ONE accumulator (dependent chain)
// MODE 2: 16x16x32, 4 independent accumulators
// MODE 3: 32x32x16, A operand re-read from LDS (ds_read_b128, conflict-free image) every MFMA, 6 reads ahead
// MODE 4: MODE 0 + s_barrier every 32 MFMAs
// MODE 5: MODE 3 + s_barrier every 32 MFMAs
// MODE 6: anti-phase: waves 0-3 run 32 MFMAs (LDS-fed) while waves 4-7 idle, barrier, swap (the attention structure)
// MODE 7: MODE 6, and the non-matrix half runs a softmax-like VALU block (32 x {fma, exp2, add}, 16 max3, 16 cvt_pk)
// MODE 8: MODE 7 + staging traffic: every thread 4 x 16-byte global loads and 4 x ds_write_b128 per pair of steps
// MODE 9: MODE 7 with the VALU block but NO exp (fma only)
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const bf16x8* __restrict__ in, float* __restrict__ out, int iters, const u32x4* __restrict__ gsrc) {
  u32x4 stg[4] = {};
  bf16x8 bsink = {}, bsink2 = {};
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16x8 a = in[tid & 255], b = in[256 + (tid & 255)];
  // fill 32 KiB of LDS
  for (int i = tid; i < 2048; i += blockDim.x) reinterpret_cast<bf16x8*>(smem)[i] = in[i & 511];
  __syncthreads();
  const int r32 = lane & 31, hh = lane >> 5;
  const int x = hh ^ (((r32 & 3) << 2) | ((r32 >> 2) & 3));
  int k_addr[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) k_addr[s] = 256 * r32 + 16 * ((2 * s) ^ x);
  f32x16 acc[4];
  f32x4 acc4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 4) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
      if (MODE == 4) __builtin_amdgcn_s_barrier();
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 64; ++i) acc4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i & 3], 0, 0, 0);
    } else {
      const bool active = (MODE < 6) || (((it & 1) == 0) == (wave < 4));
      if (MODE >= 8 && MODE != 9 && (it & 1) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stg[i] = gsrc[((it >> 1) * 2048 + i * 512 + tid) & 0xFFFFF];
      }
      if (!active && MODE >= 7) {
        // softmax-like block on this wave's own 32 accumulator values
        float mx = acc[0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(acc[0][r], acc[1][r]));
        float lsum = 0.f;
        bf16x8 pfl[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[kb][r] * 0.1f - mx * 0.1f;
            const float pv = (MODE == 9) ? v : __builtin_amdgcn_exp2f(v);
            pfl[2 * kb + (r >> 3)][r & 7] = (__bf16)pv;
            lsum += pv;
          }
        acc[2][0] += lsum;
        b = pfl[0]; a = pfl[1]; bsink = pfl[2]; bsink2 = pfl[3];
      }
      if (active) {
        bf16x8 kf[4][8];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int s = 0; s < 8; ++s) kf[g][s] = *reinterpret_cast<const bf16x8*>(smem + (g & 1) * 8192 + (g >> 1) * 16384 + k_addr[s]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int s = 0; s < 8; ++s) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g][s], b, acc[g], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int i = 0; i < 26; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      }
      if (MODE >= 8 && MODE != 9 && (it & 1) == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(smem + 32768 + ((i * 512 + tid) * 16)) = stg[i];
      }
      if (MODE >= 5) __builtin_amdgcn_s_barrier();
    }
  }
  float s = (float)bsink[0] + (float)bsink2[1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 4; ++r) s += acc4[i][r];
  }
  out[blockIdx.x * blockDim.x + tid] = s;
}

static u32x4* g_src;
template <int MODE>
void run(const char* name, int threads, int blocks, int iters, double mfma_per_wave_iter, double flop_per_mfma, const bf16x8* in, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 65536, 0, in, out, 16, g_src);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 65536, 0, in, out, iters, g_src);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  const double waves = (double)blocks * threads / 64;
  const double flops = waves * iters * mfma_per_wave_iter * flop_per_mfma;
  printf("%-70s %3d thr x %4d blk: %8.3f ms  %8.1f TFLOP/s\n", name, threads, blocks, ms, flops / ms / 1e9);
  fflush(stdout);
}

int main() {
  std::vector<unsigned short> h(512 * 8);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8* in; float* out;
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 4096 * 512 * 4);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&g_src, (size_t)(1 << 20) * 16 + 65536);
  hipMemset(g_src, 1, (size_t)(1 << 20) * 16 + 65536);
  const double F32 = 32.0 * 32 * 16 * 2, F16 = 16.0 * 16 * 32 * 2;
  const int it = 4000;
  run<0>("32x32x16, 4 accumulators, regs only, 1 wave/SIMD", 256, 256, it, 32, F32, in, out);
  run<0>("32x32x16, 4 accumulators, regs only, 2 waves/SIMD (512 thr)", 512, 256, it, 32, F32, in, out);
  run<0>("32x32x16, 4 accumulators, regs only, 2 waves/SIMD (2 blk/CU)", 256, 512, it, 32, F32, in, out);
  run<1>("32x32x16, ONE accumulator (dependent chain), 1 wave/SIMD", 256, 256, it, 32, F32, in, out);
  run<1>("32x32x16, ONE accumulator, 2 waves/SIMD", 512, 256, it, 32, F32, in, out);
  run<2>("16x16x32, 4 accumulators, regs only, 1 wave/SIMD", 256, 256, it, 64, F16, in, out);
  run<2>("16x16x32, 4 accumulators, regs only, 2 waves/SIMD", 512, 256, it, 64, F16, in, out);
  run<3>("32x32x16, A from LDS every MFMA (ds_read_b128, 6 ahead), 1 wave/SIMD", 256, 256, it, 32, F32, in, out);
  run<3>("32x32x16, A from LDS every MFMA, 2 waves/SIMD", 512, 256, it, 32, F32, in, out);
  run<4>("32x32x16 regs only + s_barrier / 32 MFMAs, 2 waves/SIMD", 512, 256, it, 32, F32, in, out);
  run<5>("32x32x16 A from LDS + s_barrier / 32 MFMAs, 2 waves/SIMD", 512, 256, it, 32, F32, in, out);
  run<6>("anti-phase halves: 32 LDS-fed MFMAs | idle, barrier, swap (512 thr)", 512, 256, it, 16, F32, in, out);
  run<6>("anti-phase halves, 228 workgroups", 512, 228, it, 16, F32, in, out);
  run<7>("anti-phase + softmax-like VALU block on the other half (32 exp)", 512, 256, it, 16, F32, in, out);
  run<9>("anti-phase + VALU block without exp", 512, 256, it, 16, F32, in, out);
  run<8>("anti-phase + VALU block + staging (4 gload + 4 ds_write_b128 / thread / 2 steps)", 512, 256, it, 16, F32, in, out);
  return 0;
}
