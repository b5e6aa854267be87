// Do matrix and vector instructions of ONE SIMD overlap on gfx950?  (the fp32 attention core behaves as if they
// do not; is the bf16 MFMA different?)  Each wave runs a loop of NM independent MFMAs and NV v_exp_f32 per iteration; the kernel
// is timed with MFMAs only, exponentials only and both, at 1 .. 4 waves per SIMD (one workgroup of 4 * W waves per CU).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, bool DO_M, bool DO_V, bool INTERLEAVE = false>      // KIND 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_32x32x16_bf16
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
  f32x4 acc[8];
  float e[16];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 16; ++i) e[i] = 0.001f * (threadIdx.x + i);
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  f32x4 aw = {a, a, a, a}, bw = {b, b, b, b};
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 big[2];
  for (int i = 0; i < 16; ++i) { big[0][i] = 0.f; big[1][i] = 0.f; }
  for (int it = 0; it < iters; ++it) {
    if (INTERLEAVE) {          // one MFMA, then its share of the vector work, pinned: MFMA, 2 x (exp, mul), MFMA, ...
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        else if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aw), __builtin_bit_cast(bf16x8, bw), acc[i], 0, 0, 0);
        else if ((i & 1) == 0) big[(i >> 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw), __builtin_bit_cast(bf16x8, bw), big[(i >> 1) & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        e[2 * i] = __builtin_amdgcn_exp2f(e[2 * i]) * 0.25f;
        e[2 * i + 1] = __builtin_amdgcn_exp2f(e[2 * i + 1]) * 0.25f;
        __builtin_amdgcn_sched_barrier(0);
      }
      continue;
    }
    if (DO_M) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aw), __builtin_bit_cast(bf16x8, bw), acc[i], 0, 0, 0);
      }
    }
    if (DO_V) {
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]) * 0.25f;     // v_exp_f32 + v_mul (dependent chain per i, 16 chains)
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < 16; ++i) s += big[0][i] + big[1][i];
  for (int i = 0; i < 16; ++i) s += e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, bool M, bool V, bool IL = false>
float run(int waves_per_simd, float *d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256), block(256 * waves_per_simd);
  hipLaunchKernelGGL((k<KIND, M, V, IL>), grid, block, 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, M, V, IL>), grid, block, 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / iters;       // ns per iteration
}

int main() {
  float *d;
  hipMalloc(&d, 256 * 1024 * sizeof(float));
  printf("ns per loop iteration (8 MFMAs and / or 16 v_exp_f32 + 16 v_mul per wave)\n");
  for (int w = 1; w <= 4; ++w) {
    const float m0 = run<0, true, false>(w, d), v = run<0, false, true>(w, d), b0 = run<0, true, true>(w, d);
    const float m1 = run<1, true, false>(w, d), b1 = run<1, true, true>(w, d);
    const float i0 = run<0, true, true, true>(w, d), i1 = run<1, true, true, true>(w, d), i2 = run<2, true, true, true>(w, d);
    printf("   interleaved (MFMA, 2 exp + 2 mul, MFMA, ...): fp32 16x16x4 %.1f | bf16 16x16x32 %.1f | 4 x bf16 32x32x16 + the same vector work %.1f\n", i0, i1, i2);
    printf("waves/SIMD %d | fp32 16x16x4: mfma %.1f  valu %.1f  both %.1f (sum %.1f, max %.1f) | bf16 16x16x32: mfma %.1f  both %.1f (sum %.1f, max %.1f)\n",
           w, m0, v, b0, m0 + v, m0 > v ? m0 : v, m1, b1, m1 + v, m1 > v ? m1 : v);
  }
  return 0;
}
