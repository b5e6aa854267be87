// Micro-benchmark: issue cost (cycles per wave-instruction on one SIMD) of the VALU ops in the attention softmax,
// alone and next to fp32 MFMAs.  hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(float *out, long long *cyc, int iters) {
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 16 independent v_exp_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
    } else if (MODE == 1) {   // 16 independent v_fma_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], 1.0001f, 0.5f);
    } else if (MODE == 2) {   // 16 v_max3-able
#pragma unroll
      for (int i = 0; i < 16; i += 2) a[i] = fmaxf(fmaxf(a[i], a[i + 1]), a[(i + 2) & 15]);
    } else if (MODE == 3) {   // 16 MFMA 16x16x4 f32, 4 chains
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[(i + 1) & 15], acc[i & 3], 0, 0, 0);
    } else if (MODE == 4) {   // 16 MFMA + 16 exp interleaved
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[(i + 1) & 15], acc[i & 3], 0, 0, 0);
        a[i] = __builtin_amdgcn_exp2f(a[i]);
      }
    } else if (MODE == 5) {   // 16 MFMA + 16 fma interleaved
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[(i + 1) & 15], acc[i & 3], 0, 0, 0);
        a[i] = fmaf(a[i], 1.0001f, 0.5f);
      }
    } else if (MODE == 6) {   // 16 MFMA + 48 fma interleaved
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[(i + 1) & 15], acc[i & 3], 0, 0, 0);
        a[i] = fmaf(a[i], 1.0001f, 0.5f);
        a[(i + 5) & 15] = fmaf(a[(i + 5) & 15], 1.0002f, 0.25f);
        a[(i + 9) & 15] = fmaf(a[(i + 9) & 15], 1.0003f, 0.125f);
      }
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_simd) {
  float *out; long long *cyc;
  int blocks = 256, threads = 256 * waves_per_simd;   // 4 SIMDs x waves_per_simd waves per block, one block per CU
  if (threads > 1024) { blocks *= threads / 1024; threads = 1024; }
  hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, sizeof(long long) * blocks);
  const int iters = 2000;
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256]; hipMemcpy(h, cyc, sizeof(long long) * 256, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
  printf("%-34s waves/SIMD=%d  cycles per iteration (16-op group) per wave: %8.1f\n", name, waves_per_simd, avg / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("16 v_exp_f32", w);
    run<1>("16 v_fma_f32", w);
    run<2>("8 v_max3 (16 fmax)", w);
    run<3>("16 mfma_16x16x4_f32", w);
    run<4>("16 mfma + 16 exp", w);
    run<5>("16 mfma + 16 fma", w);
    run<6>("16 mfma + 48 fma", w);
  }
  return 0;
}
