// Do fp32 MFMAs and an L2-resident weight stream overlap on a CU?  256 workgroups x 8 waves walk the same 2 MB buffer in 16 KB
// blocks (wave w: two full-line 16-byte loads per lane = 2 KB per block, 4-deep register ring) and issue NM v_mfma_f32_16x16x4
// per block and wave on the loaded values.  The latent chain's step is 8 MFMAs per wave against 16 KB per CU.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_stream mfma_stream.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, bool LOADS>
__global__ __launch_bounds__(512) void k(const float *__restrict__ W, float *__restrict__ out, int nblk, int reps) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
  f32x4 B[4][2];
  const float *base = W + wave * 512 + lane * 4;              // block b: W + b * 4096 floats; wave's 2 KB: two 1 KB pieces
  auto issue = [&](f32x4 (&r)[2], int b) {
    if (LOADS) { const float *p = base + (long)(b % nblk) * 4096; r[0] = *(const f32x4 *)p; r[1] = *(const f32x4 *)(p + 256); }
  };
  const float a = (float)lane;
#pragma unroll
  for (int i = 0; i < 4; ++i) { B[i][0] = c0; B[i][1] = c0; issue(B[i], i); }
  for (int r = 0; r < reps; ++r) {
    for (int b = 0; b < nblk; b += 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 v0 = B[i][0], v1 = B[i][1];
#pragma unroll
        for (int m = 0; m < NM / 2; ++m) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v0[m & 3], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v1[m & 3], c1, 0, 0, 0);
        }
        issue(B[i], b + i + 4);
      }
    }
  }
  if (c0.x + c1.y == 1.2345e30f) out[0] = c0.x;
}

template <int NM, bool LOADS> void run(const float *W, float *out, const char *what) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nblk = 128, reps = 16;                            // 2 MB, 2048 blocks in total
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, LOADS>), dim3(256), dim3(512), 0, 0, W, out, nblk, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-34s %.3f us per block  (%.1f GB/s per CU, %.1f TF/s)\n", what, ms * 1e3 / (nblk * reps),
         LOADS ? 16384.0 * nblk * reps / (ms * 1e-3) / 1e9 : 0.0, 2048.0 * NM * 8 * 256 * nblk * reps / (ms * 1e-3) / 1e12);
}

int main() {
  float *W, *out; (void)hipMalloc(&W, 2 << 20); (void)hipMalloc(&out, 64); (void)hipMemset(W, 0, 2 << 20);
  run<8, false>(W, out, "8 MFMAs per wave, no loads");
  run<0, true>(W, out, "loads only");
  run<8, true>(W, out, "8 MFMAs per wave + loads");
  run<16, true>(W, out, "16 MFMAs per wave + loads");
  run<4, true>(W, out, "4 MFMAs per wave + loads");
  return 0;
}
