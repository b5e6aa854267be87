// Do fp32 MFMAs and an L2-resident weight stream overlap on a CU?  256 workgroups x 8 waves walk the same 2 MB buffer in 16 KB
// blocks (wave w: two full-line 16-byte loads per lane = 2 KB per block, 4-deep register ring) and issue NM v_mfma_f32_16x16x4
// per block and wave on the loaded values.  The latent chain's step is 8 MFMAs per wave against 16 KB per CU.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_stream mfma_stream.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 buffer_load_x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
#ifndef BUFFER_LOADS
#define BUFFER_LOADS 0          // 1: buffer_load_dwordx4 v, v_off, s[rsrc], s_off (one address VGPR, block base in an SGPR)
#endif

// WAVES = 16: two groups of 8 waves take alternate blocks (four waves per SIMD instead of two, same work per CU)
template <int NM, bool LOADS, int WAVES = 8>
__global__ __launch_bounds__(WAVES * 64) void k(const float *__restrict__ W, float *__restrict__ out, int nblk, int reps) {
  const int tid = threadIdx.x, wave = (tid >> 6) & 7, lane = tid & 63, grp = tid >> 9;
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
  f32x4 B[4][2];
  const float *base = W + wave * 512 + lane * 4;              // block b: W + b * 4096 floats; wave's 2 KB: two 1 KB pieces
  i32x4 rs;
  {
    const unsigned long long a = (unsigned long long)W;
    rs.x = (int)(a & 0xffffffffu); rs.y = (int)((a >> 32) & 0xffffu); rs.z = 2 << 20; rs.w = 0x00020000;
  }
  const int voff = (wave * 512 + lane * 4) * 4;
  auto issue = [&](f32x4 (&r)[2], int b) {
    if (LOADS && BUFFER_LOADS) {
      const int so = ((WAVES == 16 ? 2 * b + grp : b) % nblk) * 16384;
      r[0] = buffer_load_x4(rs, voff, so, 0); r[1] = buffer_load_x4(rs, voff + 1024, so, 0);
    } else if (LOADS) { const float *p = base + (long)((WAVES == 16 ? 2 * b + grp : b) % nblk) * 4096; r[0] = *(const f32x4 *)p; r[1] = *(const f32x4 *)(p + 256); }
  };
  const float a = (float)lane;
#pragma unroll
  for (int i = 0; i < 4; ++i) { B[i][0] = c0; B[i][1] = c0; issue(B[i], i); }
  for (int r = 0; r < reps; ++r) {
    for (int b = 0; b < (WAVES == 16 ? nblk / 2 : nblk); b += 4) {
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

template <int NM, bool LOADS, int WAVES = 8> void run(const float *W, float *out, const char *what) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nblk = 128, reps = 16;                            // 2 MB, 2048 blocks in total
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, LOADS, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, W, out, nblk, reps);
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
  run<8, false, 16>(W, out, "16 waves: 8 MFMAs, no loads");
  run<8, true, 16>(W, out, "16 waves: 8 MFMAs + loads");
  return 0;
}
