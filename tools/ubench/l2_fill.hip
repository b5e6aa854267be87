// How fast can every CU pull the SAME L2-resident buffer (the weight stream of the latent chain: 1-2 MB read by all 256
// workgroups)?  One 512-thread workgroup per CU streams `bytes` with 16-byte loads, `depth` loads in flight per thread.
//   mode 0: all workgroups read the same buffer in the same order      (the chain's pattern)
//   mode 1: same buffer, every workgroup starts at a different offset  (rotated)
//   mode 2: every workgroup reads its own private slice                (no sharing; total = nwg * bytes)
// hipcc --offload-arch=gfx950 -O3 -o l2_fill l2_fill.hip ; ./l2_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const f32x4 *__restrict__ src, float *__restrict__ out, long vec_per_wg, int mode, int reps) {
  const int tid = threadIdx.x;
  const f32x4 *base = src;
  long start = 0;
  if (mode == 1) start = ((long)blockIdx.x * 7919 * 512) % vec_per_wg;
  if (mode == 2) base = src + (long)blockIdx.x * vec_per_wg;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < reps; ++r) {
    for (long i = 0; i < vec_per_wg; i += 512 * DEPTH) {
      f32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        long idx = start + i + (long)d * 512 + tid;
        if (idx >= vec_per_wg) idx -= vec_per_wg;
        v[d] = base[idx];
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) out[0] = acc.x;
}

// mode 3: the chain's block walk over a (N x K) row-major weight: blocks of 128 rows x 32 k (16 KB); wave w takes rows 16 w .. + 15
// of the block as two 1 KB pieces (8 rows x 128 B, row stride 4 K bytes), DEPTH blocks in flight; all workgroups walk in lockstep.
template <int DEPTH>
__global__ __launch_bounds__(512) void block_walk_kernel(const float *__restrict__ W, float *__restrict__ out, int N, int K, int reps) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r8 = lane >> 3, p = lane & 7;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nk = K / 32, nch = N / 128, nblk = nk * nch;
  for (int r = 0; r < reps; ++r) {
    for (int b0 = 0; b0 < nblk; b0 += DEPTH) {
      f32x4 v[DEPTH][2];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int bi = min(b0 + d, nblk - 1), j = bi / nk, k = bi % nk;
        const float *src = W + (long)(j * 128 + wave * 16 + r8) * K + k * 32 + p * 4;
        v[d][0] = *(const f32x4 *)src;
        v[d][1] = *(const f32x4 *)(src + 8 * K);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d][0] + v[d][1];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) out[0] = acc.x;
}

// mode 4: the same walk with the v2 fragment pattern: 16 rows x 64 B per load (lane = (row 0..15, 16-byte quarter)), 2 loads per block
template <int DEPTH>
__global__ __launch_bounds__(512) void half_line_kernel(const float *__restrict__ W, float *__restrict__ out, int N, int K, int reps) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fi = lane & 15, fg = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nk = K / 32, nch = N / 128, nblk = nk * nch;
  for (int r = 0; r < reps; ++r) {
    for (int b0 = 0; b0 < nblk; b0 += DEPTH) {
      f32x4 v[DEPTH][2];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int bi = min(b0 + d, nblk - 1), j = bi / nk, k = bi % nk;
        const float *src = W + (long)(j * 128 + wave * 16 + fi) * K + k * 32 + fg * 4;
        v[d][0] = *(const f32x4 *)src;
        v[d][1] = *(const f32x4 *)(src + 16);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d][0] + v[d][1];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) out[0] = acc.x;
}
// mode 5: full-line walk through global_load_lds_dwordx4 into a per-wave LDS ring (the v3 path), counted vmcnt, ds_read back
__device__ __forceinline__ void glds16(const void *sbase, int voff, unsigned m0v) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(sbase) : "memory");
}
__global__ __launch_bounds__(512) void glds_walk_kernel(const float *__restrict__ W, float *__restrict__ out, int N, int K, int reps) {
  extern __shared__ __attribute__((aligned(16))) float ring[];        // [8 waves][6 slots][512 floats]
  const int tid = threadIdx.x, lane = tid & 63, r8 = lane >> 3, p = lane & 7;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned ring_byte = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)ring + wave * 6 * 2048);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nk = K / 32, nch = N / 128, nblk = nk * nch;
  int lb = 0, ws = 0, rs = 0;
  auto issue = [&]() {
    const int bi = min(lb, nblk - 1), j = bi / nk, k = bi % nk;
    ++lb;
    const float *sb = W + (long)(j * 128 + wave * 16) * K + k * 32;
    const int voff = (r8 * K + p * 4) * 4;
    glds16(sb, voff, ring_byte + ws * 2048);
    glds16(sb, voff + 8 * K * 4, ring_byte + ws * 2048 + 1024);
    ws = ws + 1 == 6 ? 0 : ws + 1;
  };
  for (int r = 0; r < reps; ++r) {
    lb = 0;
    for (int i = 0; i < 5; ++i) issue();
    for (int b = 0; b < nblk; ++b) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      const f32x4 v0 = *(const f32x4 *)&ring[(wave * 6 + rs) * 512 + lane * 4], v1 = *(const f32x4 *)&ring[(wave * 6 + rs) * 512 + 256 + lane * 4];
      rs = rs + 1 == 6 ? 0 : rs + 1;
      acc += v0 + v1;
      issue();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ws = 0; rs = 0;
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) out[0] = acc.x;
}

int main() {
  const int nwg = 256;
  const long bytes = 2 << 20;                 // per workgroup
  const long vec = bytes / 16;
  f32x4 *src; float *out;
  (void)hipMalloc(&src, (size_t)bytes * nwg); (void)hipMalloc(&out, 64);
  (void)hipMemset(src, 0, (size_t)bytes * nwg);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int reps = 8;
  for (int mode = 0; mode < 3; ++mode) {
    for (int depth : {2, 4, 8}) {
      for (int it = 0; it < 2; ++it) {
        (void)hipEventRecord(e0);
        if (depth == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(nwg), dim3(512), 0, 0, src, out, vec, mode, reps);
        if (depth == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(nwg), dim3(512), 0, 0, src, out, vec, mode, reps);
        if (depth == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(nwg), dim3(512), 0, 0, src, out, vec, mode, reps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it == 1) printf("mode %d depth %d: %.1f us per pass, %.2f TB/s aggregate, %.1f GB/s per CU\n", mode, depth, ms * 1e3 / reps,
                            (double)bytes * nwg * reps / (ms * 1e-3) / 1e12, (double)bytes * reps / (ms * 1e-3) / 1e9);
      }
    }
  }
  for (int K : {128, 512}) {
    const int N = K == 128 ? 4096 : 1024;           // 2 MB either way
    for (int depth : {2, 4, 8}) {
      for (int it = 0; it < 2; ++it) {
        (void)hipEventRecord(e0);
        if (depth == 2) hipLaunchKernelGGL(block_walk_kernel<2>, dim3(nwg), dim3(512), 0, 0, (const float *)src, out, N, K, reps);
        if (depth == 4) hipLaunchKernelGGL(block_walk_kernel<4>, dim3(nwg), dim3(512), 0, 0, (const float *)src, out, N, K, reps);
        if (depth == 8) hipLaunchKernelGGL(block_walk_kernel<8>, dim3(nwg), dim3(512), 0, 0, (const float *)src, out, N, K, reps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it == 1) printf("block walk K=%d depth %d: %.1f us per pass, %.2f TB/s aggregate, %.1f GB/s per CU\n", K, depth, ms * 1e3 / reps,
                            (double)bytes * nwg * reps / (ms * 1e-3) / 1e12, (double)bytes * reps / (ms * 1e-3) / 1e9);
      }
    }
  }
  for (int K : {128, 512}) {
    const int N = K == 128 ? 4096 : 1024;
    for (int it = 0; it < 2; ++it) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(half_line_kernel<4>, dim3(nwg), dim3(512), 0, 0, (const float *)src, out, N, K, reps);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (it == 1) printf("half-line walk K=%d depth 4: %.1f us per pass, %.1f GB/s per CU\n", K, ms * 1e3 / reps, (double)bytes * reps / (ms * 1e-3) / 1e9);
    }
    (void)hipFuncSetAttribute((const void *)glds_walk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 6 * 2048);
    for (int it = 0; it < 2; ++it) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(glds_walk_kernel, dim3(nwg), dim3(512), 8 * 6 * 2048, 0, (const float *)src, out, N, K, reps);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (it == 1) printf("glds walk K=%d (5 ahead): %.1f us per pass, %.1f GB/s per CU\n", K, ms * 1e3 / reps, (double)bytes * reps / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
