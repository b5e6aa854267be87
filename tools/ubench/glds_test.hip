// Semantics check of global_load_lds_dwordx4 issued from inline asm (saddr form, M0 = LDS byte address of the wave's
// destination, lane l lands at M0 + 16 l), two loads in flight, counted s_waitcnt.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const float *sbase, int voff_bytes, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff_bytes), "s"(lds_byte), "s"(sbase) : "memory");
}
__global__ void k(const float *src, float *dst) {
  __shared__ __attribute__((aligned(16))) float buf[2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned ldsbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)buf + wave * 2048);
  // source-side swizzle: lane l = (row r = l >> 3, position p = l & 7) fetches chunk p ^ (r & 7) of row r (rows of 32 floats)
  const int r = lane >> 3, p = lane & 7;
  const int voff = ((wave * 16 + r) * 32 + ((p ^ (r & 7)) * 4)) * 4;
  glds16(src, voff, ldsbase);
  glds16(src, voff + 8 * 32 * 4, ldsbase + 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // read back row fi (0..15), logical chunk c: position c ^ (fi & 7)
  const int fi = lane & 15, c = lane >> 4;
  f32x4 v = *(f32x4 *)&buf[wave * 512 + fi * 32 + ((c ^ (fi & 7)) * 4)];
  dst[threadIdx.x] = v.x;      // expect src[(wave*16 + fi)*32 + c*4]
}
int main() {
  float *s, *d; (void)hipMalloc(&s, 4 * 4096); (void)hipMalloc(&d, 4 * 128);
  static float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i;
  (void)hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, s, d);
  float o[128]; (void)hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 128; ++t) { const int w = t >> 6, l = t & 63, fi = l & 15, c = l >> 4; if (o[t] != (float)((w * 16 + fi) * 32 + c * 4)) bad++; }
  printf("glds test: bad=%d of 128 (o[17]=%g)\n", bad, o[17]);
  return bad != 0;
}
