// Test helper (NOT part of the product library): a persistent "foreign" kernel that parks `workgroups` workgroups on the chip,
// each holding `lds_bytes` of LDS, until the host raises a stop word -- the stand-in for RCCL's channel kernels running on a side
// stream beside the fused training step.  With lds_bytes = 160 KB a workgroup owns its CU's
// whole LDS, so no LDS-using workgroup of the library can share that CU.
//   ctl[0]  stop word (host sets it to 1)          ctl[1]  workgroups that have started (device counts)
//   ctl[2]  workgroups that gave up on their own bound (the kernel never outlives `max_ms`)
// `ctl` is host-coherent pinned memory mapped into the device (hn_occupy_alloc).
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" {

__global__ __launch_bounds__(64) void hn_occupy_kernel(volatile unsigned *ctl, unsigned long long max_ticks) {
  extern __shared__ float park[];
  if (threadIdx.x == 0) {
    park[0] = 1.0f;                                  // (the allocation is used: not optimised away)
    __hip_atomic_fetch_add((unsigned *)ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load((unsigned *)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
      __builtin_amdgcn_s_sleep(64);
      if (__builtin_amdgcn_s_memrealtime() - t0 > max_ticks) {
        __hip_atomic_fetch_add((unsigned *)ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
}

// 64 bytes of mapped, coherent host memory: *host_out = host address, *dev_out = device address.  0 on success.
int hn_occupy_alloc(void **host_out, void **dev_out) {
  void *h = nullptr, *d = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return 1;
  for (int i = 0; i < 16; ++i) ((volatile unsigned *)h)[i] = 0;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return 2; }
  *host_out = h;
  *dev_out = d;
  return 0;
}
int hn_occupy_free(void *host) { return hipHostFree(host) == hipSuccess ? 0 : 1; }

// launch on `stream`; returns 0 on success.  max_ms bounds the kernel's life whatever the host does.
int hn_occupy_launch(void *ctl_dev, int workgroups, int lds_bytes, int max_ms, void *stream) {
  if (lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute((const void *)hn_occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return 1;
  hipLaunchKernelGGL(hn_occupy_kernel, dim3(workgroups), dim3(64), lds_bytes, (hipStream_t)stream, (volatile unsigned *)ctl_dev,
                     (unsigned long long)max_ms * 100000ull);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
