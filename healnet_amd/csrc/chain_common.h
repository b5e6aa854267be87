// Shared device-side pieces of the latent chain kernels (chain.hip: forward; bchain.hip: backward): tile geometry, explicit
// address-space accessors, the weight-stream helpers.  Included inside namespace hn { namespace { ... } } of each unit.
#pragma once
constexpr int CR = 16;                  // rows per workgroup
constexpr int CD = 128;                 // latent width
constexpr int CHID = 512;               // feed-forward hidden width (4 * CD)
constexpr int WN = 128, WK = 32;        // weight block: 128 output columns x 32 k
constexpr int WBLK = WN * WK;           // floats per block (16 KB)
constexpr int ATILE = CR * WK;          // floats per A k-tile (16 rows x 32 k, 16-byte slots XOR-swizzled by row & 7)
constexpr int XP = 132;                 // pitch of the x tile
constexpr int WSLOT = 16 * WK;          // floats per wave and block: 16 weight rows x 32 k (2 KB = two wave-wide 16-byte loads)

// ---- cluster grids (small batches: C = 2 or 4 workgroups share a row tile and exchange partial tiles through the L2) ----
// Workgroup index -> (member, tile).  Workgroups are dealt round-robin over the 8 XCDs and, on each XCD, dispatched in index
// order.  A group of 8 C consecutive workgroups holds 8 tiles: index = group * 8 C + member * 8 + tile % 8, so the C members of
// a tile share an XCD (same index mod 8: one L2 for the exchange) and are CONSECUTIVE in that XCD's dispatch order.  The
// exchange spins on the other members' flags, and nothing in HIP promises that all workgroups of a grid are resident at once (a
// CU mask, another process, an RCCL kernel parked on some CUs): with this order the oldest unfinished tile of an XCD always has
// its members resident or next in line, so the grid makes progress whenever an XCD can hold C workgroups -- the former order
// (members `tiles` workgroups apart) deadlocked as soon as the LAST member row did not fit beside the others.
// `split_order` (HN_FORCE_CLUSTER_SPLIT_ORDER=1, the A/B switch of tests/test_gpu_cluster.py) selects that former order.
__device__ __forceinline__ void cluster_decode(int bid, int C, int ntiles, int split_order, int &member, int &tile) {
  if (split_order) {
    member = bid / ntiles;
    tile = bid - member * ntiles;
    return;
  }
  const int per = 8 * C, g = bid / per, r = bid - g * per;
  member = r >> 3;
  tile = g * 8 + (r & 7);
}
// A field of the by-value argument struct read from the kernel-argument segment WHERE IT IS USED (volatile: not hoisted).  The cluster
// fields are touched once per launch, in the exchange; loaded at kernel entry like the others they stayed live across the whole weight
// stream and pushed 17 / 44 SGPRs of the chain kernels into spills (round 5).
template <typename T>
__device__ __forceinline__ T late_kernarg(size_t offset) {
  return *(const volatile T __attribute__((address_space(4))) *)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}

// Wait (one lane per member flag) until `flag` reaches `seq`, for at most `ticks` of the 100 MHz s_memrealtime clock.  A member
// that never shows up must neither hang the device nor let the tile carry on with an incomplete sum: the caller turns the tile
// into NaN, and the loss is REPORTED -- the launch's marker word in the workspace, and the launch's token in the host-mapped
// status word of the device (hn_cluster_status; the next entry point returns HN_E_CORESIDENCY, hn_l1_adam_step skips).
__device__ __forceinline__ int cluster_wait(const int *flag, int seq, unsigned ticks, int *marker, unsigned *status, unsigned token) {
  if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= seq) return 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
    __builtin_amdgcn_s_sleep(2);
    if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)ticks) {
      *marker = 1;
      if (status) __hip_atomic_store(status, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return 1;
    }
  }
  return 0;
}
// Pointers that arrive inside the argument struct are generic: hipcc emits flat_load / flat_store for them, and with flat
// operations in flight its wait-count insertion falls back to "wait for everything" in front of every use of a prefetched
// register.  Everything outside the weight stream therefore goes through explicit global-address-space accesses.
// buffer_load_dwordx4 ... idxen offen: address = base + vindex * stride + voffset, base and stride in the SGPR descriptor
__device__ f32x4 hn_sbuffer_load_x4(i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");
typedef float __attribute__((address_space(1))) gf32;
typedef f32x4 __attribute__((address_space(1))) gf32x4;
__device__ __forceinline__ float4 gld4(const gf32 *p) {
  const f32x4 v = *(const gf32x4 *)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float gld1(const gf32 *p) { return *p; }
__device__ __forceinline__ void gst4(gf32 *p, const float4 &v) {
  f32x4 t = {v.x, v.y, v.z, v.w};
  *(gf32x4 *)p = t;
}
__device__ __forceinline__ void gst1(gf32 *p, float v) { *p = v; }
// streaming store: written through instead of staying dirty in this XCD's L2 until the end-of-kernel write-back
__device__ __forceinline__ void gst4_nt(gf32 *p, const float4 &v) {
  f32x4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, (gf32x4 *)p);
}
// ... and every LDS access through address space 3 with integer offsets (a generic pointer that the compiler cannot trace back
// to the LDS symbol becomes a flat access, which counts against BOTH wait counters)
typedef float __attribute__((address_space(3))) lf32;
typedef f32x4 __attribute__((address_space(3))) lf32x4;
__device__ __forceinline__ float4 lld4(const lf32 *base, int off) {
  const f32x4 v = *(const lf32x4 *)(base + off);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lst4(lf32 *base, int off, const float4 &v) {
  f32x4 t = {v.x, v.y, v.z, v.w};
  *(lf32x4 *)(base + off) = t;
}

// SELU with expm1(x) as exp(x) - 1 (v_exp_f32): the absolute error is half an ulp of 1 (6e-8) for any x <= 0 -- what matters
// here, since the result scales a value of O(1) that is summed into 512-term dot products; the RELATIVE accuracy that expm1
// buys for |x| << 1 is below the rounding of those sums.  ocml's expm1f is ~40 instructions, a split polynomial form ~16; this
// is 6, in an epilogue that runs once per 8 blocks on the VALU the MFMAs share.
__device__ __forceinline__ float selu_f(float x) {
  const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
  // (v_exp_f32 on x * log2(e) directly: __expf wraps it in a denormal-range rescale -- v_cmp / v_cndmask / v_ldexp per element -- that
  // only matters below e^-87, where e - 1 is -1 either way; the epilogue is VALU time the fp32 MFMAs of the chunk cannot overlap)
#ifdef HN_SELU_EXPF      // (A/B of the epilogue: tools/lchain_profile.py with HN_PROF_EXTRA=-DHN_SELU_EXPF)
  const float e = __expf(fminf(x, 0.0f)) - 1.0f;
#else
  const float e = __builtin_amdgcn_exp2f(fminf(x, 0.0f) * 1.44269504088896340736f) - 1.0f;
#endif
  return scale * (x > 0.0f ? x : alpha * e);
}
// sum over the 32 lanes that share a row of the x tile (one half of a wave), delivered to all of them: four DPP steps inside each
// row of 16 lanes and one swizzle across the two rows (__shfl_xor is a ds_bpermute each: ten of them sat in every LayerNorm)
#define CH_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, true))
__device__ __forceinline__ float half_wave_sum(float v) {
  v += CH_DPP(v, 0xB1);                       // quad_perm [1,0,3,2]
  v += CH_DPP(v, 0x4E);                       // quad_perm [2,3,0,1]
  v += CH_DPP(v, 0x141);                      // row_half_mirror
  v += CH_DPP(v, 0x140);                      // row_mirror
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));      // lane ^ 16 (bit mode: and 0x1f, xor 0x10)
  return v;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

