// Internal helpers shared by the HIP translation units of libhealnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>

#include "../../include/healnet_hip.h"

namespace hn {
// development aid (HN_POISON_LDS=1): after every launch fill the LDS of every CU with NaNs on the same stream, so a kernel
// that reads LDS it has not written fails deterministically instead of inheriting benign leftovers (misc.hip)
void debug_after_launch(hipStream_t s);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Environment knobs.  ROUTE switches (HN_NO_* / HN_FORCE_*: take the generic route of a fused kernel, or force a fused one below its
// size gate -- the A/B tests compare the two routes) are read in every build.  TUNING knobs, which change a kernel's GEOMETRY (query
// tiles per wave, resident-wave targets, split sizes, cluster limits), exist only in a build made with -DHN_TUNING_KNOBS
// (HN_EXTRA_HIPCC_FLAGS=-DHN_TUNING_KNOBS, the sweep tools under tools/): a product build cannot be steered into an untested
// geometry from the environment.
#ifdef HN_TUNING_KNOBS
static inline const char *tuning_env(const char *name) { return getenv(name); }
#else
static inline const char *tuning_env(const char *) { return nullptr; }
#endif

// hn_set_kernel_timers (api_blocks.hip): brackets the launches of a named kernel class with the caller's event pairs
struct KernelTimerScope {
  hipEvent_t stop;
  hipStream_t s;
  KernelTimerScope(const char *kernel, hipStream_t stream);
  ~KernelTimerScope() { if (stop) (void)hipEventRecord(stop, s); }
};

// thread-local error string behind hn_last_error_string()
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define HN_HIP_CHECK(expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) return ::hn::fail(HN_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

#define HN_LAUNCH_CHECK(name)                                                                    \
  do {                                                                                           \
    hipError_t _e = hipGetLastError();                                                           \
    if (_e != hipSuccess) return ::hn::fail(HN_E_HIP, "launch of %s failed: %s", name, hipGetErrorString(_e)); \
    ::hn::debug_after_launch(s);                                                                 \
  } while (0)

#define HN_REQUIRE(cond, code, ...)                                                              \
  do {                                                                                           \
    if (!(cond)) return ::hn::fail(code, __VA_ARGS__);                                           \
  } while (0)

// Raw buffer loads through an SGPR buffer descriptor (V#), bound to the LLVM intrinsics by name (in this toolchain,
// ROCm 7.2, the __builtin_amdgcn_raw_buffer_load_b128 builtin is lowered to a single dword load).
// Why every guarded operand load in this library goes through them: a "load or 0" on a per-lane predicate makes
// hipcc branch around the load and wait vmcnt(0) right behind it, i.e. one serialized memory round trip per
// element (and it re-sinks the load into the branch even when the source loads unconditionally and selects).
// With a descriptor whose num_records ends at the last valid row, out-of-range rows simply read as 0 in
// hardware: no predicate, no branch, all loads of a tile in flight together.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 hn_buffer_load_x4_raw(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float hn_buffer_load_x1_raw(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
#ifndef HN_LOAD_AUX
#define HN_LOAD_AUX 0      // cache-policy bits of every buffer load (development: -DHN_LOAD_AUX=17 = sc0 sc1, system scope)
#endif
__device__ __forceinline__ f32x4 hn_buffer_load_x4(i32x4 rsrc, int voffset, int soffset, int) {
  return hn_buffer_load_x4_raw(rsrc, voffset, soffset, HN_LOAD_AUX);
}
__device__ __forceinline__ float hn_buffer_load_x1(i32x4 rsrc, int voffset, int soffset, int) {
  return hn_buffer_load_x1_raw(rsrc, voffset, soffset, HN_LOAD_AUX);
}
__device__ __forceinline__ i32x4 make_rsrc(const void *base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = (int)(a & 0xffffffffu);
  r.y = (int)((a >> 32) & 0xffffu);   // stride 0: raw buffer, byte-offset range check against num_records
  r.z = (int)bytes;
  r.w = 0x00020000;                   // DATA_FORMAT = 32-bit
  return r;
}
__device__ __forceinline__ float4 buf4(i32x4 rsrc, int byte_off) {
  const f32x4 v = hn_buffer_load_x4(rsrc, byte_off, 0, 0);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 buf4s(i32x4 rsrc, int byte_off, int sgpr_off) {
  const f32x4 v = hn_buffer_load_x4(rsrc, byte_off, sgpr_off, 0);
  return make_float4(v.x, v.y, v.z, v.w);
}
__host__ __device__ static inline unsigned rsrc_bytes(long rows, long ld, long width) { return (unsigned)(((rows - 1) * ld + width) * 4); }

// Packed context layout of the fused forward (rank-D binding, LayerNorm-ed context).  A normalised row sums to
// zero, so the last of its D channels is redundant: s = sum_{d<D} q_d z_d = sum_{d<D-1} (q_d - q_{D-1}) z_d.  The
// D-1 kept channels are stored in the columns that the first `ks` k-steps of the QK^T MFMA chain read
// (step c of 16-column block s touches columns 16 s + 4 g + c), so the chain runs ks = ceil((D-1)/4) steps
// instead of dp/4: 3 instead of 4 for an RGB image (D = 13), 5 instead of 8 for a volume (D = 18).
__host__ __device__ constexpr int packed_slot(int c, int ks) {
  return ks <= 4 ? (c / ks) * 4 + c % ks
                 : (c < 16 ? c : 16 + ((c - 16) / (ks - 4)) * 4 + (c - 16) % (ks - 4));
}
// inverse: kept channel stored in column `slot`, or -1 for an unused column
__host__ __device__ constexpr int packed_chan(int slot, int ks) {
  return ks <= 4 ? ((slot & 3) < ks ? (slot >> 2) * ks + (slot & 3) : -1)
                 : (slot < 16 ? slot : (((slot - 16) & 3) < ks - 4 ? 16 + ((slot - 16) >> 2) * (ks - 4) + ((slot - 16) & 3) : -1));
}
static inline int packed_steps(int D, int dp) {           // 0: keep the natural layout
  const int ks = (D - 1 + 3) / 4;
  return (D >= 2 && D <= dp - 1 && ks >= 1 && ks < dp / 4) ? ks : 0;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

// bump allocator over the caller-provided workspace
struct Arena {
  char *base;
  size_t size, off;
  bool overflow;
  Arena(void *p, size_t n) : base((char *)p), size(n), off(0), overflow(false) {}
  template <typename T> T *take(size_t count) {
    size_t bytes = align_up(count * sizeof(T), 256);
    if (base == nullptr) { off += bytes; return nullptr; }   // sizing pass
    if (off + bytes > size) { overflow = true; off += bytes; return nullptr; }
    T *r = (T *)(base + off);
    off += bytes;
    return r;
  }
};

// ------------------------------------------------------------------------------------------------
// generic fused GEMM:  C[m, col(n)] = act( alpha * sum_k pro(A[m,k]) * W[n,k] + bias[n] ) (+ R[m,n])
// ------------------------------------------------------------------------------------------------
enum { PRO_NONE = 0, PRO_AFFINE = 1, PRO_LAYERNORM = 2 };
enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_GLU_SELU = 2, ACT_GLU_GELU = 3 };

// ------------------------------------------------------------------------------------------------
// Dropout (nn.Dropout at healnet/models/healnet.py:381,421 on the attention probabilities and :347 on
// the feed-forward output).  Counter-based: the keep decision of an element is a pure function of
// (seed, offset, stream id of the block, row, column), so the forward core, both backward kernels and the mask export
// used by the tests regenerate identical masks whatever their tiling.  Philox4x32 with 7 rounds (the shortest variant that
// passes BigCrush in Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- 10 is the library default's safety
// margin, and inside the attention core every round is paid per score quad); one call yields the four decisions of an aligned
// column quad (row, 4 c .. 4 c + 3): keep iff word >= thr, P(keep) = 1 - p.
// ATTENTION masks (stream ids without DROP_SID_FF) take 16-bit decisions, SIXTEEN per call (round 4; round 3 took eight): the
// call of (row R with bits 4 and 5 clear, quad c) decides the column quad of the four rows R, R + 16, R + 32, R + 48 -- a lane's
// query tiles in the cores are 16 rows apart, so the four tiles of a wave share ONE call.  Row R + 16 k reads, from word r of the
// call (column 4 c + r), the 16-bit WINDOW that starts at byte k of the word, cyclically: bytes (1,0), (2,1), (3,2), (0,3) for
// k = 0 .. 3, high byte first; keep iff window >= thr, thr = p * 2^16 rounded down, kept values scaled by 1 / (1 - p) as nn.Dropout
// does.  Every window is a uniform 16-bit number, so every decision has P(keep) = 1 - thr / 2^16 EXACTLY (the rate keeps its 2^-16
// resolution -- 8-bit decisions would have quantised it to 1/256); the windows of rows 16 apart share one byte, as the LOW byte
// of one and the HIGH byte of the other: decision k looks at its low byte only when its high byte equals the threshold's (1 in
// 256), so the covariance of the two decisions is <= 2^-8 / 4 (correlation <= 0.5 %, exactly 0 when thr is a multiple of 256,
// e.g. p = 0.25 / 0.5); columns, other rows, blocks, offsets and seeds stay independent calls.  tests/test_gpu_dropout.py bounds
// it.  Inside the image cores the generator is what dropout costs (~55 vector instructions per call against the 28 MFMAs of a
// 16-token step): one call per lane and step instead of round 3's two (round 2: four).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t DROP_SID_FF = 0x80000000u;      // stream ids of feed-forward blocks carry this bit
struct DropCfg {
  uint32_t thr;        // feed-forward: p * 2^32; attention: p * 2^16; 0 = dropout disabled
  float scale;         // 1 / (1 - p)
  uint32_t seed_lo, seed_hi, sid, offset;
  const uint32_t *offset_dev;   // hn_rng.offset_dev: added to `offset` on the device (graph replays), or NULL
};
// the counter word every Philox call of a launch uses (a wave-uniform scalar load when the device word is set)
__device__ __forceinline__ uint32_t drop_counter(const DropCfg &d) { return d.offset_dev ? d.offset + *d.offset_dev : d.offset; }
static inline DropCfg make_drop(float p, uint64_t seed, uint32_t offset, uint32_t sid, const uint32_t *offset_dev = nullptr) {
  DropCfg d;
  d.thr = 0; d.scale = 1.0f; d.seed_lo = (uint32_t)seed; d.seed_hi = (uint32_t)(seed >> 32); d.sid = sid; d.offset = offset;
  d.offset_dev = offset_dev;
  if (p > 0.0f && (sid & DROP_SID_FF)) {
    double t = (double)p * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    if (d.thr == 0) d.thr = 1;
    d.scale = (float)(1.0 / (1.0 - (double)p));
  } else if (p > 0.0f) {
    double t = (double)p * 65536.0;
    d.thr = t >= 65535.0 ? 65535u : (uint32_t)t;
    if (d.thr == 0) d.thr = 1;
    d.scale = (float)(1.0 / (1.0 - (double)p));      // nn.Dropout's factor (the keep rate itself is p rounded down to 2^-16)
  }
  return d;
}
constexpr int PHILOX_ROUNDS = 7;
__device__ __forceinline__ void philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < PHILOX_ROUNDS; ++r) {
    // 64-bit products: ONE v_mad_u64_u32 each instead of a v_mul_hi_u32 + v_mul_lo_u32 pair (32-bit integer multiplies issue at a
    // quarter of the vector rate; cfg2 b = 32 step with dropout 12.76 -> 12.46 ms)
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// multipliers (scale or 0) of the aligned quad (row, 4 * quad .. 4 * quad + 3)
__device__ __forceinline__ void drop_quad(const DropCfg &d, uint32_t quad, uint32_t row, float (&m)[4]) {
  uint32_t w[4];
  philox4x32(d.seed_lo, d.seed_hi, quad, row, d.sid, drop_counter(d), w);
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = w[r] >= d.thr ? d.scale : 0.0f;
}
// ---- attention masks: rows R (bits 4, 5 clear), R + 16, R + 32, R + 48 share the call of (R, quad)
// window k of a word, in the word's top half (the bottom half is don't-care: the compares are against thr << 16)
__device__ __forceinline__ uint32_t drop_window(uint32_t w, uint32_t k) { return __builtin_amdgcn_alignbit(w, w, (16u + 8u * k) & 31u); }
// all four rows at once: keep[k][r] = decision of (R + 16 k, 4 quad + r); `row` must have bits 4 and 5 clear
// (keep decisions only -- the callers fold the scale elsewhere; 7 vector instructions per word: w << 16, w << 8, rotate, 4 compares)
__device__ __forceinline__ void drop_rows4(const DropCfg &d, uint32_t quad, uint32_t row, bool (&keep)[4][4]) {
  uint32_t w[4];
  philox4x32(d.seed_lo, d.seed_hi, quad, row, d.sid, drop_counter(d), w);
  const uint32_t th = d.thr << 16;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    keep[0][r] = (w[r] << 16) >= th;
    keep[1][r] = (w[r] << 8) >= th;
    keep[2][r] = w[r] >= th;
    keep[3][r] = __builtin_amdgcn_alignbit(w[r], w[r], 8) >= th;
  }
}
// two rows 16 apart: lo[r] / hi[r] = decisions of (R, 4 quad + r) / (R + 16, 4 quad + r); `row` must have bit 4 clear
__device__ __forceinline__ void drop_pair(const DropCfg &d, uint32_t quad, uint32_t row, bool (&lo)[4], bool (&hi)[4]) {
  uint32_t w[4];
  philox4x32(d.seed_lo, d.seed_hi, quad, row & ~48u, d.sid, drop_counter(d), w);
  const uint32_t th = d.thr << 16, k = (row >> 4) & 3u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    lo[r] = drop_window(w[r], k) >= th;
    hi[r] = drop_window(w[r], k + 1) >= th;
  }
}
// one row (any row): its own window of its group's call
__device__ __forceinline__ void drop_quad_attn(const DropCfg &d, uint32_t quad, uint32_t row, bool (&keep)[4]) {
  uint32_t w[4];
  philox4x32(d.seed_lo, d.seed_hi, quad, row & ~48u, d.sid, drop_counter(d), w);
  const uint32_t th = d.thr << 16, k = (row >> 4) & 3u;
#pragma unroll
  for (int r = 0; r < 4; ++r) keep[r] = drop_window(w[r], k) >= th;
}
__device__ __forceinline__ float drop_one(const DropCfg &d, uint32_t col, uint32_t row) {
  uint32_t w[4];
  const bool ff = (d.sid & DROP_SID_FF) != 0;
  philox4x32(d.seed_lo, d.seed_hi, col >> 2, ff ? row : (row & ~48u), d.sid, drop_counter(d), w);
  const uint32_t c = col & 3;
  const uint32_t v = c == 0 ? w[0] : (c == 1 ? w[1] : (c == 2 ? w[2] : w[3]));
  if (!ff) return drop_window(v, (row >> 4) & 3u) >= (d.thr << 16) ? d.scale : 0.0f;
  return v >= d.thr ? d.scale : 0.0f;
}
// Four lanes that are adjacent in a wave (a DPP quad) and hold the SAME column quad of four different rows -- the layout of
// attn_bwd_dkv_kernel: lane e of the quad has column 4 c + e of rows row0 .. row0 + 3 -- need 4 calls, not 16: lane e runs the
// call of row row0 + e, then word e of every call is fetched from the lane that ran it (quad_perm broadcasts, one VALU move each).
// m[r] = multiplier of (row0 + r, 4 quad + e).
// (attention masks: the call of row row0 + e is the one of its group of four, every row picks its own window)
__device__ __forceinline__ void drop_quad_transposed(const DropCfg &d, uint32_t quad, uint32_t row0, uint32_t e, float (&m)[4]) {
  uint32_t w[4];
  philox4x32(d.seed_lo, d.seed_hi, quad, (row0 + e) & ~48u, d.sid, drop_counter(d), w);
  const uint32_t th = d.thr << 16;
#define HN_QUAD_BCAST(v, r) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (r) * 0x55, 0xf, 0xf, true)
#define HN_QUAD_WORD(r)                                                                                          \
  {                                                                                                              \
    const uint32_t t0_ = HN_QUAD_BCAST(w[0], r), t1_ = HN_QUAD_BCAST(w[1], r), t2_ = HN_QUAD_BCAST(w[2], r),     \
                   t3_ = HN_QUAD_BCAST(w[3], r);                                                                 \
    const uint32_t v_ = e == 0 ? t0_ : (e == 1 ? t1_ : (e == 2 ? t2_ : t3_));                                    \
    m[r] = drop_window(v_, ((row0 + (r)) >> 4) & 3u) >= th ? d.scale : 0.0f;                                     \
  }
  HN_QUAD_WORD(0) HN_QUAD_WORD(1) HN_QUAD_WORD(2) HN_QUAD_WORD(3)
#undef HN_QUAD_WORD
#undef HN_QUAD_BCAST
}
struct GemmArgs {
  const float *A; long lda; long strideA;
  const float *W; long ldw; long strideW;      // W[n][k], NT
  float *C; long ldc; long strideC;
  const float *bias; long strideBias;          // may be NULL
  const float *R; long ldr; long strideR;      // residual, may be NULL (may alias C)
  int M, N, K, batch;
  int pro; const float *gamma; const float *beta; float eps;   // prologue on A (over k)
  float alpha;
  int act;
  int glu_offset;                              // row offset of the gate half in W / bias (GLU acts)
  int col_group, col_group_pitch;              // out col = (n / col_group) * col_group_pitch + n % col_group (0 = identity)
  // optional SECOND product on the same (prologued) A in the same launch: C2 = alpha2 * pro(A) W2^T (no bias / act / residual;
  // same K and ldw; N a multiple of 64).  The Q and K/V projections of a latent self-attention block share their input.
  const float *W2; float *C2; long ldc2; int N2; float alpha2; int col_group2, col_group_pitch2;
};
int launch_gemm(const GemmArgs &g, hipStream_t s);
// fp32 patch-bag K/V projection on LDS-DMA staged operands (gemm_nt.hip): C = alpha * A Ws^T + bs, Ws / bs staged by
// launch_gemm_nt_stage (LayerNorm affine of the context folded in, zero-padded to a multiple of 16 columns)
struct GemmNtArgs {
  const float *A; long lda;                    // (M, K), rows 16-byte aligned; columns K .. lda-1 must hold finite values
  const float *W; long ldw;                    // staged weight (N, ldw >= gemm_nt_ldws(K)), zero beyond K
  const float *bias;                           // staged bias row (N) or NULL
  float *C; long ldc;
  int M, N, K;
  float alpha;
  int col_group, col_group_pitch;              // as GemmArgs
  int ntm, ntn;                                // internal
};
bool gemm_nt_eligible(long M, int N, int K, long lda, const float *A, int col_group, int col_group_pitch, long ldc, const float *C);
int gemm_nt_ldws(int K);
size_t gemm_nt_stage_floats(int N, int K);
int gemm_nt_padded_cols(int N, int col_group, int col_group_pitch);
// col_group > 0: the staged image is head re-pitched (pitch rows per group of col_group source rows, pads zero): run the GEMM
// behind it with N = gemm_nt_padded_cols(...) and no column mapping
int launch_gemm_nt_stage(const float *W, long ldw, const float *gamma, const float *beta, const float *bias, int N, int K, float *Ws,
                         float *bs, hipStream_t s, int col_group = 0, int col_group_pitch = 0);
int launch_gemm_nt(const GemmNtArgs &g, int variant, hipStream_t s);
// fp32-exact GEMMs on the bf16 matrix pipe from three-plane operand images (gemm_x6.hip)
struct GemmX6Args {
  const unsigned short *Ap; int a_rt;          // planes of the context rows: (KT, a_rt, 3, 64, 8) bf16, a_rt padded to the tile
  const unsigned short *Wp; int w_rt;          // planes of the staged weight rows, w_rt padded to the tile
  const float *bias;                           // staged bias row (N) or NULL
  float *C; long ldc;
  int M, N, KT;                                // KT = ceil(K / 16) k-steps
  float alpha;
  int ntm, ntn; size_t a_bytes, w_bytes;       // internal
  int nsplit, kslice;                          // internal (split-k mode: C = partials (nsplit, M, ldc))
};
size_t x6_plane_bytes(long rows, int K, int row_tile);
int launch_x6_split(const float *X, long ldx, const float *scale, long R, int K, int row_tile, unsigned short *P, hipStream_t s,
                    int col_group = 0, int col_group_pitch = 0);
int launch_gemm_nt_x6(const GemmX6Args &g, int variant, hipStream_t s);
constexpr int X6_ROW_TILE = 8;                 // plane images are padded to 8 row tiles of 32 (the 256-row tile of the product kernels)
bool gemm_x6_enabled();                        // route switch HN_NO_X6_GEMM=1: the fp32-MFMA kernels of gemm_nt.hip (A/B tests)
bool gemm_nt_x6_eligible(long M, int N, int K);
int gemm_nt_x6_variant(long M);                // tile geometry by row count (launch_gemm_nt_x6's `variant`)
static inline int x6_col_tiles(int cols, int pad) { return ((cols + 31) / 32 + pad - 1) / pad * pad; }
// transposed images (rows of the image = columns of X, k = row index of X) and the long-contraction TN product on them
// pair order of the k index (rows of X) inside a transposed image: the 32 rows of a PAIR of 16-row tiles are dealt to two k-steps
// so that lane group g of the attention backward (rows 4 g .. 4 g + 3 of each tile) owns whole 16-byte fragment slots:
// k-step 2 P + gg, half h, element e  <-  row 32 P + 8 gg + 4 h + (e & 3) + 16 (e >> 2).  Any order works as long as both operands of
// a product use the same one; images of R % 32 == 0 rows are always built in pair order (x6_pair_order).
static inline bool x6_pair_order(long R) { return R % 32 == 0; }
int launch_x6_split_t(const float *X, long ldx, long R, int C, int col_tile, int ones_col, unsigned short *P, hipStream_t s);
bool gemm_tn_x6_eligible(long K, int M, int N);
size_t gemm_tn_x6_image_bytes(long K, int cols, int col_tile);
int launch_gemm_tn_x6(const unsigned short *At, const unsigned short *Bt, long K, int M, int N, float *G, long ldg, float *colsum, float *scratch,
                      size_t scratch_floats, hipStream_t s);
static inline int x6_row_tiles(long rows) { return (int)(((rows + 31) / 32 + X6_ROW_TILE - 1) / X6_ROW_TILE * X6_ROW_TILE); }
// long-contraction TN product C (+)= alpha * A^T B (+ colsum of A) on LDS-DMA staged k-tiles (gemm_nt.hip): G = dKV^T z of a patch bag
bool gemm_tn_glds_eligible(const float *A, long lda, const float *B, long ldb, int M, int N, int K);
int launch_gemm_tn_glds(const float *A, long lda, const float *B, long ldb, float *C, long ldc, int M, int N, int K, float alpha,
                        int accumulate, float *scratch, size_t scratch_floats, float *colsum, int colsum_accumulate, hipStream_t s,
                        int variant = 0);
// bf16-MFMA form of C = alpha * (A gamma + beta) W^T for the patch-bag K/V projection under core_precision = bf16 (gemm_bf16.hip):
// operands rounded to bf16 once, fp32 accumulation.  Ab = launch_rows_to_bf16(A) (M rows of gemm_bf16_pitch(K) bf16, once per
// forward), `stage` = gemm_bf16_stage_floats(N, K) floats of 16-byte aligned scratch per call
int gemm_bf16_pitch(int K);
bool gemm_bf16_shape_ok(long M, int N, int K);
bool gemm_bf16_eligible(const GemmArgs &g);
size_t gemm_bf16_stage_floats(int N, int K);
int launch_rows_to_bf16(const float *A, long lda, long M, int K, uint16_t *out, hipStream_t s);
int launch_gemm_bf16(const GemmArgs &g, const uint16_t *Ab, float *stage, hipStream_t s, uint16_t *K16 = nullptr, uint16_t *V16 = nullptr, int tokens = 0);
int launch_q_rows_to_bf16(const float *Q, long ldq, int b, int heads, int L, int Lp, uint16_t *Qf, hipStream_t s);

// Several skinny products of one shape in ONE launch, operands given per entry (the one-token projections of all layers of a
// forward: they do not depend on the latent array, and each is a ~9 us latency-bound launch on its own)
constexpr int HN_SKINNY_MAXZ = 16;
struct GemmSkinnyMulti {
  int nz;
  const float *A[HN_SKINNY_MAXZ]; long lda;
  const float *W[HN_SKINNY_MAXZ]; long ldw;
  const float *gamma[HN_SKINNY_MAXZ], *beta[HN_SKINNY_MAXZ];     // PRO_AFFINE operands (pro != PRO_NONE)
  const float *bias[HN_SKINNY_MAXZ];                              // may hold NULLs
  float *C[HN_SKINNY_MAXZ]; long ldc;
  int M, N, K, pro, act;
};
int launch_gemm_skinny_multi(const GemmSkinnyMulti &g, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// latent chain (chain.hip): out-projection / broadcast add -> feed-forward block -> next block's projections, one launch
// ------------------------------------------------------------------------------------------------
struct ChainArgs {
  int rows, L;                          // b * l_c rows of width 128 (16 per workgroup); l_c rows per sample
  const float *x_in; float *x_out;      // (rows, 128); x_out may be NULL or alias x_in
  float *x_mid;                         // training: x after the head stage (the feed-forward block's input) is kept on the tape; NULL otherwise
  int head;                             // 0: x = x_in; 1: x = x_in + LeakyReLU(O W_out^T + b_out); 2: x = x_in + y[row / L];
                                        // 3: as 1, with O built here from the split partials of a shared-context (rank-D) core:
                                        //    merged over the splits, normalised, run through the folded value projection
  const float *Opart, *Mpart, *Lpart;   // head 3: (b, heads, nsplit, Lp, dp), (b, heads, nsplit, Lp) x 2 (attention.hip)
  int nsplit, Lp, dp, heads, dh;        //         dp in {16, 32}, heads <= 8, dh in {16, 32, 64}, inner_o = heads * dh
  const float *wvf;                     //         (heads * dh, dp) folded value projection (vfold_kernel), row dp-1 = the beta term
  float *stats;                         //         (b, heads, L, 2) merged (max, sum) per row for hn_attn_probs, or NULL
  const float *O; int ldo, inner_o;     // head 1: merged attention output (rows, inner_o), inner_o a multiple of 32, <= 512
  const float *w_out, *b_out;           //         to_out.0.weight (128, inner_o), to_out.0.bias
  const float *y;                       // head 2: (b, 128) block output of a one-token cross-attention
  int has_ff, gate;                     // feed-forward block (hn_gate)
  const float *f_nw, *f_nb, *w1, *b1, *w2, *b2;
  int nq, nkv;                          // widths of the next attention block's projections (multiples of 128; 0 = none)
  const float *p_nw, *p_nb;             // that block's LayerNorm on x (NULL: none)
  const float *wq; float *Q; int ldq; float alpha_q;
  const float *wkv; float *KV; int ldkv;
  // Folded query of a shared-context (rank-D) block (inference; qfold_stage in vfold_kernel): with qf set, wq is the staged
  // (128, 128) product of W_q with the block's folded key weights (scale, gamma and the packed channel order included), nq = 128,
  // and the Q stage leaves what qfold_mfma_kernel would have produced from Q -- wave w = head w, 16 packed columns each:
  // qf (b, qf_heads = 8, Lp = L, 16) head-major, the Cauchy-Schwarz score bound of every row (sqrt(|qf row|^2 * qf_D), attention.hip)
  // and the fallback flag for bounds beyond the fp32-safe range
  float *qf, *qf_bound; int *qf_flag; int qf_heads, qf_D;
  float *xhat_out;                      // training: LN'(x) of the projection stage (rows, 128) goes to the tape (the backward's dW_q / dW_kv operand), or NULL
  // cluster mode (small batches; inference forward): exchange buffer (rows / 16 * 4 tiles of 16 x 128 floats), one flag per (tile,
  // member) + one error marker, zeroed at the start of the forward, and this chain's 1-based sequence number within the forward
  float *xchg; int *xflags; int seq;
  // Models staged into padded images (l_d < 128, odd head widths): the operands keep the shapes above, and
  //   dv      LayerNorm statistics over the first dv columns only (0 = 128; the pad columns of x, gamma, beta, weights are zero)
  //   o_cols  head 1: O has o_cols valid columns (multiple of 4, <= inner_o), the rest of the contraction reads as zero (0 = inner_o)
  //   q_cols / kv_cols  Q / KV keep their first q_cols / kv_cols columns (multiples of 16; 0 = nq / nkv): pitch ldq / ldkv
  // rows need not be a multiple of 16: every (rows, .) operand is then ALLOCATED for rows rounded up to 16 (the tail rows are
  // computed and stored like any other; nothing reads them).
  int dv, o_cols, q_cols, kv_cols;
  DropCfg ff_drop;                      // training: nn.Dropout on the feed-forward output (:347); thr == 0: off
  int cluster;                          // internal: members per row tile (launch_latent_chain decides)
  int tiles;                            // internal: row tiles (the grid may hold idle workgroups beyond them in cluster mode)
  unsigned *status; unsigned token, wait_ticks;      // internal: cluster_before_launch (status word of the device, this launch's token, wait bound)
  int split_order;                      // internal: HN_FORCE_CLUSTER_SPLIT_ORDER (chain_common.h cluster_decode)
  int inject_loss;                      // internal: hn_cluster_config(enable = 2) -- the last member of every tile withholds its flag
};
constexpr int CHAIN_XCHG_FLOATS = 256 * 16 * 128;      // <= 256 workgroups x one partial tile
constexpr int CHAIN_XCHG_FLAGS = 2048 + 1;           // cluster chains: <= 256 (tile, member) flags; layer chains (lchain.hip): one per row tile, b <= 256
bool latent_chain_supported(int rows, int d, int hidden);
int launch_latent_chain(const ChainArgs &a, hipStream_t s);
// ------------------------------------------------------------------------------------------------
// layer chain (lchain.hip): the whole latent side between two shared-context cores -- segments of [head] -> feed-forward block ->
// [projections -> latent self-attention], the x tile in LDS throughout, the self-attention a stage family of the weight ring
// ------------------------------------------------------------------------------------------------
constexpr int LSEG_MAX = 8;
constexpr int LAYER_MAXBLK = 1024;       // weight / K / V^T blocks of one launch (its LDS block table)
struct LSeg {
  int head;                             // 0: x as it is; 2: x += y[sample]; 3: out-projection of the core's merged split partials (segment 0 only);
                                        // 4: out-projection of the self-attention output the previous segment (proj = 1) left in LDS
  int gate;                             // feed-forward block (hn_gate)
  int proj;                             // 0: none; 1: Q | K | V (512 each) + the latent self-attention; 2: folded query of the next rank-D block (last segment)
  int kv_slot;                          // proj 1: which K / V^T image of the launch (0, 1, ...: in segment order)
  float alpha_q;                        // proj 1: scale of the query projection (softmax scale in log2 units)
  const float *w_out, *b_out;           // head 3 / 4: (128, 512), (128)
  const float *y;                       // head 2: (b, 128)
  const float *f_nw, *f_nb, *w1, *b1, *w2, *b2;
  const float *p_nw, *p_nb, *wq, *wkv;  // proj: LayerNorm on x, (512 | 128, 128) query weights, (1024, 128) key / value weights
  float *x_out;                         // x after the feed-forward block, (rows, 128), or NULL
  float *stats;                         // proj 1: (b, 8, 128, 2) softmax (max, sum) of the self-attention rows for hn_attn_probs, or NULL
};
struct LayerChainArgs {
  int b, nseg;                          // samples (l_c = 128 rows each, 8 workgroups), segments
  const float *x_in;                    // (b * 128, 128)
  const float *Opart, *Mpart, *Lpart, *wvf; int nsplit, Lp, heads, dh;      // head 3 (ChainArgs)
  float *stats3;                        // head 3: (b, heads, 128, 2) merged (max, sum) per row, or NULL
  float *qf, *qf_bound; int *qf_flag; int qf_D;                             // proj 2 (ChainArgs)
  float *kbuf, *vtbuf; long kv_stride;  // K (b, 8, 128, 64) and V^T (b, 8, 64, 128) images, kv_stride floats between the slots
  int *xflags; int flag_count, seq;     // one flag per row tile + a marker, zeroed at the start of the forward; first sequence number (one per proj 1 segment)
  unsigned *status; unsigned token, wait_ticks; int inject_loss, flag_marker;      // internal (cluster_before_launch)
  LSeg seg[LSEG_MAX];
};
bool latent_layer_enabled();            // false: HN_NO_SELF_IN_CHAIN (route switch)
int latent_layer_segment_blocks(int head, int proj);      // blocks a segment adds to its launch (<= LAYER_MAXBLK - 6 per launch)
int launch_latent_layer(const LayerChainArgs &a, hipStream_t s);
// backward of the latent chain (bchain.hip): projection backward of an attention block -> feed-forward block backward -> the
// out-projection backward of the attention block in front of it, one launch; weight gradients by launch_gemm_tn_multi
struct BChainArgs {
  int rows, L;
  const float *dy;                     // (rows, 128) gradient entering the chain (w.r.t. the output of its last block in forward order)
  float *dx_out;                       // (rows, 128) gradient leaving: w.r.t. the input of the feed-forward block (of the P block without one)
  int has_p;                           // ---- P: dxh = dQ W_q + dKV W_kv ; G = LN'(dxh; p_x, p_nw) + dy
  const float *dQ; int lddq, nq;       // (rows, nq), scaled as the q projection's output
  const float *dKV; int lddkv, nkv;    // (rows, nkv) or nkv = 0 (cross blocks: the context carries no gradient)
  const float *wqT, *wkvT;             // transposed projection weights (128, nq), (128, nkv)
  const float *p_x, *p_nw;             // the attention block's input (tape) and its LayerNorm weight (NULL: no LayerNorm)
  int has_ff, gate;                    // ---- FF
  const float *f_x;                    // the feed-forward block's input (tape)
  const float *f_nw, *f_nb, *w1, *b1;  // LayerNorm affine, first layer (1024, 128) + bias
  const float *w2T, *w1T;              // transposes: (512, 128) of net.2.weight, (128, 1024) of net.0.weight
  float *H, *dU, *Xhat, *dYff;         // out: h (rows, 512), dU (rows, 1024), LN(f_x) (rows, 128), G entering the block (rows, 128)
  int has_out, inner_o;                // ---- OUT: dpre = G * LeakyReLU'(f_x - o_x) ; dO = dpre W_out
  const float *o_x, *woT;              // the attention block's input (tape), W_out^T (inner_o, 128)
  float *dPre, *dO; int lddo;          // out: (rows, 128), (rows, inner_o)
  float *lnpart;                       // out: (rows / 16, 4, 128) per-workgroup partial [dgamma_p, dbeta_p, dgamma_f, dbeta_f]
  // cluster mode (<= 128 row tiles): two exchange buffers of (tiles * C) partial tiles, two flag sets + an error marker (zeroed at
  // the start of the backward), this chain's 1-based sequence number within the backward
  float *xchg; int *xflags; int seq;
  // staged (padded) models, as in ChainArgs: LayerNorm over the first dv columns (0 = 128); dQ / dKV have q_cols / kv_cols valid
  // columns (multiples of 4; 0 = nq / nkv; the rest of the contraction reads as zero), dO keeps its first o_cols columns
  // (multiple of 16; 0 = inner_o).  rows need not be a multiple of 16 (operands allocated for the rounded-up count; the
  // LayerNorm partial sums skip the tail rows, the weight-gradient products contract over `rows` only).
  int dv, q_cols, kv_cols, o_cols;
  DropCfg ff_drop;                     // the forward's dropout on the feed-forward output (thr == 0: off)
  int cluster;                         // internal
  int tiles;                           // internal
  unsigned *status; unsigned token, wait_ticks;      // internal (cluster_before_launch)
  int split_order, inject_loss;        // internal
};
bool latent_bchain_supported(int rows, int d, int hidden);
int launch_latent_bchain(const BChainArgs &a, hipStream_t s);
// several TN products C_i += A_i^T B_i (+ colsum_i += column sums of A_i) over ONE contraction length K in one launch + one reduce
constexpr int TN_MULTI_MAX = 32;       // products of one batched launch (round 5: the chains of a whole layer; 6 until then: one chain)
constexpr int TN_MULTI_LN_MAX = 16;    // LayerNorm partial-sum entries of one batched launch (the argument struct stays below 4 KB)
struct TnProduct {
  const float *A; long lda;            // (K, M)
  const float *B; long ldb;            // (K, N)
  float *C; long ldc;                  // (M, N), accumulated into
  int M, N;
  float *colsum;                       // (M) accumulated into, or NULL
  long part_off, cs_off;               // internal: scratch offsets of the split partials
  // internal: products of one launch that accumulate into the SAME C (a module shared by two blocks of a layer: the latent self block
  // runs behind every modality) are folded by ONE reduce pass, in launch order -- `next`: the following product of the chain or
  // -1; `follower`: not the head of its chain (its reduce blocks return at once)
  int next, follower;
};
struct LnPartial { const float *part; int nwg, width; long stride; float *out; int next, follower; };     // out[c] += sum_w part[w * stride + c]
struct GemmTnMulti {
  int n, n_ln;
  TnProduct p[TN_MULTI_MAX];
  LnPartial ln[TN_MULTI_LN_MAX];
  int tile0[TN_MULTI_MAX + 1];         // internal
  int K, kslice, nsplit;
  float *scratch;
};
static_assert(sizeof(GemmTnMulti) <= 3840, "GemmTnMulti travels as a kernel argument (4 KB limit)");
size_t gemm_tn_multi_scratch_floats(const GemmTnMulti &m);
int launch_gemm_tn_multi(GemmTnMulti &m, float *scratch, size_t scratch_floats, hipStream_t s);
const float *transpose_cache_lookup(const float *src, long ld, int rows, int cols);
constexpr int CHAIN_MERGE_GROUP = 12;        // head 3 folds the splits in groups of this many (registers)
constexpr int CHAIN_MERGE_MAX_SPLITS = 48;   // ... and at most this many in all (more: merge_vproj_kernel)
// folded value projection of a shared-context block for ChainArgs.wvf (the image merge_vproj_kernel builds per workgroup)
struct VfoldMulti {                    // one entry per layer of a modality (<= HN_SKINNY_MAXZ): value half of to_kv, context LayerNorm affine
  int n;
  const float *w_v[16], *gamma[16], *beta[16];
  float *out; long out_stride;         // entry z writes out + z * out_stride, (heads * dh, 16)
  int D, heads, dh, pack_ks;
  // query side (qout != NULL): W_f = (folded key weights of the layer)^T W_q, (heads * 16, l_d) per layer -- the latent chain in
  // front of the block then projects LN(x) straight to the folded query (ChainArgs.qf)
  const float *w_k[16], *w_q[16];
  float cscale; int l_d;
  float *qout; long qout_stride;
  // a second ROLE of the launch (workgroups with blockIdx.y >= n): the latent array's broadcast over the batch + the forward's
  // pre-zeroed flags (broadcast_rows_kernel's work: a launch at the ~5 us floor less per forward); bc_dst == NULL: none
  const float *bc_src; float *bc_dst; long bc_per, bc_total; int *bc_zero; int bc_nzero; int bc_rows;
};
int launch_vfold(const VfoldMulti &v, hipStream_t s);
int vfold_plan(const VfoldMulti &v, VfoldMulti *vv, int *gx, int *gy, int *gz);      // the grid + argument block of the vfold roles (chain.hip)

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------
int launch_encode(const void *data, int in_dtype, int b, int n_axes, const int *spatial, int C, int F, float max_freq,
                  int fourier, int normalize, float eps, float *out, int ld_out, hipStream_t s, int ones_col = -1,
                  int pack_ks = 0, int axis0_begin = 0, int axis0_total = 0);
// the arguments of one launch_encode call, and the inference forward's prelude as ONE launch (encode.hip prelude_kernel): the
// one-token modality's encode + vfold's roles (folded projections, latent broadcast, flags) + the RGB image's encode
struct EncodeCall {
  const void *data; int dtype, b, n_axes; const int *spatial; int C, F; float max_freq; int fourier, normalize; float eps;
  float *out; int ld_out, ones_col, pack_ks;
};
bool encode_prelude_eligible(const EncodeCall &tab, const EncodeCall &img);
int launch_encode_prelude(const EncodeCall &tab, const EncodeCall &img, const VfoldMulti &v, hipStream_t s);      // axis0_total > 0: `spatial[0]` rows from axis0_begin of a longer axis
int launch_encode_bf16ctx(const void *data, int in_dtype, int b, int n_axes, const int *spatial, int C, int F, float max_freq,
                          int fourier, float eps, uint16_t *zb, uint16_t *zT, int Np, int DV, int ns, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// attention pieces
// ------------------------------------------------------------------------------------------------
struct AttnCoreArgs {
  const float *Q;  long q_b, q_h; int ldq;      // (b, h, Lp rows, DP) pre-scaled queries, zero padded rows/cols
  const float *Kp; long k_b, k_h; int ldk;      // key rows:   Kp + b*k_b + h*k_h + t*ldk + d
  const float *Vp; long v_b, v_h; int ldv;      // value rows
  const uint8_t *mask;                          // (b, N) or NULL
  float *Opart, *Mpart, *Lpart;                 // (b, h, nsplit, Lp, DP), (b, h, nsplit, Lp) x2
  int b, h, Lq, Lp, N, dp;                      // Lq valid query rows, Lp = Lq rounded up to 16, dp in {16,32,64,128}
  int nsplit, chunk;                            // tokens per split (multiple of 16)
  int ones_col;                                 // rank-D binding with D <= dp-1: synthetic ones column dp-1 (see attention.hip)
  int ones_in_mem;                              // ... and the context rows already carry 1.0 there (written by K1)
  int qk_steps;                                 // packed context layout: QK^T k-steps to run (0 = all dp/4)
  const float *bound; const int *bound_flag;    // ONES + LayerNorm-ed context: per-row score bounds (b, h, Lp) and the
                                                // device flag that disables them (see qfold_kernel); NULL = running max
  float *Ofinal; int ldo, dh; float *stats;     // nsplit == 1 only: write the normalised O (b*Lq, ldo) + stats directly (no merge kernel)
  DropCfg drop;                                 // training: dropout on the probabilities (thr == 0: off; needs ones_col == 0)
  int drop_rowsum;                              // ... shared-context binding: also accumulate sum_t p'_t in column dp-1
  int nq;                                       // query tiles per wave the token split was planned for (0: the kernel's default)
};
int launch_attn_core(const AttnCoreArgs &a, hipStream_t s);
bool launch_qfold_mfma_bf16(const float *Q, int ldq_row, const float *w_k, int D, const float *gamma, float cscale, uint16_t *Qf,
                            int b, int h, int L, int Lp, int dh, hipStream_t s, float *bound, int *bound_flag);
// small batches of the dp = 16 shared-context binding: fewer query tiles per wave (more work items) and at most 12 splits, so that
// the chain behind the block can merge them itself (chain.hip head 3) instead of a merge launch over up to 256 splits
int attn_core_nq_small_batch(int dp, int b, int h, int Lp);
// one workgroup per (sample, head) with K / V in LDS: the latent self-attention shape (self_attention.hip)
bool self_core_lds_eligible(const AttnCoreArgs &a);
int launch_self_core_lds(const AttnCoreArgs &a, hipStream_t s);

struct AttnCoreBf16Args {                        // bf16-MFMA core of the shared-context binding (attention_bf16.hip)
  const uint16_t *Qf;                            // (b, h, Lp, 32) bf16 folded queries
  const uint16_t *zb, *zT;                       // (b, Np, 32) token-major / (b, DV, Np) channel-major context images
  const uint8_t *mask;                           // (b, N) or NULL
  float *Opart, *Mpart, *Lpart;                  // (b, h, nsplit, Lp, DV), (b, h, nsplit, Lp) x2
  int b, h, Lq, Lp, N, Np, DV;
  int nsplit, chunk;                             // tokens per split (multiple of 32)
  int ns;                                        // operand planes: 1 = plain bf16, 2 = hi + lo pairs ("bf16x3")
  int no_pipeline;                               // development knob HN_BF16_NO_PIPELINE: the general loop also for the bounded, unmasked case
  const float *bound; const int *bound_flag;     // per-row score bounds + fallback flag (qfold_bf16_kernel), or NULL
  int expl, k_pitch;                             // explicit K / V binding (see the kernel): zb = K image with k_pitch bytes per token row, zT = V image per (b, head), Qf (b, h, Lp, 64)
};
// bf16 slots per context / query row of the QK^T contraction (see attention_bf16.hip)
__host__ __device__ constexpr int bf16_row_slots(int DV, int ns) { return ns == 1 ? 32 : (DV == 16 ? 64 : 96); }
int launch_attn_core_bf16(const AttnCoreBf16Args &a, hipStream_t s);
int launch_qfold_bf16(const float *Q, int ldq_row, const float *w_k, int D, const float *gamma, float cscale, uint16_t *Qf,
                      int b, int h, int L, int Lp, int dh, int DV, int ns, hipStream_t s, float *bound = nullptr, int *bound_flag = nullptr);
// waves_per_simd > 0: size the token split for that many resident waves per SIMD (kernels with more than 128 VGPRs hold 3:
// a split sized for 4 would run a second, mostly idle round)
void attn_core_geometry(int b, int h, int Lp, int N, int dp, int *nsplit, int *chunk, int waves_per_simd = 0, int nq = 0);

int launch_qfold(const float *Q, int ldq_row, const float *w_k, int D, const float *gamma, float cscale,
                 float *Qf, int b, int h, int L, int Lp, int dh, int dp, hipStream_t s, int pack_ks = 0, float *bound = nullptr,
                 int *bound_flag = nullptr);
int launch_merge_vproj(const float *Opart, const float *Mpart, const float *Lpart, int nsplit, int b, int h,
                       int L, int Lp, int dp, int D, const float *gamma, const float *beta, const float *w_v,
                       int dh, float *O, int ldo, float *stats, float *oprime_save, hipStream_t s, int pack_ks = 0, int srow = 0);
int launch_merge_explicit(const float *Opart, const float *Mpart, const float *Lpart, int nsplit, int b, int h,
                          int L, int Lp, int dp, int dh, float *O, int ldo, float *stats, hipStream_t s);
int launch_probs(const float *Q, long q_b, long q_h, int ldq, int dp, const float *Kp, long k_b, long k_h, int ldk,
                 const uint8_t *mask, const float *stats, float *P, int b, int h, int L, int N, hipStream_t s);
int launch_importance(const float *Q, long q_b, long q_h, int ldq, int dp, const float *Kp, long k_b, long k_h, int ldk,
                      const uint8_t *mask, const float *stats, float *I, int b, int h, int L, int N, hipStream_t s);


int launch_copy(float *dst, const float *src, long n, hipStream_t s);
int launch_fourier_encode(const float *x, float *out, long n, int F, float max_freq, hipStream_t s);
int launch_glu_gate(const float *x, float *out, long rows, int hid, int gelu, hipStream_t s);
int launch_temperature_softmax(const float *x, float *y, long rows, int n, float temperature, hipStream_t s);
int launch_fill_bytes(uint8_t *dst, uint8_t value, long n, hipStream_t s);
int launch_dropout_apply(const float *src, const float *add, float *out, long rows, int cols, const DropCfg &d, hipStream_t s);
int launch_dropout_mask(uint8_t *mask, long rows, int cols, const DropCfg &d, hipStream_t s);
// ---- cluster-mode chain launches (chain.hip / bchain.hip): host side of the co-residency contract ----
// A cluster launch spins on the flags of workgroups of the same grid.  Three things keep that safe:
//  * the dispatch order of the grid (cluster_decode, chain_common.h) keeps the members of a tile together, so that a grid
//    progresses on a partly occupied chip (an RCCL kernel on a side stream, a CU mask);
//  * two cluster launches of one device never run at once: cluster_before_launch orders a launch behind the previous one when
//    that went to ANOTHER stream, through an event recorded right behind that launch (cluster_after_launch) -- a stream handle is
//    only ever compared, never used after the call it arrived in (the old guard recorded on a possibly destroyed stream);
//  * a wait that still runs into its bound (HN_CLUSTER_TIMEOUT_US, default 100 ms) turns the tile into NaN and stores the
//    launch's token into the device's host-mapped STATUS WORD.  cluster_poll() -- called first thing by every fused entry point
//    -- sees a non-zero word, switches cluster mode off for the device (sticky), DRAINS the device, clears the word and returns
//    HN_E_CORESIDENCY once (cleared before the drain, an Adam step enqueued earlier could still read a clean word);
//    hn_l1_adam_step's kernel reads the word and leaves parameters and moments alone while it is set.  On a capturing stream
//    nothing can be drained: the word stays set (every entry point keeps failing) until hn_cluster_status acknowledges it.
struct ClusterTicket { unsigned *status; unsigned token, wait_ticks; int inject_loss; };
bool cluster_enabled(int dev);                                         // false after a lost exchange / hn_cluster_config(enable = 0) / HN_NO_CHAIN_CLUSTER
void cluster_before_launch(int dev, hipStream_t s, ClusterTicket *t);
void cluster_after_launch(int dev, hipStream_t s);
int cluster_poll(const char *who, hipStream_t s);                      // HN_OK or HN_E_CORESIDENCY (current device)
const unsigned *cluster_status_device_word(int dev);                   // device-visible address of the status word, or NULL (never allocated)
int cluster_status(int dev, int acknowledge, hn_cluster_info *info);
int cluster_config(int dev, int enable, int timeout_us);

// training-step tail (train.hip)
int launch_surv_nll(const float *logits, const long long *y, const float *cens, const float *weights, int b, int K, float alpha,
                    float eps, float grad_scale, float *loss, float *dlogits, float *hazards, float *survival, float *risk,
                    hipStream_t s);
constexpr int L1_ADAM_PARTIALS = 1024 + 16;      // one partial |p| sum per block, then the step's snapshot of the cluster status word
int launch_l1_adam(float *p, const float *g, float *m, float *v, long n, double l1, double grad_scale, double lr, double beta1,
                   double beta2, double eps, int step, float *reg_loss, float *partial, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// backward building blocks (backward.hip)
// ------------------------------------------------------------------------------------------------
struct GemmExArgs {                              // C[i,j] (+)= alpha * sum_c A(i,c) * B(j,c)
  const float *A; long a_rs, a_cs, strideA;      // element (row i, contraction c) at A + i*a_rs + c*a_cs
  const float *B; long b_rs, b_cs, strideB;
  float *C; long ldc, strideC;
  int M, N, K, batch;
  float alpha;
  int accumulate;
  int k_total;                                   // internal (split-k launches): full contraction length, 0 otherwise
  float *colsum; int colsum_accumulate;          // TN form only: also colsum[i] (+)= sum_c A(i, c) (the bias gradient of dW = dY^T X)
};
constexpr int GEMM_EX_SPLITS = 32;
int launch_gemm_ex(const GemmExArgs &g, hipStream_t s, float *scratch = nullptr);
int launch_colsum(const float *X, long ld, long rows, int cols, float scale, float *out, int accumulate, hipStream_t s,
                  float *scratch = nullptr);
size_t reduce_scratch_floats(long max_mn, int max_cols);
size_t ln_bwd_scratch_floats(long rows, int d);
// transposed-weight cache of one backward pass (backward.hip)
void transpose_cache_begin();
void transpose_cache_end();
void transpose_cache_add(const float *src, long ld, int rows, int cols);
size_t transpose_cache_floats();
int transpose_cache_run(float *buf, size_t buf_floats, hipStream_t s);
int launch_ln_bwd(const float *x, const float *dy, const float *gamma, long rows, int d, float *dx, int dx_accumulate,
                  float *dgamma, float *dbeta, float *scratch, hipStream_t s);
int launch_ln_fwd(const float *x, const float *gamma, const float *beta, long rows, int d, float *y, hipStream_t s, int dv = 0);
int launch_leaky_bwd(const float *dy, const float *x_out, const float *x_in, float *dpre, long n, hipStream_t s);
int launch_glu_bwd(float *u, const float *dh, float *h_out, long rows, int hid, int gelu, hipStream_t s);
int launch_add_into(const float *src, float *dst, long n, int accumulate, hipStream_t s);
size_t head_bwd_scratch_floats(int b, int d, int out_dims);
int launch_head_bwd(const float *x, int b, int L, int d, const float *nw, const float *nb, const float *w, int out_dims,
                    const float *dlogits, float *dx, float *dnw, float *dnb, float *dw, float *dbias, float *scratch,
                    hipStream_t s, int dv = 0, int *zero = nullptr, int nzero = 0);      // zero: nzero ints cleared on the way (the backward chains' cluster flags)

// ------------------------------------------------------------------------------------------------
// attention backward (attention_bwd.hip)
// ------------------------------------------------------------------------------------------------
struct AttnBwdArgs {
  const float *Q;  long q_b, q_h; int ldq;      // scaled query operand of the forward core
  const float *dO; long do_b, do_h; int lddo;   // gradient of the (normalised) attention output, same fragment layout
  const float *Kp; long k_b, k_h; int ldk;
  const float *Vp; long v_b, v_h; int ldv;
  const uint8_t *mask;
  const float *stats;                            // (b, h, Lq, 2) from the forward
  const float *delta;                            // (b, h, Lq): sum_d dO * O
  float *dQpart;                                 // (b, h, nsplit, Lp, dp)
  float *dKV; float dk_scale;                    // dkv kernel: compact (b*N, 2*inner) output
  unsigned short *dkv3; int dkv3_ct;             // ... or (LDS kernel, two token tiles per wave) straight into the transposed three-plane image of
                                                 // gemm_x6.hip (dkv3_ct column tiles, k-steps in PAIR order: x6_pair_row), dKV itself not written
  int b, h, Lq, Lp, N, dp, nsplit, chunk;
  DropCfg drop;                                  // the forward's dropout on the probabilities (thr == 0: off)
  int drop_rowsum;                               // shared-context binding under dropout: V carries a ones column dp-1
  int qk_steps;                                  // > 0: packed shared context, k-steps of the channel contractions (see attn_core)
  // one split (the latent self-attention, short contexts): the dQ kernel writes the finished rows itself -- (b*Lq rows, ld dq_ld), head hi
  // at column hi * dq_pitch, dq_width valid columns, times dq_scale -- exactly what dq_reduce would make of the single partial; NULL: partials
  float *dQfinal; int dq_ld, dq_pitch, dq_width; float dq_scale;
};
int launch_pack_fold(float *x, int ld, int h, int D, int dp, int ks, int mode, long rows, hipStream_t s, int srow = 0);
int launch_attn_bwd_dq(const AttnBwdArgs &a, hipStream_t s);
// explicit binding, dp = 64: dQ on a workgroup-shared LDS ring of K / V tiles (attention_lds.hip)
bool attn_bwd_dq_lds_eligible(const AttnBwdArgs &a);
int launch_attn_bwd_dq_lds(const AttnBwdArgs &a, hipStream_t s);
int launch_dq_reduce(const float *part, int nsplit, int b, int h, int L, int Lp, int dp, int width, float scale, float *out,
                     int ld_out, int head_pitch, hipStream_t s);
int launch_attn_bwd_dkv(const AttnBwdArgs &a, int dh, int inner, hipStream_t s, bool *wrote_planes = nullptr);
// latent self-attention: dQ and dK/dV side by side in one launch (attention_bwd.hip); false = not this shape, nothing launched
bool launch_attn_bwd_self_pair(const AttnBwdArgs &a, int dh, int inner, hipStream_t s, int *rc_out);
int launch_rowdot_heads(const float *X, int ldx, int xpitch, const float *Y, int ldy, int ypitch, int h, int L, int width,
                        long rows, float *delta, hipStream_t s);
int launch_head_affine(const float *src, int lds, int spitch, const float *mul, int ldm, int mpitch, const float *colscale,
                       const float *coladd, float scale, int h, int width, int dpitch, int ldd, long rows, float *dst,
                       hipStream_t s);
// rank-D binding under dropout (row-sum channel dp-1 of the saved average, see attention_bwd.hip)
int launch_srow_affine(const float *saved, const float *dA, const float *gamma, const float *beta, int mode, int h, int D, int dp,
                       long rows, float *dst, hipStream_t s);
int launch_kv_weight_grads(const float *G, const float *cs, const float *w, const float *gamma, const float *beta, int nrows, int D,
                           float *dw, float *dgamma, float *dbeta, hipStream_t s, float *scratch);
int launch_segsum(const float *X, int seg, int cols, int nseg, float *out, hipStream_t s);
// backward of the one-token cross block in three launches (backward.hip)
bool onetoken_bwd_fused_ok(int b, int qd);
int launch_onetoken_bwd(const float *dy, const float *x_out, const float *x_in, int b, int L, int qd, const float *w_out, long ldwo,
                        int inner, const float *V, const float *ctx, int ld_ctx, int D, const float *w_v, const float *gamma,
                        const float *beta, float *dyb, float *dV, float *dw_out, float *db_out, float *dw_v, float *dgamma, float *dbeta,
                        float *partial, hipStream_t s);

// misc
int launch_broadcast_rows(const float *src, float *dst, long n_per, int b, hipStream_t s, int *zero = nullptr, int nzero = 0);
int launch_head(const float *x, int b, int L, int d, const float *nw, const float *nb, const float *w,
                const float *bias, int out_dims, float *logits, hipStream_t s, int dv = 0);
int launch_pad_rows(const float *src, int ld_src, float *dst, int ld_dst, long rows, int cols, hipStream_t s);
// Staged models (api_entry.hip): a table of 2-D pieces in ONE launch -- dst[r, c] (+)= r < rows_src && c < cols_src ? src[r, c] : 0 over
// the piece's (rows_dst, cols_dst) rectangle.  Weights into their zero-padded images, padded gradients back onto the real ones.
struct StagePiece { const float *src; float *dst; int rows_src, cols_src, ld_src, rows_dst, cols_dst, ld_dst; };
constexpr int STAGE_MAX = 84;
struct StageTable { int n, accumulate; StagePiece p[STAGE_MAX]; };
int launch_stage(const StageTable &t, hipStream_t s);
int launch_add_row_broadcast(const float *y, const float *x_in, float *x_out, int b, int L, int d, hipStream_t s);
int launch_fill(float *dst, float value, long n, hipStream_t s);

}  // namespace hn
