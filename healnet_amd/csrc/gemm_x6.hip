// fp32-exact NT / TN GEMMs of the patch bag on the bf16 matrix pipe: every fp32 operand is split into THREE bf16 planes
//
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)        (round to nearest even; both differences exact)
//
// 3 x 8 significand bits (+ the signs of m and l) hold all 24 bits of an fp32, so the split is EXACT for every normal x below the
// largest bf16 (3.39e38: the top 0.4 % of the last binade rounds to Inf) whose last piece does not underflow; a product x y is then
// the sum of nine bf16 x bf16 products, each exact in fp32.  Six of them are formed (h h, h m, m h, h l, m m, l h) -- the three
// dropped ones are 2^-27.4 |x y| rms (at most 2^-23 in the worst alignment of both operands; tests/test_x6_split_math.py), a quarter
// of the rms rounding error of ONE fp32 product (2^-25.2) -- by v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 x 32 cycles for
// 32 x 32 x 16 products against 16 x 32 cycles of v_mfma_f32_16x16x4_f32, i.e. the fp32 result at 3/8 of the fp32 MFMA time (the
// bf16 pipe is 16 x the fp32 pipe on gfx950).
// tools/ubench/gemm_x6_bench.hip measures both kernels against an fp64 reference: the error of this one is not larger.
//
// The K/V projection of healnet/models/healnet.py:405 on a (b * N, D) patch bag (32 768 x 773 -> 1024 at BASELINE configs[3]) and
// its weight gradient are half of the training step on the fp32 MFMA (gemm_nt.hip, 0.85 / 0.82 of that peak); the normalised bag
// is the same for every layer of the model, so its planes are built once per step.
//
// Plane images are stored as the MFMA wants them ("fragment-major"):
//
//     P[kt][rt][plane][lane][8]   bf16,   lane -> row rt * 32 + (lane & 31),  k = kt * 16 + (lane >> 5) * 8 + 0 .. 7
//
// so one (kt, rt, plane) fragment is 1 KB that a wave moves global -> LDS with ONE LDS-DMA instruction (source and destination both
// linear in the lane index) and reads back with ONE ds_read_b128 at base + 16 * lane: no swizzle, no bank conflict, no address
// arithmetic in the loop, full 128-byte lines everywhere.  A workgroup's share of a 16-wide k-step is contiguous in memory.
#include "common.h"
#pragma clang diagnostic ignored "-Winline-asm"

namespace hn {

typedef __attribute__((ext_vector_type(8))) __bf16 x6_bf16x8;

__device__ __forceinline__ void x6_glds16(const i32x4 &rsrc, unsigned lds_byte, int voffset, unsigned soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_byte), "v"(voffset), "s"(rsrc), "s"(soffset)
               : "memory", "m0");
}
template <int N> __device__ __forceinline__ void x6_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ unsigned x6_cvt_pk(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// two fp32 -> (h, m, l) pairs packed as bf16x2 words
__device__ __forceinline__ void x6_split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  h = x6_cvt_pk(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = x6_cvt_pk(ra, rb);
  const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
  l = x6_cvt_pk(sa, sb);
}

// ------------------------------------------------------------------------------------------------
// split: X (R, K) fp32 row-major -> planes (KT, Rt, 3, 64, 8).  Rows >= R and columns >= K are zero (Rt may be padded up to the
// GEMM's tile).  `scale` (K) multiplies the columns (the LayerNorm gamma folded into a staged weight) or is NULL.
// A wave owns one row tile and FOUR consecutive k-steps (64 columns: two full lines of every row).
// ------------------------------------------------------------------------------------------------
struct X6SplitArgs {
  const float *X; long ldx;
  const float *scale;
  int R, K, Rt, KT;
  unsigned short *P;
  int col_group, col_group_pitch;      // head re-pitching of the ROWS of a staged weight (as gemm_nt_stage_kernel)
};

__global__ __launch_bounds__(256) void x6_split_kernel(X6SplitArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rt = blockIdx.x, kt0 = (blockIdx.y * 4 + wave) * 4;
  if (kt0 >= g.KT) return;
  int row = rt * 32 + (lane & 31);
  bool live = row < g.R;
  if (g.col_group > 0) {
    const int c = row % g.col_group_pitch;
    live = live && c < g.col_group;
    row = (row / g.col_group_pitch) * g.col_group + c;
  }
  const float *x = g.X + (long)(live ? row : 0) * g.ldx;
  float v[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k0 = (kt0 + q) * 16 + (lane >> 5) * 8;
    if (live && k0 + 8 <= g.K && (g.ldx & 3) == 0) {
      const f32x4 a = *(const f32x4 *)&x[k0], b = *(const f32x4 *)&x[k0 + 4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[q][e] = a[e]; v[q][4 + e] = b[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[q][e] = (live && k0 + e < g.K) ? x[k0 + e] : 0.0f;
    }
    if (g.scale) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[q][e] *= (k0 + e < g.K) ? g.scale[k0 + e] : 0.0f;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (kt0 + q >= g.KT) break;
    u32x4 h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned hh, mm, ll;
      x6_split2(v[q][2 * e], v[q][2 * e + 1], hh, mm, ll);
      h[e] = hh; m[e] = mm; l[e] = ll;
    }
    u32x4 *dst = (u32x4 *)(g.P + ((((long)(kt0 + q) * g.Rt + rt) * 3) * 64 + lane) * 8);
    dst[0] = h; dst[64] = m; dst[128] = l;
  }
}

// ------------------------------------------------------------------------------------------------
// C[m, n] = alpha * sum_k A[m, k] W[n, k] + bias[n] from plane images.  Workgroup = NWM x NWN waves, a wave owns WM x WN tiles of
// 32 x 32; stage = one 16-wide k-step of the workgroup's (BM + BN) rows x 3 planes; S-slot ring, one barrier per stage.
// The MFMA takes the WEIGHT fragment as its A operand: a lane ends up with four consecutive output columns of one row.
// ------------------------------------------------------------------------------------------------
// ABL: development ablations (tools/ubench/gemm_x6_bench.hip): 1 = no output stores, 2 = no operand loads behind the prologue,
// 4 = no barrier / load waits in the loop, 8 = no fragment reads in the loop.  The product instantiates ABL = 0 only.
template <int WM, int WN, int NWM, int NWN, int S, int MODE = 0, int ABL = 0>
__global__ __launch_bounds__(NWM *NWN * 64) void gemm_nt_x6_kernel(GemmX6Args g) {
  constexpr int NW = NWM * NWN, TA = NWM * WM, TW = NWN * WN;            // row tiles of 32 per workgroup
  constexpr int LWA = TA * 3 / NW, LWW = (TW * 3 + NW - 1) / NW, LW = LWA + LWW;   // 1 KB pieces per stage and wave: context, weight
  // (every wave issues the same number of pieces -- the counted waits rely on it: when the weight pieces do not split evenly the
  // last ones are fetched twice into pad slots)
  constexpr int STAGE = (TA * 3 + LWW * NW) * 1024;
  static_assert((TA * 3) % NW == 0, "the context pieces of a stage must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char x6_lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int m_tile, n_tile, kt_begin = 0, nk = g.KT, z = 0;
  if (MODE == 0) {
    const int id = blockIdx.x, xcd = id & 7, seq = id >> 3;
    n_tile = seq % g.ntn; m_tile = (seq / g.ntn) * 8 + xcd;              // the column tiles of one row block back to back on one XCD
    if (m_tile >= g.ntm) return;
  } else {
    // split-k: slice-major runs of consecutive work items per XCD (an XCD streams ~nsplit / 8 slices of both operands)
    const int tiles = g.ntm * g.ntn, total = tiles * g.nsplit, per = (total + 7) >> 3;
    const int lin = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (lin >= total) return;
    z = lin / tiles;
    const int tile = lin - z * tiles;
    m_tile = tile % g.ntm; n_tile = tile / g.ntm;
    kt_begin = z * g.kslice;
    nk = min(g.kslice, g.KT - kt_begin);
  }
  const i32x4 rsA = make_rsrc(g.Ap, (unsigned)min((size_t)0xfffffff0u, g.a_bytes));
  const i32x4 rsW = make_rsrc(g.Wp, (unsigned)min((size_t)0xfffffff0u, g.w_bytes));
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)x6_lds);
  const int voff = lane << 4;
  const unsigned a_step = (unsigned)g.a_rt * 3072u, w_step = (unsigned)g.w_rt * 3072u;   // bytes per k-step of the whole image
  const unsigned a_tile = (unsigned)kt_begin * a_step + (unsigned)(m_tile * TA) * 3072u;
  const unsigned w_tile = (unsigned)kt_begin * w_step + (unsigned)(n_tile * TW) * 3072u;

  auto issue = [&](int kt) {
#pragma unroll
    for (int q = 0; q < LW; ++q) {
      if (q < LWA) {
        const int u = wave + NW * q;
        x6_glds16(rsA, lds_base + (unsigned)((kt % S) * STAGE + u * 1024), voff, (unsigned)kt * a_step + a_tile + (unsigned)u * 1024u);
      } else {
        const int u = wave + NW * (q - LWA);
        x6_glds16(rsW, lds_base + (unsigned)((kt % S) * STAGE + (TA * 3 + u) * 1024), voff,
                  (unsigned)kt * w_step + w_tile + (unsigned)min(u, TW * 3 - 1) * 1024u);
      }
    }
  };

  const int wm = wave / NWN, wn = wave % NWN;
  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const unsigned char *fa = x6_lds + (wm * WM * 3) * 1024 + lane * 16;
  const unsigned char *fw = x6_lds + ((TA + wn * WN) * 3) * 1024 + lane * 16;
  auto mfma_set = [&](const x6_bf16x8(&af)[WM], const x6_bf16x8(&wf)[WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
  };
  auto read_a = [&](int kt, int p, x6_bf16x8(&af)[WM]) {
    const unsigned char *sa = fa + (kt % S) * STAGE;
#pragma unroll
    for (int i = 0; i < WM; ++i) af[i] = *(const x6_bf16x8 *)(sa + (i * 3 + p) * 1024);
  };
  auto read_w = [&](int kt, int p, x6_bf16x8(&wf)[WN]) {
    const unsigned char *sw = fw + (kt % S) * STAGE;
#pragma unroll
    for (int j = 0; j < WN; ++j) wf[j] = *(const x6_bf16x8 *)(sw + (j * 3 + p) * 1024);
  };
  // this wave's share of a stage has landed when at most the loads of `behind` later stages are outstanding (loads retire in order)
  auto wait_behind = [&](int behind) {
    if (S >= 3 && behind >= 2) x6_wait_vmcnt<2 * LW>();
    else if (S >= 2 && behind >= 1) x6_wait_vmcnt<LW>();
    else x6_wait_vmcnt<0>();
  };

  // The barrier sits in the MIDDLE of a k-step.  A step's six products run as two halves: {l h, m m, m h} needs the context planes
  // l, m and the weight planes h, m; {h l, h m, h h} the context plane h and all weight planes.  When a wave arrives at the barrier
  // of step kt it has issued the first half, every fragment of step kt is in its registers (slot kt % S is free for step kt + S)
  // and its own share of step kt + 1 has landed; behind the barrier it requests the first-half fragments of step kt + 1 and hides
  // their LDS round trip under the second half of step kt -- no LDS latency is exposed behind a barrier.
#pragma unroll
  for (int t = 0; t < S; ++t)
    if (t < nk) issue(t);
  wait_behind(min(S - 1, nk - 1));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  x6_bf16x8 al[WM], am[WM], ah[WM], wh[2][WN], wmid[2][WN], wl[WN];
  read_a(0, 2, al); read_a(0, 1, am); read_w(0, 0, wh[0]); read_w(0, 1, wmid[0]);
  auto step = [&](int kt, x6_bf16x8(&whc)[WN], x6_bf16x8(&wmc)[WN], x6_bf16x8(&whn)[WN], x6_bf16x8(&wmn)[WN]) {
    mfma_set(al, whc);                      // (its operands crossed the back edge: the wait in front of it is for reads long landed)
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABL & 8) || kt == 0) { read_a(kt, 0, ah); read_w(kt, 2, wl); }
    __builtin_amdgcn_sched_barrier(0);
    mfma_set(am, wmc); mfma_set(am, whc);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk) {
      if (!(ABL & 4)) {
        wait_behind(min(S - 2, nk - 2 - kt));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (!(ABL & 2) && kt + S < nk) issue(kt + S);
      if (!(ABL & 8)) { read_a(kt + 1, 2, al); read_a(kt + 1, 1, am); read_w(kt + 1, 0, whn); read_w(kt + 1, 1, wmn); }
      else {
#pragma unroll
        for (int j = 0; j < WN; ++j) { whn[j] = whc[j]; wmn[j] = wmc[j]; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_set(ah, wl); mfma_set(ah, wmc); mfma_set(ah, whc);
    __builtin_amdgcn_sched_barrier(0);
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    step(kt, wh[0], wmid[0], wh[1], wmid[1]);
    step(kt + 1, wh[1], wmid[1], wh[0], wmid[0]);
  }
  if (kt < nk) step(kt, wh[0], wmid[0], wh[1], wmid[1]);

  // epilogue.  D of mfma(W fragment, A fragment): row (= output column within the 32-block) (r & 3) + 8 (r >> 2) + 4 (lane >> 5),
  // column (= output row) lane & 31
  const int m0 = m_tile * TA * 32, n0 = n_tile * TW * 32;
  float *out = MODE == 0 ? g.C : g.C + (long)z * g.M * g.ldc;           // split-k: slice z's partial, (M, ldc) with ldc = the padded width
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int m = m0 + (wm * WM + i) * 32 + (lane & 31);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + (wn * WN + j) * 32 + 8 * q + 4 * (lane >> 5);
        if (n >= g.N || ((ABL & 1) && g.alpha != 12345.0f)) continue;
        f32x4 o;
        if (MODE == 0) {
          const f32x4 bv = g.bias ? *(const f32x4 *)&g.bias[n] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * q + e] * g.alpha + bv[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * q + e];
        }
        *(f32x4 *)&out[(long)m * g.ldc + n] = o;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Long-contraction TN product G[i, j] = sum_r A[r, i] B[r, j] (the patch-bag weight gradient G = dKV^T z, healnet.py:405's autograd:
// 1024 x 773 over 32 768 rows) is the SAME kernel on TRANSPOSED plane images -- "rows" = the columns i (j) of A (B), k = the row
// index r -- cut into k-slices (MODE 1) whose partials x6_tn_reduce_kernel folds in a fixed order.
//   * x6_split_t_kernel builds such an image: lane (column c = ct * 32 + (lane & 31), rows r0 + (lane >> 5) * 8 + 0 .. 7) gathers its
//     eight values with eight dword loads (each a pair of full 128-byte segments per wave) and writes three 16-byte pieces.
//   * colsum_i = sum_r A[r, i] (the bias-gradient / LayerNorm-beta term) is one more column of the product: the image of B carries a
//     synthetic ONES column at index N (its pad), so G[:, N] is the column sum, accumulated like everything else.
// ------------------------------------------------------------------------------------------------
struct X6SplitTArgs {
  const float *X; long ldx;
  long R; int C;                      // X is (R, C); the image has KT = ceil(R / 16) k-steps of Ct column tiles
  int Ct, KT, ones_col;               // ones_col >= C: that image row is all ones (h = 1, m = l = 0) for r < R; -1: none
  int pair;                           // k index in pair order (x6_pair_order, common.h)
  unsigned short *P;
};

__global__ __launch_bounds__(256) void x6_split_t_kernel(X6SplitTArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ct = blockIdx.x, kt0 = (blockIdx.y * 4 + wave) * 4;
  if (kt0 >= g.KT) return;
  const int col = ct * 32 + (lane & 31);
  const bool live = col < g.C, ones = col == g.ones_col;
  const float *x = g.X + (live ? col : 0);
  float v[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int kt = kt0 + q, h = lane >> 5;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long r = g.pair ? (long)(kt >> 1) * 32 + 8 * (kt & 1) + 4 * h + (e & 3) + 16 * (e >> 2) : (long)kt * 16 + h * 8 + e;
      v[q][e] = (live && r < g.R) ? x[r * g.ldx] : ((ones && r < g.R) ? 1.0f : 0.0f);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (kt0 + q >= g.KT) break;
    u32x4 h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned hh, mm, ll;
      x6_split2(v[q][2 * e], v[q][2 * e + 1], hh, mm, ll);
      h[e] = hh; m[e] = mm; l[e] = ll;
    }
    u32x4 *dst = (u32x4 *)(g.P + ((((long)(kt0 + q) * g.Ct + ct) * 3) * 64 + lane) * 8);
    dst[0] = h; dst[64] = m; dst[128] = l;
  }
}

// G[i, j] = sum_z part[z][i][j] (j < N), colsum[i] = sum_z part[z][i][N]; one thread per four columns
__global__ __launch_bounds__(256) void x6_tn_reduce_kernel(const float *__restrict__ part, int nsplit, int M, int N, int ldp, float *__restrict__ G,
                                                           long ldg, float *__restrict__ colsum) {
  const int q4 = ldp >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)M * q4) return;
  const int i = (int)(idx / q4), j = (int)(idx % q4) * 4;
  if (j > N) return;
  f32x4 acc = *(const f32x4 *)&part[(long)i * ldp + j];
  for (int z = 1; z < nsplit; ++z) acc += *(const f32x4 *)&part[((long)z * M + i) * ldp + j];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (j + e < N) G[(long)i * ldg + j + e] = acc[e];
    else if (j + e == N && colsum) colsum[i] = acc[e];
  }
}

bool gemm_x6_enabled() {
  static const bool off = getenv("HN_NO_X6_GEMM") != nullptr;      // route switch (A/B): the fp32-MFMA kernels
  return !off;
}
static bool x6_forced() {
  static const bool on = getenv("HN_FORCE_X6_GEMM") != nullptr;    // route switch (A/B tests): this route below its size gates
  return on;
}
// large row counts only.  Measured at N = 1024, K = 773 (tools/ubench/gemm_x6_bench.hip), fp32 MFMA / 256 x 256 tiles at one
// workgroup per CU / 256 x 128 tiles at two: 32 768 rows 425 / 232 / 256 us, 12 288 rows 200 / 103 / 109, 8192 rows 106 / 98 / 70,
// 4096 rows 63 / 93 / 62 (+ the image of the bag: 14 us per 8192 rows, once per forward) -- from 8192 rows on this route wins,
// below 12 288 rows on the smaller tile (gemm_nt_x6_variant)
bool gemm_nt_x6_eligible(long M, int N, int K) {
  const bool big = x6_forced() ? (M >= 64 && N >= 32 && K >= 8) : (M >= 8192 && N >= 256 && K >= 64);
  return gemm_x6_enabled() && big && N % 4 == 0 && x6_plane_bytes(M, K, X6_ROW_TILE) < 0xfffffff0u;
}

int gemm_nt_x6_variant(long M) { return M < 12288 ? 1 : 0; }

size_t x6_plane_bytes(long rows, int K, int row_tile) {
  const long rt = (rows + 31) / 32, rtp = (rt + row_tile - 1) / row_tile * row_tile;
  return (size_t)((K + 15) / 16) * (size_t)rtp * 3072;
}

int launch_x6_split(const float *X, long ldx, const float *scale, long R, int K, int row_tile, unsigned short *P, hipStream_t s,
                    int col_group, int col_group_pitch) {
  HN_REQUIRE(X && P, HN_E_NULL, "x6_split: NULL operand");
  X6SplitArgs a{};
  a.X = X; a.ldx = ldx; a.scale = scale; a.R = (int)R; a.K = K; a.P = P;
  const long rt = (R + 31) / 32;
  a.Rt = (int)((rt + row_tile - 1) / row_tile * row_tile);
  a.KT = (K + 15) / 16;
  a.col_group = col_group; a.col_group_pitch = col_group_pitch;
  KernelTimerScope timer("x6_split", s);
  hipLaunchKernelGGL(x6_split_kernel, dim3((unsigned)a.Rt, (unsigned)ceil_div(a.KT, 16)), dim3(256), 0, s, a);
  HN_LAUNCH_CHECK("x6_split");
  return HN_OK;
}

// the dynamic-LDS attribute of a kernel is a per-device property: one flag per device ordinal and entry point (a benign race: two
// threads of a first call may both set it)
static bool *x6_attr_flag(bool (&flags)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  return &flags[dev];
}
static int x6_set_lds(const void *fn, int bytes) {
  HN_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return HN_OK;
}

int launch_gemm_nt_x6(const GemmX6Args &g_in, int variant, hipStream_t s) {
  GemmX6Args g = g_in;
  HN_REQUIRE(g.Ap && g.Wp && g.C, HN_E_NULL, "gemm_nt_x6: NULL operand");
  HN_REQUIRE((g.ldc & 3) == 0 && ((uintptr_t)g.C & 15) == 0 && (g.N & 3) == 0, HN_E_SHAPE, "gemm_nt_x6: output rows must be 16-byte aligned");
  const int bm = 256, bn = variant == 1 ? 128 : 256;
  g.ntm = ceil_div(g.M, bm); g.ntn = ceil_div(g.N, bn);
  HN_REQUIRE(g.a_rt % (bm / 32) == 0 && g.a_rt >= g.ntm * (bm / 32) && g.w_rt % (bn / 32) == 0 && g.w_rt >= g.ntn * (bn / 32), HN_E_SHAPE,
             "gemm_nt_x6: plane images are not padded to the tile (%d, %d row tiles)", g.a_rt, g.w_rt);
  g.a_bytes = (size_t)g.KT * g.a_rt * 3072; g.w_bytes = (size_t)g.KT * g.w_rt * 3072;
  HN_REQUIRE(g.a_bytes < 0xfffffff0u && g.w_bytes < 0xfffffff0u, HN_E_UNSUPPORTED, "gemm_nt_x6: plane image beyond 4 GB");
  g.nsplit = 1; g.kslice = g.KT;
  const long blocks = (long)ceil_div(g.ntm, 8) * 8 * g.ntn;
  KernelTimerScope timer("gemm_nt_x6", s);
  static bool attr_dev[64] = {};
  bool &attr = *x6_attr_flag(attr_dev);
  if (!attr) {
    int rc;
    if ((rc = x6_set_lds((const void *)gemm_nt_x6_kernel<4, 2, 2, 4, 3>, 3 * 48 * 1024)) != HN_OK) return rc;
    if ((rc = x6_set_lds((const void *)gemm_nt_x6_kernel<4, 2, 2, 2, 2>, 2 * 36 * 1024)) != HN_OK) return rc;
#ifdef HN_GEMM_NT_BENCH
    if ((rc = x6_set_lds((const void *)gemm_nt_x6_kernel<4, 2, 2, 4, 3, 0, 1>, 3 * 48 * 1024)) != HN_OK) return rc;
    if ((rc = x6_set_lds((const void *)gemm_nt_x6_kernel<4, 2, 2, 4, 3, 0, 2>, 3 * 48 * 1024)) != HN_OK) return rc;
    if ((rc = x6_set_lds((const void *)gemm_nt_x6_kernel<4, 2, 2, 4, 3, 0, 4>, 3 * 48 * 1024)) != HN_OK) return rc;
    if ((rc = x6_set_lds((const void *)gemm_nt_x6_kernel<4, 2, 2, 4, 3, 0, 15>, 3 * 48 * 1024)) != HN_OK) return rc;
#endif
    attr = true;
  }
  switch (variant) {
    case 0: hipLaunchKernelGGL((gemm_nt_x6_kernel<4, 2, 2, 4, 3>), dim3((unsigned)blocks), dim3(512), 3 * 48 * 1024, s, g); break;
    case 1: hipLaunchKernelGGL((gemm_nt_x6_kernel<4, 2, 2, 2, 2>), dim3((unsigned)blocks), dim3(256), 2 * 36 * 1024, s, g); break;
#ifdef HN_GEMM_NT_BENCH
    case 10: hipLaunchKernelGGL((gemm_nt_x6_kernel<4, 2, 2, 4, 3, 0, 1>), dim3((unsigned)blocks), dim3(512), 3 * 48 * 1024, s, g); break;
    case 11: hipLaunchKernelGGL((gemm_nt_x6_kernel<4, 2, 2, 4, 3, 0, 2>), dim3((unsigned)blocks), dim3(512), 3 * 48 * 1024, s, g); break;
    case 12: hipLaunchKernelGGL((gemm_nt_x6_kernel<4, 2, 2, 4, 3, 0, 4>), dim3((unsigned)blocks), dim3(512), 3 * 48 * 1024, s, g); break;
    case 13: hipLaunchKernelGGL((gemm_nt_x6_kernel<4, 2, 2, 4, 3, 0, 15>), dim3((unsigned)blocks), dim3(512), 3 * 48 * 1024, s, g); break;
#endif
    default: return fail(HN_E_UNSUPPORTED, "gemm_nt_x6: variant %d", variant);
  }
  HN_LAUNCH_CHECK("gemm_nt_x6");
  return HN_OK;
}

int launch_x6_split_t(const float *X, long ldx, long R, int C, int col_tile, int ones_col, unsigned short *P, hipStream_t s) {
  HN_REQUIRE(X && P, HN_E_NULL, "x6_split_t: NULL operand");
  X6SplitTArgs a{};
  a.X = X; a.ldx = ldx; a.R = R; a.C = C; a.ones_col = ones_col; a.P = P;
  a.Ct = x6_col_tiles(C + (ones_col >= 0 ? 1 : 0), col_tile);
  HN_REQUIRE(ones_col < 0 || (ones_col >= C && ones_col < a.Ct * 32), HN_E_SHAPE, "x6_split_t: ones column %d", ones_col);
  a.KT = (int)((R + 15) / 16);
  a.pair = x6_pair_order(R) ? 1 : 0;
  KernelTimerScope timer("x6_split_t", s);
  hipLaunchKernelGGL(x6_split_t_kernel, dim3((unsigned)a.Ct, (unsigned)ceil_div(a.KT, 16)), dim3(256), 0, s, a);
  HN_LAUNCH_CHECK("x6_split_t");
  return HN_OK;
}

// the TN tile: 256 (i) x 160 (j), 8 waves of 32 x 160 -- 773 + 1 columns are 25 tiles of 32 = 5 x 160 exactly
constexpr int X6_TN_TI = 8, X6_TN_TJ = 5;
bool gemm_tn_x6_eligible(long K, int M, int N) {
  // (8192 rows: 78 us + the two images against 131 us on the fp32 MFMA; 12 288: 110 against 185)
  const bool big = x6_forced() ? (K >= 64 && M >= 32 && N >= 8) : (K >= 8192 && M >= 256 && N >= 64);
  return gemm_x6_enabled() && big &&
         (size_t)((K + 15) / 16) * x6_col_tiles(M, X6_TN_TI) * 3072 < 0xfffffff0u && (size_t)((K + 15) / 16) * x6_col_tiles(N + 1, X6_TN_TJ) * 3072 < 0xfffffff0u;
}
size_t gemm_tn_x6_image_bytes(long K, int cols, int col_tile) { return (size_t)((K + 15) / 16) * x6_col_tiles(cols, col_tile) * 3072; }

// G (M, N) = A^T B and colsum (M) = column sums of A from the transposed images At (x6_col_tiles(M, 8) tiles) and Bt (x6_col_tiles(N + 1, 5)
// tiles, ones column at N); `scratch` holds the k-slice partials
int launch_gemm_tn_x6(const unsigned short *At, const unsigned short *Bt, long K, int M, int N, float *G, long ldg, float *colsum, float *scratch,
                      size_t scratch_floats, hipStream_t s) {
  HN_REQUIRE(At && Bt && G && scratch, HN_E_NULL, "gemm_tn_x6: NULL operand");
  GemmX6Args g{};
  g.Ap = At; g.a_rt = x6_col_tiles(M, X6_TN_TI); g.Wp = Bt; g.w_rt = x6_col_tiles(N + 1, X6_TN_TJ);
  g.KT = (int)((K + 15) / 16);
  g.M = M; g.N = g.w_rt * 32; g.ldc = g.w_rt * 32; g.C = scratch; g.alpha = 1.0f;
  g.ntm = g.a_rt / X6_TN_TI; g.ntn = g.w_rt / X6_TN_TJ;
  g.a_bytes = (size_t)g.KT * g.a_rt * 3072; g.w_bytes = (size_t)g.KT * g.w_rt * 3072;
  HN_REQUIRE(g.a_bytes < 0xfffffff0u && g.w_bytes < 0xfffffff0u, HN_E_UNSUPPORTED, "gemm_tn_x6: plane image beyond 4 GB");
  // one round of workgroups (1 per CU): as many slices as fit 256 workgroups, at least 16 k-steps each
  const int tiles = g.ntm * g.ntn;
  int nsplit = 256 / tiles;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > g.KT / 16) nsplit = g.KT / 16 > 0 ? g.KT / 16 : 1;
  while (nsplit > 1 && (size_t)nsplit * M * g.ldc > scratch_floats) --nsplit;
  HN_REQUIRE((size_t)nsplit * M * g.ldc <= scratch_floats, HN_E_WORKSPACE, "gemm_tn_x6: scratch %zu floats < one partial", scratch_floats);
  g.kslice = ceil_div(g.KT, nsplit);
  g.nsplit = ceil_div(g.KT, g.kslice);
  const int total = tiles * g.nsplit, blocks = ceil_div(total, 8) * 8;
  constexpr int LDS = 3 * (X6_TN_TI * 3 + 16) * 1024;
  static bool attr_dev[64] = {};
  bool &attr = *x6_attr_flag(attr_dev);
  if (!attr) {
    int rc = x6_set_lds((const void *)gemm_nt_x6_kernel<1, X6_TN_TJ, X6_TN_TI, 1, 3, 1>, LDS);
    if (rc != HN_OK) return rc;
    attr = true;
  }
  {
    KernelTimerScope timer("gemm_tn_x6", s);
    hipLaunchKernelGGL((gemm_nt_x6_kernel<1, X6_TN_TJ, X6_TN_TI, 1, 3, 1>), dim3((unsigned)blocks), dim3(512), LDS, s, g);
    HN_LAUNCH_CHECK("gemm_tn_x6");
  }
  KernelTimerScope timer("x6_tn_reduce", s);
  const long threads = (long)M * (g.ldc / 4);
  hipLaunchKernelGGL(x6_tn_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, scratch, g.nsplit, M, N, (int)g.ldc, G, ldg, colsum);
  HN_LAUNCH_CHECK("x6_tn_reduce");
  return HN_OK;
}

}  // namespace hn
