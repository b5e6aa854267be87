// fp32 NT GEMM for the patch-bag K/V projection (healnet/models/healnet.py:405, `to_kv` on a (b * N, D) context: 32 768 x 773 ->
// 1024 at BASELINE configs[3], 52 GF per block of the model) on the exact fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
//   C[m, col(n)] = alpha * sum_k A[m, k] * Ws[n, k] + bs[n]
//
// with the context LayerNorm's affine half (PreNorm.norm_context :316-319) FOLDED into the staged weight once per call
// (gemm_nt_stage_kernel):  (z * gamma + beta) W^T = z (W * gamma)^T + W beta, so the loader moves raw operand bytes and no vector
// instruction touches them.
//
// What the round-3 kernel (gemm_big_kernel, gemm.hip) spent outside its MFMAs, by ablation: LDS stores 50 us,
// global loads 38 us, barriers 29 us, epilogue 60 us of 475.  This kernel removes the first, thins the rest:
//   * operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write) in full 128-byte
//     lines; the XOR swizzle of the LDS image (16-byte slot ^= row & 7) is applied on the SOURCE address, the destination of a
//     wave's load is linear (M0 + 16 * lane), the fragment reads apply the same involution: conflict-free ds_read_b128;
//   * an S-stage ring with ONE raw s_barrier per 32-wide k-tile and counted vmcnt: the loads of tile t + S - 1 are issued right
//     behind the barrier that frees their slot and stay in flight across the following barriers;
//   * the MFMA takes the weight rows as its A operand and the context rows as B, so a lane owns FOUR CONSECUTIVE output columns
//     of one row: the epilogue is 16 dwordx4 stores per wave instead of 64 dword stores;
//   * K = 773 runs as 24 full k-tiles + ONE 16-wide step (784 columns instead of 800): the staged weight is zero beyond K.
#include "common.h"
#pragma clang diagnostic ignored "-Winline-asm"      // (M0 on the clobber list of hn_glds16: nothing else in these kernels uses it)

namespace hn {

// LDS-DMA, 16 bytes per lane: lane l of the wave lands at LDS byte address `lds_byte` + 16 l (wave-uniform, through M0), fetched from
// rsrc base + voffset (per lane) + soffset (uniform), zero when out of the descriptor's range.  Issued from INLINE ASM on purpose:
// through the LLVM intrinsic hipcc tracks the pending LDS write and waits vmcnt(0) in front of the first ds_read behind every issue
// -- whenever the instruction still carries its memory operand, which depends on unrelated code around it (the NT kernel below was
// spared by luck, the TN kernel was not: 459 us instead of 4xx) -- and there is no way to tell it that a barrier protocol orders
// the two.  From asm the compiler sees no memory access: every wait on these loads is a counted s_waitcnt written by hand.
__device__ __forceinline__ void hn_glds16(const i32x4 &rsrc, unsigned lds_byte, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_byte), "v"(voffset), "s"(rsrc), "s"(soffset)
               : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_byte_address(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Ws[n', k] = W[n, k] * gamma[k] for k < K, 0 for K <= k < ldws;  bs[n'] = sum_k W[n, k] * beta[k] (+ bias[n]).  One wave per row.
// Head re-pitching happens HERE: with col_group > 0 the staged image has col_group_pitch rows per group of col_group source rows
// (n' = (n / col_group) * pitch + n % col_group; the pad rows and their bias entries are zero), so the GEMM behind it writes a
// dense, 16-byte aligned row whatever the head width (27, 63, 103: the reference's tuned shapes) and the pad columns of K / V
// come out as zeros without a fill launch.
__global__ __launch_bounds__(256) void gemm_nt_stage_kernel(const float *__restrict__ W, long ldw, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, const float *__restrict__ bias, int Np, int K,
                                                            float *__restrict__ Ws, int ldws, float *__restrict__ bs, int col_group,
                                                            int col_group_pitch) {
  const int lane = threadIdx.x & 63, np = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (np >= Np) return;
  int n = np;
  bool pad = false;
  if (col_group > 0) { const int c = np % col_group_pitch; pad = c >= col_group; n = (np / col_group_pitch) * col_group + (pad ? 0 : c); }
  if (pad) {
    for (int k = lane; k < ldws; k += 64) Ws[(long)np * ldws + k] = 0.0f;
    if (lane == 0) bs[np] = 0.0f;
    return;
  }
  const float *w = W + (long)n * ldw;
  float *o = Ws + (long)np * ldws;
  float acc = 0.0f;
  // four column groups of 64 per trip, their loads issued together (one dependent round trip per 256 columns instead of per 64)
  for (int k0 = 0; k0 < ldws; k0 += 256) {
    float x[4], gm[4], bt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + 64 * u + lane, kc = k < K ? k : K - 1;
      x[u] = w[kc];
      gm[u] = gamma ? gamma[kc] : 1.0f;
      bt[u] = beta ? beta[kc] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + 64 * u + lane;
      if (k < ldws) o[k] = k < K ? x[u] * gm[u] : 0.0f;
      if (k < K) acc += x[u] * bt[u];
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
  if (lane == 0) bs[np] = acc + (bias ? bias[n] : 0.0f);
}

// ABL: development ablations (tools/ubench/gemm_f32_bench.hip): 1 = no output stores, 2 = no operand loads behind the prologue,
// 4 = no barrier / load waits in the loop.  The product instantiates ABL = 0 only.
template <int WM, int WN, int NWM, int NWN, int S, int ABL = 0>
__global__ __launch_bounds__(NWM *NWN * 64) void gemm_nt_glds_kernel(GemmNtArgs g) {
  constexpr int NW = NWM * NWN, BM = NWM * WM * 16, BN = NWN * WN * 16;
  constexpr int STAGE = (BM + BN) * 32;                  // floats per ring slot: A rows then W rows, 32 floats (one 128-byte line) each
  constexpr int LA = BM / 8, LB = BN / 8;                // 1 KB load instructions per tile (8 rows each)
  constexpr int LW = (LA + LB) / NW, LWA = LA / NW;      // per wave; the first LWA of them fetch A
  static_assert(LA % NW == 0 && LB % NW == 0, "tile rows must split evenly over the waves");
  __shared__ __attribute__((aligned(16))) float lds[S * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id = blockIdx.x, xcd = id & 7, seq = id >> 3;
  const int n_tile = seq % g.ntn, m_tile = (seq / g.ntn) * 8 + xcd;      // the column tiles of one row block back to back on one XCD
  if (m_tile >= g.ntm) return;
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int lda = (int)g.lda, ldw = (int)g.ldw;
  const int rows_a = min(BM, g.M - m0), rows_w = min(BN, g.N - n0);
  // descriptors based at the tile: rows past the end read as zero, offsets stay small whatever M
  const int a_bytes = rows_a * lda * 4, w_bytes = rows_w * ldw * 4;
  const i32x4 rsA = make_rsrc(g.A + (long)m0 * lda, (unsigned)a_bytes);
  const i32x4 rsW = make_rsrc(g.W + (long)n0 * ldw, (unsigned)w_bytes);

  // loader: lane (r8 = lane >> 3, p = lane & 7) fetches the 16-byte slot p ^ r8 of row 8 u + r8 and lands at slot p
  const int r8 = lane >> 3, p = lane & 7;
  const int voffA = r8 * lda * 4 + ((p ^ r8) << 4), voffW = r8 * ldw * 4 + ((p ^ r8) << 4);
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_byte_address(lds));
  auto issue = [&](int kt) {
#pragma unroll
    for (int q = 0; q < LW; ++q) {
      const int u = wave + NW * q;
      const unsigned dst = lds_base + (unsigned)(((kt % S) * STAGE + u * 256) * 4);
      // (the scalar offset must stay inside the descriptor -- the range check subtracts it from num_records: clamped, so a piece that
      // starts past the last row of a ragged tile reads zeros instead of wrapping)
      if (q < LWA) hn_glds16(rsA, dst, voffA, min(8 * u * lda * 4 + kt * 128, a_bytes));
      else hn_glds16(rsW, dst, voffW, min(8 * (u - LA) * ldw * 4 + kt * 128, w_bytes));
    }
  };

  const int wm = wave / NWN, wn = wave % NWN;
  const int fi = lane & 15, fg = lane >> 4;
  // fragment reads: row (block * 16 + fi), logical slot 4 s2 + fg -> physical slot ^ (row & 7) = ^ (fi & 7)
  const int sl0 = (fg ^ (fi & 7)) << 2;                            // floats; the s2 = 1 slot is sl0 ^ 16
  const int a_base = (wm * WM * 16 + fi) * 32, w_base = (BM + wn * WN * 16 + fi) * 32;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  struct Frags { f32x4 a[WM], w[WN]; };
  auto read_frags = [&](int kt, int s2, Frags &f) {                // 16 columns of k-tile kt: one ds_read_b128 per 16-row block
    const float *st = lds + (kt % S) * STAGE;
    const int sl = sl0 ^ (s2 << 4);
#pragma unroll
    for (int i = 0; i < WM; ++i) f.a[i] = *(const f32x4 *)&st[a_base + i * 512 + sl];
#pragma unroll
    for (int j = 0; j < WN; ++j) f.w[j] = *(const f32x4 *)&st[w_base + j * 512 + sl];
  };
  auto mfma_step = [&](const Frags &f) {                           // 4 MFMAs per accumulator
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[j][e], f.a[i][e], acc[i][j], 0, 0, 0);
  };

  // bias row of this wave's columns (staged: W beta), requested before the loop
  f32x4 bv[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = n0 + wn * WN * 16 + j * 16 + 4 * fg;
    bv[j] = (g.bias && n < g.N) ? *(const f32x4 *)&g.bias[n] : (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int nk = (g.K + 31) >> 5;
  const int tail_steps = ((g.K - (nk - 1) * 32) + 15) >> 4;       // 1 or 2 sixteen-column steps in the last k-tile
#pragma unroll
  for (int t = 0; t < S; ++t)
    if (t < nk) issue(t);
  // The barrier sits in the MIDDLE of a k-tile: when a wave arrives it has read both fragment sets of tile kt (slot kt % S is free
  // for the loads of tile kt + S) and its own share of tile kt + 1 has landed; behind it the first fragments of tile kt + 1 are
  // requested under the second half of tile kt's MFMAs -- no LDS round trip is exposed behind a barrier.
  // Counted waits.  Loads retire in order, so "this wave's share of tile t has landed" == "at most the loads of the tiles issued
  // behind t are outstanding".  In front of the first barrier tiles 0 .. S-1 are issued (tile 0 needed: up to S-1 tiles may stay in
  // flight); at the mid-tile barrier of tile kt tiles .. kt+S-1 are issued and tile kt+1 is needed: up to S-2 tiles (none with two
  // slots: a tile's loads then have exactly one tile time -- ~2 us of MFMAs -- to land).
  auto wait_outstanding = [&](int tiles) {
    if (S >= 3 && tiles >= 2) wait_vmcnt<2 * LW>();
    else if (S >= 2 && tiles == 1) wait_vmcnt<LW>();
    else wait_vmcnt<0>();
  };
  auto wait_tile = [&](int kt) { wait_outstanding(min(S - 2, nk - 1 - kt)); };
  wait_outstanding(min(S - 1, nk - 1));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Frags f0, f1;
  read_frags(0, 0, f0);
  for (int kt = 0; kt < nk; ++kt) {
    const bool last = kt + 1 == nk, two = !last || tail_steps == 2;
    if (two) read_frags(kt, 1, f1);
    if (last) {
      // columns K .. of the last step(s) belong to the row's pad / the next row: whatever they hold (NaNs included) must not reach
      // the accumulators -- the A fragment is masked in registers (once per tile; the staged weight is zero there as well)
      const int kc = kt * 32 + 4 * fg;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f0.a[i][e] = kc + e < g.K ? f0.a[i][e] : 0.0f;
          if (two) f1.a[i][e] = kc + 16 + e < g.K ? f1.a[i][e] : 0.0f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(f0);
    __builtin_amdgcn_sched_barrier(0);
    if (!last) {
      if (!(ABL & 4)) {
        wait_tile(kt + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (!(ABL & 2) && kt + S < nk) issue(kt + S);
      read_frags(kt + 1, 0, f0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (two) mfma_step(f1);
  }

  // epilogue.  D of mfma(W fragment, A fragment): row (= output column within the 16-block) 4 fg + r, column (= output row) fi
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int m = m0 + wm * WM * 16 + i * 16 + fi;
    if (m < g.M) {
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = n0 + wn * WN * 16 + j * 16 + 4 * fg;
        if (n < g.N && (!(ABL & 1) || g.alpha == 12345.0f)) {
          const long oc = g.col_group > 0 ? (long)(n / g.col_group) * g.col_group_pitch + (n % g.col_group) : n;
          *(f32x4 *)&g.C[(long)m * g.ldc + oc] = acc[i][j] * g.alpha + bv[j];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TN product on the same staging: C[i, j] = sum_r A[r, i] * B[r, j] over a LONG contraction (the patch-bag weight gradient
// G = dKV^T z of healnet.py:405's autograd: 1024 x 773 over 32 768 rows, 52 GF per block), split over row slices into partials that
// gemm_tn_reduce_kernel folds in a fixed order.  Both operands are contraction-major, so a k-tile is 32 rows of 512 contiguous
// bytes per operand, landed by LDS-DMA as it lies ([k][128] images, no swizzle: the fragment reads below are conflict-free or
// 2-way at a rate where it does not matter).
//   * 773 = 6 * 128 + 5: on 128-wide tiles the seventh column tile is almost empty (7.6 % of the round-3 kernel's MFMAs).  Here a
//     tile is 128 (i) x NB * 16 (j) with NB = 7: 7 x 112 = 784 columns, 1.4 % padding; the loads still fetch 128 columns per row.
//   * a wave owns 32 (i) x 112 (j): per 4-row step ONE ds_read_b64 (i = 32 w + 2 fi + e: 2 blocks), one ds_read_b128 (j = 4 fi + e:
//     4 blocks) and 3 ds_read_b32 (j = 64 + 16 e + fi) feed 14 MFMAs; the MFMA takes the B fragment as its A operand, so a lane
//     ends up with FOUR CONSECUTIVE j of one i: dwordx4 partial stores (partial pitch = tiles * 112, 16-byte aligned).
//   * colsum_i = sum_r A[r, i] (the bias-gradient / LayerNorm-beta term) rides along in the first column tile: its threads re-read
//     the landed A tile from LDS (4 ds_read_b128 + 16 adds per thread and k-tile).
// ------------------------------------------------------------------------------------------------
struct GemmTnGArgs {
  const float *A; long lda;        // (K, M)
  const float *B; long ldb;        // (K, N)
  float *part; long ldp;           // (nsplit, M, ldp) partials
  float *cs_part;                  // (nsplit * 8, M) column-sum partials or NULL
  int M, N, K, kslice, nsplit, ntm, ntn;
};

// MAP: 0 = tile index fastest (a column block of A stays on one XCD, every XCD streams all of B), 1 = slice-major runs of
// consecutive work items per XCD (an XCD owns ~1 row slice: each operand row enters one L2).  ABL: ablations as above (bench only).
template <int NB, int MAP = 0, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_tn_glds_kernel(GemmTnGArgs g) {
  constexpr int S = 2, STAGE = 2 * 32 * 128, LW = 8;       // floats per ring slot: A tile [32][128] then B tile [32][128]
  constexpr int BNT = NB * 16;
  __shared__ __attribute__((aligned(16))) float lds[S * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, z;
  if (MAP == 0) { tile = blockIdx.x; z = blockIdx.y; }
  else {
    const int tiles = g.ntm * g.ntn, total = tiles * g.nsplit, per = (total + 7) >> 3;
    const int lin = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (lin >= total) return;
    z = lin / tiles; tile = lin - z * tiles;
  }
  const int m_tile = tile % g.ntm, n_tile = tile / g.ntm;
  const int m0 = m_tile * 128, n0 = n_tile * BNT;
  const int k_begin = z * g.kslice, rows = min(g.K, k_begin + g.kslice) - k_begin;
  const int lda = (int)g.lda, ldb = (int)g.ldb;
  // descriptors based at (slice row 0, tile column 0): rows past the slice read as zero
  const long a_bytes = ((long)(rows - 1) * lda + (g.M - m0)) * 4, b_bytes = ((long)(rows - 1) * ldb + (g.N - n0)) * 4;
  const unsigned a_clamp = (unsigned)(a_bytes > 0 ? a_bytes : 0), b_clamp = (unsigned)(b_bytes > 0 ? b_bytes : 0);
  const i32x4 rsA = make_rsrc(g.A + (long)k_begin * lda + m0, a_clamp);
  const i32x4 rsB = make_rsrc(g.B + (long)k_begin * ldb + n0, b_clamp);
  // loader: one instruction = 2 rows x 512 bytes; lane -> (row lane >> 5, 16-byte piece lane & 31)
  const int voffA = (lane >> 5) * lda * 4 + ((lane & 31) << 4), voffB = (lane >> 5) * ldb * 4 + ((lane & 31) << 4);
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_byte_address(lds));
  auto issue = [&](int kt) {
#pragma unroll
    for (int q = 0; q < LW; ++q) {
      const int u = wave + 4 * q;                                 // u < 16: A rows 2u, 2u + 1; else B rows 2(u - 16) ..
      const unsigned dst = lds_base + (unsigned)(((kt % S) * STAGE + u * 256) * 4);
      // (scalar offsets clamped into the descriptor: rows past the slice read zeros, never wrap)
      // (32-bit on purpose: K * ld * 4 < 2^31 is part of the eligibility test, and 64-bit scalar products in front of every piece
      // cost the kernel 8 %: 445 against 411 us)
      if (q < 4) hn_glds16(rsA, dst, voffA, (int)min((unsigned)((kt * 32 + 2 * u) * lda * 4), a_clamp));
      else hn_glds16(rsB, dst, voffB, (int)min((unsigned)((kt * 32 + 2 * (u - 16)) * ldb * 4), b_clamp));
    }
  };
  const int fi = lane & 15, fg = lane >> 4;
  const int a_off = fg * 128 + 32 * wave + 2 * fi;                // + 512 per 4-row step
  const int b_off = 32 * 128 + fg * 128;
  struct Frags { float2 a[4]; f32x4 b[4]; float c[4][NB - 4]; };
  auto read_frags = [&](int kt, int half, Frags &f) {
    const float *st = lds + (kt % S) * STAGE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int step = (half * 4 + q) * 512;
      f.a[q] = *(const float2 *)&st[a_off + step];
      f.b[q] = *(const f32x4 *)&st[b_off + step + 4 * fi];
      if (NB == 8) {
        const f32x4 t = *(const f32x4 *)&st[b_off + step + 64 + 4 * fi];
#pragma unroll
        for (int e = 0; e < NB - 4; ++e) f.c[q][e] = t[e];
      } else {
#pragma unroll
        for (int e = 0; e < NB - 4; ++e) f.c[q][e] = st[b_off + step + 64 + 16 * e + fi];
      }
    }
  };
  f32x4 acc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mfma_half = [&](const Frags &f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float av[2] = {f.a[q].x, f.a[q].y};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[q][j], av[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 4; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.c[q][j - 4], av[i], acc[i][j], 0, 0, 0);
      }
    }
  };
  const bool do_cs = g.cs_part != nullptr && n_tile == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  const int cs_off = (tid >> 5) * 128 + ((tid & 31) << 2);         // rows (tid >> 5) + 8 q of the A tile, 4 columns

  const int nk = (rows + 31) >> 5;
  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  // two slots: the tile needed at a mid-tile barrier is the only one in flight (see gemm_nt_glds_kernel)
  auto wait_tile = [&](int) { wait_vmcnt<0>(); };
  Frags f0, f1;
  if (nk > 0) {
    if (nk > 1) wait_vmcnt<LW>(); else wait_vmcnt<0>();            // tile 0 landed, tile 1 may be in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(0, 0, f0);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const bool last = kt + 1 == nk;
    read_frags(kt, 1, f1);
    if (do_cs) {
      asm volatile("" ::: "memory");
      const float *st = lds + (kt % S) * STAGE;
#pragma unroll
      for (int q = 0; q < 4; ++q) csum += *(const f32x4 *)&st[cs_off + q * 1024];
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(f0);
    __builtin_amdgcn_sched_barrier(0);
    if (!last) {
      if (!(ABL & 4)) {
        wait_tile(kt + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (!(ABL & 2) && kt + S < nk) issue(kt + S);
      read_frags(kt + 1, 0, f0);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_half(f1);
  }

  // 8 row groups x 128 columns of partial column sums straight to the scratch (the reduce folds nsplit * 8 of them per column).
  // NOT through LDS: one ordinary LDS store anywhere in this kernel makes hipcc wait vmcnt(0) in front of the first fragment read
  // behind every LDS-DMA issue (it then sees the ring as written memory the reads may alias) -- measured: 459 us instead of 4xx
  if (do_cs) {
    const int m = m0 + ((tid & 31) << 2);
    if (m < g.M) *(f32x4 *)&g.cs_part[((long)z * 8 + (tid >> 5)) * g.M + m] = csum;
  }
  // D of mfma(B fragment, A fragment): row rho = 4 fg + r <-> j, column fi <-> i.  Blocks 0..3: j = 4 rho + block -> the lane's
  // four blocks hold j = 16 fg + 4 r + {0, 1, 2, 3}; blocks 4..: j = 64 + 16 (block - 4) + 4 fg + {r}
  float *P = g.part + (long)z * g.M * g.ldp;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 32 * wave + 2 * fi + i;
    if (m >= g.M || ((ABL & 1) && g.K != 12345)) continue;
    float *row = P + (long)m * g.ldp + n0;
#pragma unroll
    for (int r = 0; r < 4; ++r) *(f32x4 *)&row[16 * fg + 4 * r] = (f32x4){acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
    if (NB == 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *(f32x4 *)&row[64 + 16 * fg + 4 * r] = (f32x4){acc[i][4][r], acc[i][NB - 3][r], acc[i][NB - 2][r], acc[i][NB - 1][r]};
    } else {
#pragma unroll
      for (int j = 4; j < NB; ++j) *(f32x4 *)&row[64 + 16 * (j - 4) + 4 * fg] = acc[i][j];
    }
  }
}

// C[m, n] (+)= alpha * sum_z part[z, m, n] for n < N; colsum[m] (+)= sum_z cs_part[z, m].  Fixed order: bitwise reproducible.
// blockIdx.y = output row (row M: the column sums), a thread takes four consecutive columns of it: one 16-byte load per slice (the
// partial rows are 16-byte aligned), four slices in flight per trip, no index division.  (One element per thread with a 64-bit
// divide and 16 clamped loads ran at 1.2 TB/s: 24 us for 29 MB.)
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float *__restrict__ part, int nsplit, int M, int N, long ldp,
                                                             float *__restrict__ C, long ldc, float alpha, int accumulate,
                                                             const float *__restrict__ cs_part, float *__restrict__ cs_out, int cs_accumulate) {
  const int m = blockIdx.y;
  if (m >= M) {                                             // column sums: nsplit * 8 row-group partials per column, 256 columns per row of
    const int c = (m - M) * 256 + threadIdx.x;              // blocks, eight independent chains per thread (fixed order)
    if (blockIdx.x != 0 || c >= M) return;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int np = nsplit * 8;
    for (int k = 0; k < np; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += cs_part[(long)(k + u) * M + c];
    }
    const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    cs_out[c] = cs_accumulate ? cs_out[c] + acc : acc;
    return;
  }
  const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (n >= N) return;
  const long slab = (long)M * ldp;
  const float *src = part + (long)m * ldp + n;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 4 <= nsplit; k += 4) {
    const f32x4 v0 = *(const f32x4 *)&src[(long)k * slab], v1 = *(const f32x4 *)&src[(long)(k + 1) * slab];
    const f32x4 v2 = *(const f32x4 *)&src[(long)(k + 2) * slab], v3 = *(const f32x4 *)&src[(long)(k + 3) * slab];
    acc += v0; acc += v1; acc += v2; acc += v3;
  }
  for (; k < nsplit; ++k) acc += *(const f32x4 *)&src[(long)k * slab];
  float *dst = C + (long)m * ldc + n;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n + e < N) dst[e] = accumulate ? dst[e] + alpha * acc[e] : alpha * acc[e];
}

bool gemm_tn_glds_eligible(const float *A, long lda, const float *B, long ldb, int M, int N, int K) {
  return K >= 4096 && M >= 128 && M < 65000 && N >= 112 && M % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
         (long)K * lda * 4 < (1L << 31) && (long)K * ldb * 4 < (1L << 31);
}

// scratch_floats: capacity of `scratch` (partials + column-sum partials)
int launch_gemm_tn_glds(const float *A, long lda, const float *B, long ldb, float *C, long ldc, int M, int N, int K, float alpha,
                        int accumulate, float *scratch, size_t scratch_floats, float *colsum, int colsum_accumulate, hipStream_t s,
                        int variant) {
  HN_REQUIRE(A && B && C && scratch, HN_E_NULL, "gemm_tn_glds: NULL operand");
  GemmTnGArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  // 112- or 128-wide column tiles, whichever pads less
  const int nb = ceil_div(N, 112) * 112 <= ceil_div(N, 128) * 128 ? 7 : 8;
  g.ntm = ceil_div(M, 128); g.ntn = ceil_div(N, nb * 16);
  g.ldp = (long)g.ntn * nb * 16;
  const int tiles = g.ntm * g.ntn;
  int nsplit = 512 / tiles;                                  // 64 KB of LDS: 2 workgroups per CU, one resident round
  if (nsplit < 1) nsplit = 1;
  const size_t per = (size_t)M * g.ldp + (colsum ? 8 * M : 0);
  if ((size_t)nsplit * per > scratch_floats) nsplit = (int)(scratch_floats / per);
  HN_REQUIRE(nsplit >= 1, HN_E_WORKSPACE, "gemm_tn_glds: scratch %zu floats < %zu", scratch_floats, per);
  g.kslice = ceil_div(ceil_div(K, nsplit), 32) * 32;
  g.nsplit = ceil_div(K, g.kslice);
  g.part = scratch;
  g.cs_part = colsum ? scratch + (size_t)g.nsplit * M * g.ldp : nullptr;
  const dim3 grid1((unsigned)(ceil_div(tiles * g.nsplit, 8) * 8));
  {
  KernelTimerScope timer("gemm_tn_glds", s);
  if (nb == 8) hipLaunchKernelGGL((gemm_tn_glds_kernel<8, 1>), grid1, dim3(256), 0, s, g);
#ifdef HN_GEMM_NT_BENCH
  else if (variant == 1) hipLaunchKernelGGL((gemm_tn_glds_kernel<7, 0>), dim3(tiles, g.nsplit), dim3(256), 0, s, g);
  else if (variant == 2) hipLaunchKernelGGL((gemm_tn_glds_kernel<7, 1, 1>), grid1, dim3(256), 0, s, g);
  else if (variant == 3) hipLaunchKernelGGL((gemm_tn_glds_kernel<7, 1, 2>), grid1, dim3(256), 0, s, g);
  else if (variant == 4) hipLaunchKernelGGL((gemm_tn_glds_kernel<7, 1, 4>), grid1, dim3(256), 0, s, g);
  else if (variant == 5) hipLaunchKernelGGL((gemm_tn_glds_kernel<7, 1, 7>), grid1, dim3(256), 0, s, g);
#endif
  else hipLaunchKernelGGL((gemm_tn_glds_kernel<7, 1>), grid1, dim3(256), 0, s, g);
  (void)variant;
  }
  HN_LAUNCH_CHECK("gemm_tn_glds");
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)ceil_div(ceil_div(N, 4), 256), (unsigned)(M + (colsum ? ceil_div(M, 256) : 0))), dim3(256), 0, s,
                     scratch, g.nsplit, M, N, g.ldp, C, ldc, alpha, accumulate, g.cs_part, colsum, colsum_accumulate);
  HN_LAUNCH_CHECK("gemm_tn_reduce");
  return HN_OK;
}

// N = output columns of the SOURCE weight; with col_group > 0 the staged / written width is gemm_nt_padded_cols(N, ...)
bool gemm_nt_eligible(long M, int N, int K, long lda, const float *A, int col_group, int col_group_pitch, long ldc, const float *C) {
  const int np = gemm_nt_padded_cols(N, col_group, col_group_pitch);
  return M >= 2048 && np >= 32 && np % 4 == 0 && K >= 64 && lda % 4 == 0 && lda >= K && ((uintptr_t)A & 15) == 0 && ldc % 4 == 0 &&
         ((uintptr_t)C & 15) == 0 && (col_group == 0 || (N % col_group == 0 && col_group_pitch >= col_group));
}
int gemm_nt_padded_cols(int N, int col_group, int col_group_pitch) { return col_group > 0 ? N / col_group * col_group_pitch : N; }
int gemm_nt_ldws(int K) { return (K + 15) / 16 * 16; }
size_t gemm_nt_stage_floats(int N, int K) { return (size_t)N * gemm_nt_ldws(K) + (size_t)(N + 63) / 64 * 64 + 64; }

// variant 0 (the product's): 128 x 128 tile, 4 waves of 64 x 64, 2 ring slots = 64 KB of LDS, 2 workgroups per CU.  Measured at cfg4's
// shape (tools/ubench/gemm_f32_bench.hip, profiles/r04_a_gemm_nt_ab.log): 382 us = 0.863 of the fp32 MFMA peak against 459 us
// (0.718) for gemm_big_kernel; 3 slots at one workgroup per CU 448 us, 256 x 128 on 8 waves 407 us, 256 x 256 (128 x 64 per wave)
// 392 us; without its stores 369 us, without loads / barriers 378-383 us, MFMAs + fragment reads alone 363 us (0.909: the clock).
int launch_gemm_nt(const GemmNtArgs &g_in, int variant, hipStream_t s) {
  GemmNtArgs g = g_in;
  HN_REQUIRE(g.A && g.W && g.C, HN_E_NULL, "gemm_nt: NULL operand");
  HN_REQUIRE(g.ldw % 4 == 0 && g.ldw >= gemm_nt_ldws(g.K) && ((uintptr_t)g.W & 15) == 0, HN_E_SHAPE, "gemm_nt: staged weight pitch %ld", g.ldw);
  if (variant == 0 && g.N <= 128) variant = g.N <= 32 ? 20 : (g.N <= 64 ? 21 : 22);      // narrow outputs: the streaming tilings
  int bm = (variant >= 2 && variant <= 4) ? 256 : 128, bn = variant == 4 ? 256 : 128;
  if (variant >= 20 && variant <= 22) { bm = 64; bn = 32 << (variant - 20); }
  g.ntm = ceil_div(g.M, bm); g.ntn = ceil_div(g.N, bn);
  const long blocks = (long)ceil_div(g.ntm, 8) * 8 * g.ntn;
  HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "gemm_nt: grid too large");
  KernelTimerScope timer("gemm_nt_glds", s);
  switch (variant) {
    case 0: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 2, 2, 2>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    // narrow outputs (one 16 .. 64-wide head: 2 * dhp = 32 / 64 / 128 columns): an HBM stream of the context, not a matrix problem --
    // 64-row tiles (two to three workgroups per CU) and four ring slots keep ~100 KB per CU in flight
    case 20: hipLaunchKernelGGL((gemm_nt_glds_kernel<2, 1, 2, 2, 4>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    case 21: hipLaunchKernelGGL((gemm_nt_glds_kernel<2, 2, 2, 2, 4>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    case 22: hipLaunchKernelGGL((gemm_nt_glds_kernel<2, 4, 2, 2, 3>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
#ifdef HN_GEMM_NT_BENCH
    case 1: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 2, 2, 3>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 4, 2, 2>), dim3((unsigned)blocks), dim3(512), 0, s, g); break;
    case 3: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 4, 2, 3>), dim3((unsigned)blocks), dim3(512), 0, s, g); break;
    case 4: hipLaunchKernelGGL((gemm_nt_glds_kernel<8, 4, 2, 4, 2>), dim3((unsigned)blocks), dim3(512), 0, s, g); break;
    case 10: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 2, 2, 2, 1>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    case 11: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 2, 2, 2, 2>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    case 12: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 2, 2, 2, 4>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    case 13: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 2, 2, 2, 7>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
#endif
    default: return fail(HN_E_UNSUPPORTED, "gemm_nt: variant %d", variant);
  }
  HN_LAUNCH_CHECK("gemm_nt_glds");
  return HN_OK;
}

int launch_gemm_nt_stage(const float *W, long ldw, const float *gamma, const float *beta, const float *bias, int N, int K, float *Ws,
                         float *bs, hipStream_t s, int col_group, int col_group_pitch) {
  const int np = gemm_nt_padded_cols(N, col_group, col_group_pitch);
  hipLaunchKernelGGL(gemm_nt_stage_kernel, dim3(ceil_div(np, 4)), dim3(256), 0, s, W, ldw, gamma, beta, bias, np, K, Ws, gemm_nt_ldws(K), bs,
                     col_group, col_group_pitch);
  HN_LAUNCH_CHECK("gemm_nt_stage");
  return HN_OK;
}

}  // namespace hn
