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

// ------------------------------------------------------------------------------------------------
// The same product for a NARROW A (M <= 64 output rows per tile: G = dKV^T z of a one-head model, dim_head 16 / 27 -> M = 32 / 64;
// the reference's tuned TCGA shapes).  On the 128-row tile above half or three quarters of the MFMAs would multiply columns nobody
// owns; here a tile is MT (32 | 64) x NB * 16 and the four waves split the M tile (MW = MT / 32 ways) AND the k-tile's eight 4-row
// steps (KW = 4 / MW ways): every wave still owns a 32 x 112 accumulator, fed by 1 / KW of the steps.  The KW partial accumulators
// of a tile are folded through LDS in a fixed order after the last k-tile (the ring is free by then), so a workgroup writes ONE
// partial tile.  A k-tile is 32 rows x (MT + 128) floats = 20 / 24 KB by LDS-DMA, 5 / 6 instructions per wave; two ring slots fit
// three workgroups per CU.  Bound by the B stream (z: 101 MB at the tuned shapes), not by the matrix pipe.
// ------------------------------------------------------------------------------------------------
template <int MT, int NB>
__global__ __launch_bounds__(256) void gemm_tn_narrow_kernel(GemmTnGArgs g) {
  constexpr int S = 2, MW = MT / 32, KW = 4 / MW, QS = 4 / KW;      // QS steps of each half of a k-tile per wave
  constexpr int STAGE = 32 * MT + 32 * 128;
  constexpr int NA = MT / 8, RA = 256 / MT, LW = (NA + 16) / 4;     // A: NA instructions of RA rows; B: 16 instructions of 2 rows
  constexpr int BNT = NB * 16;
  constexpr int RED = (KW - 1) * MW * 2 * NB * 4 * 64;             // floats of the cross-k-group fold
  constexpr int LDSF = S * STAGE > RED + 1024 ? S * STAGE : RED + 1024;      // (+ 256 x 4 floats: the column-sum fold)
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % MW, kw = wave / MW;
  const int tiles = g.ntm * g.ntn, total = tiles * g.nsplit, per = (total + 7) >> 3;
  const int lin = (blockIdx.x & 7) * per + (blockIdx.x >> 3);     // an XCD owns a run of consecutive (slice, tile) items
  if (lin >= total) return;
  const int z = lin / tiles, tile = lin - z * tiles;
  const int m_tile = tile % g.ntm, n_tile = tile / g.ntm;
  const int m0 = m_tile * MT, n0 = n_tile * BNT;
  const int k_begin = z * g.kslice, rows = min(g.K, k_begin + g.kslice) - k_begin;
  const int lda = (int)g.lda, ldb = (int)g.ldb;
  const long a_bytes = ((long)(rows - 1) * lda + (g.M - m0)) * 4, b_bytes = ((long)(rows - 1) * ldb + (g.N - n0)) * 4;
  const unsigned a_clamp = (unsigned)(a_bytes > 0 ? a_bytes : 0), b_clamp = (unsigned)(b_bytes > 0 ? b_bytes : 0);
  const i32x4 rsA = make_rsrc(g.A + (long)k_begin * lda + m0, a_clamp);
  const i32x4 rsB = make_rsrc(g.B + (long)k_begin * ldb + n0, b_clamp);
  const int voffA = (lane / (MT / 4)) * lda * 4 + ((lane % (MT / 4)) << 4), voffB = (lane >> 5) * ldb * 4 + ((lane & 31) << 4);
  const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_byte_address(lds));
  auto issue = [&](int kt) {
#pragma unroll
    for (int q = 0; q < LW; ++q) {
      const int u = wave + 4 * q;                                 // u < NA: A rows RA u ..; else B rows 2 (u - NA) ..  (NA % 4 == 0: q decides)
      const unsigned dst = lds_base + (unsigned)(((kt % S) * STAGE + u * 256) * 4);
      if (4 * q < NA) hn_glds16(rsA, dst, voffA, (int)min((unsigned)((kt * 32 + RA * u) * lda * 4), a_clamp));
      else hn_glds16(rsB, dst, voffB, (int)min((unsigned)((kt * 32 + 2 * (u - NA)) * ldb * 4), b_clamp));
    }
  };
  const int fi = lane & 15, fg = lane >> 4;
  const int a_off = fg * MT + 32 * wm + 2 * fi;                   // + 4 MT per 4-row step
  const int b_off = 32 * MT + fg * 128;                           // + 512 per step
  struct Frags { float2 a[QS]; f32x4 b[QS]; float c[QS][NB - 4]; };
  auto read_frags = [&](int kt, int half, Frags &f) {
    const float *st = lds + (kt % S) * STAGE;
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const int step = half * 4 + kw * QS + q;
      f.a[q] = *(const float2 *)&st[a_off + step * 4 * MT];
      f.b[q] = *(const f32x4 *)&st[b_off + step * 512 + 4 * fi];
      if (NB == 8) {
        const f32x4 t = *(const f32x4 *)&st[b_off + step * 512 + 64 + 4 * fi];
#pragma unroll
        for (int e = 0; e < NB - 4; ++e) f.c[q][e] = t[e];
      } else {
#pragma unroll
        for (int e = 0; e < NB - 4; ++e) f.c[q][e] = st[b_off + step * 512 + 64 + 16 * e + fi];
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
    for (int q = 0; q < QS; ++q) {
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
  // column sums of A: 256 threads = (1024 / MT) row groups x MT / 4 column quads, 32 MT / 1024 rows each
  constexpr int RG = 1024 / MT, CQ = 32 / RG;
  const bool do_cs = g.cs_part != nullptr && n_tile == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  const int cs_off = (tid / (MT / 4)) * MT + ((tid % (MT / 4)) << 2);

  const int nk = (rows + 31) >> 5;
  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  Frags f0, f1;
  if (nk > 0) {
    if (nk > 1) wait_vmcnt<LW>(); else wait_vmcnt<0>();
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
      for (int q = 0; q < CQ; ++q) csum += *(const f32x4 *)&st[cs_off + q * RG * MT];
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(f0);
    __builtin_amdgcn_sched_barrier(0);
    if (!last) {
      wait_vmcnt<0>();                                            // two slots: the tile needed next is the only one in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + S < nk) issue(kt + S);
      read_frags(kt + 1, 0, f0);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_half(f1);
  }
  // fold the KW k-groups of a tile: groups 1.. park their accumulators in LDS (lane-major, 16-byte pieces), group 0 adds them in
  // order.  (All fragment reads of the ring are behind the barrier.)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 *red = (f32x4 *)lds;
  if (kw > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) red[(((kw - 1) * MW + wm) * 2 * NB + i * NB + j) * 64 + lane] = acc[i][j];
  }
  if (do_cs) red[RED / 4 + tid] = csum;
  __syncthreads();
  if (do_cs && tid < MT / 4) {                                    // the RG row groups of a column quad, in order: ONE column-sum row per slice
    f32x4 t = red[RED / 4 + tid];
#pragma unroll
    for (int rg = 1; rg < RG; ++rg) t += red[RED / 4 + rg * (MT / 4) + tid];
    const int m = m0 + (tid << 2);
    if (m < g.M) *(f32x4 *)&g.cs_part[(long)z * g.M + m] = t;
  }
  if (kw > 0) return;
#pragma unroll
  for (int k2 = 1; k2 < KW; ++k2)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] += red[(((k2 - 1) * MW + wm) * 2 * NB + i * NB + j) * 64 + lane];
  float *P = g.part + (long)z * g.M * g.ldp;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 32 * wm + 2 * fi + i;
    if (m >= g.M) continue;
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
// blockIdx.y = output row (rows >= M: the column sums), a thread takes four consecutive columns of it: one 16-byte load per slice (the
// partial rows are 16-byte aligned), four slices in flight per trip, no index division.  (One element per thread with a 64-bit
// divide and 16 clamped loads ran at 1.2 TB/s: 24 us for 29 MB.)
// SG = slice groups per workgroup: a narrow output (M = 32: 32 rows x 194 column quads) has too few (row, quad) pairs to fill the
// chip and ~100 slices to sum per pair -- one chain of dependent loads per thread took 50+ us.  With SG > 1 a workgroup is
// SG x (256 / SG) threads, group sg sums the slices k = sg (mod SG) and the groups are folded through LDS in index order.
template <int SG>
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float *__restrict__ part, int nsplit, int M, int N, long ldp,
                                                             float *__restrict__ C, long ldc, float alpha, int accumulate,
                                                             const float *__restrict__ cs_part, float *__restrict__ cs_out, int cs_accumulate,
                                                             int cs_groups) {
  constexpr int QB = 256 / SG;                              // column quads per workgroup
  __shared__ f32x4 fold[SG > 1 ? 256 : 1];
  const int m = blockIdx.y;
  if (m >= M) {                                             // column sums: nsplit * cs_groups row-group partials per column, 256 columns
    const int c = (m - M) * 256 + threadIdx.x;              // per row of blocks, eight independent chains per thread (fixed order)
    if (blockIdx.x != 0 || c >= M) return;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int np = nsplit * cs_groups;
    int k = 0;
    for (; k + 8 <= np; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += cs_part[(long)(k + u) * M + c];
    }
    for (; k < np; ++k) a8[k & 7] += cs_part[(long)k * M + c];
    const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    cs_out[c] = cs_accumulate ? cs_out[c] + acc : acc;
    return;
  }
  const int sg = threadIdx.x / QB, n = (blockIdx.x * QB + threadIdx.x % QB) * 4;
  const long slab = (long)M * ldp;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    const float *src = part + (long)m * ldp + n;
    int k = sg;
    for (; k + 3 * SG < nsplit; k += 4 * SG) {
      const f32x4 v0 = *(const f32x4 *)&src[(long)k * slab], v1 = *(const f32x4 *)&src[(long)(k + SG) * slab];
      const f32x4 v2 = *(const f32x4 *)&src[(long)(k + 2 * SG) * slab], v3 = *(const f32x4 *)&src[(long)(k + 3 * SG) * slab];
      acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; k < nsplit; k += SG) acc += *(const f32x4 *)&src[(long)k * slab];
  }
  if (SG > 1) {
    fold[threadIdx.x] = acc;
    __syncthreads();
    if (sg != 0) return;
#pragma unroll
    for (int u = 1; u < SG; ++u) acc += fold[u * QB + threadIdx.x];
  }
  if (n >= N) return;
  float *dst = C + (long)m * ldc + n;
  float old[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) old[e] = (accumulate && n + e < N) ? dst[e] : 0.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n + e < N) dst[e] = old[e] + alpha * acc[e];
}

#ifdef HN_GEMM_NT_BENCH
static bool g_tn_bench_skip_reduce = false;
#endif
bool gemm_tn_glds_eligible(const float *A, long lda, const float *B, long ldb, int M, int N, int K) {
  return K >= 4096 && M >= 32 && M < 65000 && N >= 112 && M % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
         (long)K * lda * 4 < (1L << 31) && (long)K * ldb * 4 < (1L << 31);
}

// row tile of the TN product: 128 (gemm_tn_glds_kernel) or the narrow kernel's 64 / 32 where those pad M less
static int gemm_tn_row_tile(int M) {
  // measured over 32 768 x 773 (tools/ubench/gemm_f32_bench, incl. the reduce): M = 32: 32 us on 32-row tiles (43 on 64, 82 on
  // 128); M = 64: 42 / 46 / 82; M = 128: 63 / 64 / 89; M = 256: 114 on 64-row tiles against 139 on 128
  if (M <= 64) return 32;
  if (M <= 256) return 64;
  return 128;
}

// floats of scratch one row slice needs (partials + column-sum partials); the routes size their scratch for >= 32 slices of it
size_t gemm_tn_glds_slice_floats(int M, int N, bool colsum) {
  const int nb = ceil_div(N, 112) * 112 <= ceil_div(N, 128) * 128 ? 7 : 8;
  const int mt = gemm_tn_row_tile(M);
  return (size_t)M * ceil_div(N, nb * 16) * nb * 16 + (colsum ? (size_t)(mt == 128 ? 8 : 1) * M : 0);
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
  int mt = gemm_tn_row_tile(M);
#ifdef HN_GEMM_NT_BENCH
  if (variant >= 32) { mt = variant & ~3; variant = 0; }     // bench: force the row tile (32 | 64 | 128)
#endif
  const int cs_groups = mt == 128 ? 8 : 1;                   // rows of column-sum partials per slice (the narrow kernel folds its row groups)
  g.ntm = ceil_div(M, mt); g.ntn = ceil_div(N, nb * 16);
  g.ldp = (long)g.ntn * nb * 16;
  const int tiles = g.ntm * g.ntn;
  int nsplit = (mt == 128 ? 512 : 768) / tiles;              // 64 KB (48 / 42 KB narrow) of LDS: 2 (3) workgroups per CU, one resident round
  if (nsplit < 1) nsplit = 1;
  const size_t per = (size_t)M * g.ldp + (colsum ? (size_t)cs_groups * M : 0);
  if ((size_t)nsplit * per > scratch_floats) nsplit = (int)(scratch_floats / per);
  HN_REQUIRE(nsplit >= 1, HN_E_WORKSPACE, "gemm_tn_glds: scratch %zu floats < %zu", scratch_floats, per);
  g.kslice = ceil_div(ceil_div(K, nsplit), 32) * 32;
  g.nsplit = ceil_div(K, g.kslice);
  g.part = scratch;
  g.cs_part = colsum ? scratch + (size_t)g.nsplit * M * g.ldp : nullptr;
  const dim3 grid1((unsigned)(ceil_div(tiles * g.nsplit, 8) * 8));
  {
  KernelTimerScope timer("gemm_tn_glds", s);
  if (mt == 64) {
    if (nb == 8) hipLaunchKernelGGL((gemm_tn_narrow_kernel<64, 8>), grid1, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_tn_narrow_kernel<64, 7>), grid1, dim3(256), 0, s, g);
  } else if (mt == 32) {
    if (nb == 8) hipLaunchKernelGGL((gemm_tn_narrow_kernel<32, 8>), grid1, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_tn_narrow_kernel<32, 7>), grid1, dim3(256), 0, s, g);
  } else if (nb == 8) hipLaunchKernelGGL((gemm_tn_glds_kernel<8, 1>), grid1, dim3(256), 0, s, g);
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
#ifdef HN_GEMM_NT_BENCH
  if (g_tn_bench_skip_reduce) return HN_OK;
#endif
  const unsigned red_rows = (unsigned)(M + (colsum ? ceil_div(M, 256) : 0));
  if (M >= 512)
    hipLaunchKernelGGL(gemm_tn_reduce_kernel<1>, dim3((unsigned)ceil_div(ceil_div(N, 4), 256), red_rows), dim3(256), 0, s, scratch, g.nsplit, M, N, g.ldp, C,
                       ldc, alpha, accumulate, g.cs_part, colsum, colsum_accumulate, cs_groups);
  else
    hipLaunchKernelGGL(gemm_tn_reduce_kernel<16>, dim3((unsigned)ceil_div(ceil_div(N, 4), 16), red_rows), dim3(256), 0, s, scratch, g.nsplit, M, N, g.ldp, C,
                       ldc, alpha, accumulate, g.cs_part, colsum, colsum_accumulate, cs_groups);
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
