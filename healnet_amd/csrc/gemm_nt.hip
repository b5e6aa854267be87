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

namespace hn {

__device__ void hn_glds16(i32x4 rsrc, __attribute__((address_space(3))) void *lds, int size, int voffset, int soffset, int offset,
                          int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Ws[n, k] = W[n, k] * gamma[k] for k < K, 0 for K <= k < ldws;  bs[n] = sum_k W[n, k] * beta[k] (+ bias[n]).  One wave per row.
__global__ __launch_bounds__(256) void gemm_nt_stage_kernel(const float *__restrict__ W, long ldw, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, const float *__restrict__ bias, int N, int K,
                                                            float *__restrict__ Ws, int ldws, float *__restrict__ bs) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float *w = W + (long)n * ldw;
  float *o = Ws + (long)n * ldws;
  float acc = 0.0f;
  for (int k = lane; k < ldws; k += 64) {
    float v = 0.0f;
    if (k < K) {
      const float x = w[k];
      v = gamma ? x * gamma[k] : x;
      if (beta) acc += x * beta[k];
    }
    o[k] = v;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
  if (lane == 0) bs[n] = acc + (bias ? bias[n] : 0.0f);
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
  const i32x4 rsA = make_rsrc(g.A + (long)m0 * lda, (unsigned)(rows_a * lda * 4));
  const i32x4 rsW = make_rsrc(g.W + (long)n0 * ldw, (unsigned)(rows_w * ldw * 4));

  // loader: lane (r8 = lane >> 3, p = lane & 7) fetches the 16-byte slot p ^ r8 of row 8 u + r8 and lands at slot p
  const int r8 = lane >> 3, p = lane & 7;
  const int voffA = r8 * lda * 4 + ((p ^ r8) << 4), voffW = r8 * ldw * 4 + ((p ^ r8) << 4);
  auto issue = [&](int kt) {
    float *st = lds + (kt % S) * STAGE;
#pragma unroll
    for (int q = 0; q < LW; ++q) {
      const int u = wave + NW * q;
      auto dst = (__attribute__((address_space(3))) void *)(st + u * 256);
      if (q < LWA) hn_glds16(rsA, dst, 16, voffA, 8 * u * lda * 4 + kt * 128, 0, 0);
      else hn_glds16(rsW, dst, 16, voffW, 8 * (u - LA) * ldw * 4 + kt * 128, 0, 0);
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
  auto wait_tile = [&](int kt) {                                   // own loads of tile kt done: at most the tiles behind it outstanding
    const int ahead = min(S - 1, nk - 1 - kt);
    if (S >= 3 && ahead == 2) wait_vmcnt<2 * LW>();
    else if (S >= 2 && ahead >= 1) wait_vmcnt<LW>();
    else wait_vmcnt<0>();
  };
  wait_tile(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Frags f0, f1;
  read_frags(0, 0, f0);
  for (int kt = 0; kt < nk; ++kt) {
    const bool last = kt + 1 == nk, two = !last || tail_steps == 2;
    if (two) read_frags(kt, 1, f1);
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

bool gemm_nt_eligible(long M, int N, int K, long lda, const float *A, int col_group, int col_group_pitch, long ldc, const float *C) {
  return M >= 2048 && N >= 256 && N % 4 == 0 && K >= 64 && lda % 4 == 0 && lda >= K && ((uintptr_t)A & 15) == 0 && ldc % 4 == 0 &&
         ((uintptr_t)C & 15) == 0 && (col_group == 0 || (col_group % 4 == 0 && col_group_pitch % 4 == 0));
}
int gemm_nt_ldws(int K) { return (K + 15) / 16 * 16; }
size_t gemm_nt_stage_floats(int N, int K) { return (size_t)N * gemm_nt_ldws(K) + (size_t)(N + 63) / 64 * 64 + 64; }

// variant: 0 = 128 x 128 / 4 waves / 2 slots (64 KB, 2 workgroups per CU), 1 = 128 x 128 / 4 waves / 3 slots, 2 = 256 x 128 / 8 waves / 2
// slots, 3 = 256 x 128 / 8 waves / 3 slots
int launch_gemm_nt(const GemmNtArgs &g_in, int variant, hipStream_t s) {
  GemmNtArgs g = g_in;
  HN_REQUIRE(g.A && g.W && g.C, HN_E_NULL, "gemm_nt: NULL operand");
  HN_REQUIRE(g.ldw % 4 == 0 && g.ldw >= gemm_nt_ldws(g.K) && ((uintptr_t)g.W & 15) == 0, HN_E_SHAPE, "gemm_nt: staged weight pitch %ld", g.ldw);
  const int bm = (variant >= 2 && variant <= 4) ? 256 : 128, bn = variant == 4 ? 256 : 128;
  g.ntm = ceil_div(g.M, bm); g.ntn = ceil_div(g.N, bn);
  const long blocks = (long)ceil_div(g.ntm, 8) * 8 * g.ntn;
  HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "gemm_nt: grid too large");
  switch (variant) {
    case 0: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 2, 2, 2>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    case 1: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 2, 2, 3>), dim3((unsigned)blocks), dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 4, 2, 2>), dim3((unsigned)blocks), dim3(512), 0, s, g); break;
    case 3: hipLaunchKernelGGL((gemm_nt_glds_kernel<4, 4, 4, 2, 3>), dim3((unsigned)blocks), dim3(512), 0, s, g); break;
#ifdef HN_GEMM_NT_BENCH
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
                         float *bs, hipStream_t s) {
  hipLaunchKernelGGL(gemm_nt_stage_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, s, W, ldw, gamma, beta, bias, N, K, Ws, gemm_nt_ldws(K), bs);
  HN_LAUNCH_CHECK("gemm_nt_stage");
  return HN_OK;
}

}  // namespace hn
