// Split-KV flash attention for a short latent query block against a long context, exact fp32 on the
// CDNA4 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Replaces the materialised-score attention of the reference (healnet/models/healnet.py
// Attention.forward :407-425: einsum QK^T :409, *scale :409, mask :411-415, softmax(x/0.5) :419/:364-365,
// einsum PV :424): the (b*h, L, N) score / probability tensors never exist.
//
// Shape regime.  The query side is the latent array: L = l_c rows (128 by default), while the context
// has N = 1 .. 6e5 tokens.  Tiling over queries (classic flash attention) gives no parallelism, so the
// grid splits the TOKENS: one wave owns NQ 16-row query tiles of one (sample, head) and walks a
// contiguous token range in 16-token steps, keeping running (max, sum, O) per query row; a merge
// kernel combines the splits in fixed order (deterministic).
//
// Register-only dataflow (no LDS, no barriers; waves are fully independent):
//   S^T = K Q^T is computed with A = K tile (M = 16 tokens), B = Q tile (N = 16 query rows).  With the
//   16x16x4 C/D map (col = lane & 15, row = 4*(lane >> 4) + reg) lane (g, j) then holds the scores of
//   query row j for tokens 4g + r, r = 0..3 -- which is exactly the A-operand layout of the P V product
//   (A[i = lane & 15][k = lane >> 4]) if the k-chunk of MFMA r is defined as tokens {4g + r : g = 0..3}.
//   Token order inside a sum is free, so P never leaves its registers: exp2 in place, feed to PV.
//   The contraction index of QK^T is likewise permuted (step (s, c) uses d = 16 s + 4 g + c) so that
//   both operand fragments are contiguous 16-byte loads.
//
// Softmax: logits arrive pre-scaled to log2 units (2 * dim_head^-1/2 * log2(e) folded into Q), running
// max with a lazy rescale (only when some score exceeds the running max by > 2^8), masked / out-of-range
// tokens contribute exactly 0.
//
// Two operand bindings share the kernel:
//   explicit  : K, V = projected keys / values, dp = dim_head padded to 16/32/64/128
//   rank-D    : K = V = the normalised context z (b, N, dp) shared by all heads, queries pre-folded
//               with W_k and gamma (qfold_kernel), values projected after the reduction
//               (merge_vproj_kernel).  Used when D (13 for an RGB image + 2-axis Fourier features)
//               is below dim_head: 4.5x fewer executed FLOPs, identical math up to fp32 rounding.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace hn {

constexpr float kNegBig = -1.0e30f;
constexpr float kRescaleThreshold = 8.0f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ONES (rank-D binding only; requires D <= DP - 1): column DP-1 of the shared context row is a synthetic
// ones column injected in registers (memory keeps its zero padding):
//   * QK^T: the first MFMA of the chain takes -m (the running reference max of its query row, replicated in a
//     persistent register quad that only changes on a rescale) as its C operand, so the chain itself delivers
//     s - m: no per-score subtraction on the VALU;
//   * P V : the V fragment carries 1 there, so accumulator column DP-1 is sum_t p = the softmax
//     denominator: no per-score addition on the VALU, and it is rescaled together with O.
// What is left per score is one v_exp_f32 and 3/4 of a max (overflow guard on p).  This matters more than
// on the bf16 path: measured MFMA-pipe utilisation tracks  MFMA cycles / (MFMA cycles + VALU cycles) of the
// loop (58 % at ~150 VALU ops per 32 MFMAs, 66 % at ~110, 74 % at ~60) and interleaving the two inside a
// wave (software pipelining QK^T(t+1) under softmax(t): tried, 6 % slower) does not help -- the fp32 MFMA
// runs at exactly the fp32 vector rate and evidently competes with VALU work for the SIMD's fp32 lanes, so
// every VALU op removed from the loop is MFMA time gained.
// KS = number of QK^T k-steps (4 per 16-column block); < 4*DT only with the packed context layout (common.h).
__device__ __forceinline__ float f4c(const float4 &v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// (second launch bound = waves per SIMD the register allocation must leave room for: the dp = 16 image core is planned for FOUR --
// attn_core_geometry's 4096 resident waves; a fifth of the 512 registers each -- whatever the two instances of its token loop do)
template <int DT, int NQ, bool ONES, int KS = 4 * DT, bool DROP = false>
__global__ __launch_bounds__(256, (DT == 1 && NQ == 4 && !DROP) ? 4 : 1) void attn_core_kernel(AttnCoreArgs a, int ngroups, int gy, int waves_per_block) {
  constexpr int DP = 16 * DT;
  const int L = a.Lq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;

  // XCD-aware block remap: consecutive work items of one sample stay on one XCD (blocks are dealt
  // round-robin to the 8 XCDs), so the sample's context is fetched into one L2 only.  Speed only.
  long total = (long)gridDim.x;
  long id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int split = (int)(id % a.nsplit);
  const int yb = (int)((id / a.nsplit) % gy);
  const int bh = (int)(id / ((long)a.nsplit * gy));
  const int qg = yb * waves_per_block + wave;
  if (qg >= ngroups) return;
  const int bi = bh / a.h, hi = bh % a.h;

  // ---- query fragments (B operand): lane (g, j) holds Q[row = tile*16 + j][16 s + 4 g + c]
  float4 qf[NQ][DT];
  const float *qbase = a.Q + (long)bi * a.q_b + (long)hi * a.q_h;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int row = (qg * NQ + i) * 16 + j;
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      qf[i][s] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < L) qf[i][s] = *(const float4 *)(qbase + (long)row * a.ldq + 16 * s + 4 * g);
    }
  }

  f32x4 O[NQ][DT];
  f32x4 negm[NQ];            // ONES: -m replicated, the C operand of the first QK^T MFMA (delivers s - m for free)
  float m[NQ], l[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    negm[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m[i] = ONES ? 0.0f : kNegBig;
    l[i] = 0.0f;
#pragma unroll
    for (int d = 0; d < DT; ++d) O[i][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  bool unset = true;      // ONES: no unmasked token seen yet (wave-uniform)
  // ONES with a LayerNorm-ed context: the reference of every row is its Cauchy-Schwarz score bound (qfold_kernel), fixed for
  // the whole launch: no running max, no overflow guard, no rescale (wave-uniform switch; the flag is set on the device).
  // kept as a scalar INTEGER: a bool carried into the loop comes back as v_cndmask / v_cmp pairs (i1 copies through a VGPR) in
  // front of every branch on it -- VALU instructions in a loop where every VALU cycle is MFMA time lost
  const int unbounded = (ONES && a.bound != nullptr) ? __builtin_amdgcn_readfirstlane(*a.bound_flag) : 1;
  const bool bounded = unbounded == 0;
  if (bounded) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int row = min((qg * NQ + i) * 16 + j, a.Lp - 1);
      m[i] = a.bound[(long)bh * a.Lp + row];
      negm[i] = (f32x4){-m[i], -m[i], -m[i], -m[i]};
    }
    unset = false;
  }

  const int t_begin = split * a.chunk;
  const int t_end = min(a.N, t_begin + a.chunk);
  const float *kbase = a.Kp + (long)bi * a.k_b + (long)hi * a.k_h;
  const float *vbase = a.Vp + (long)bi * a.v_b + (long)hi * a.v_h;
  const uint8_t *mrow = a.mask ? a.mask + (long)bi * a.N : nullptr;

  // K / V fragments come through buffer descriptors (SRSRC): the per-lane offsets are loop invariant and the
  // tile base advances in an SGPR, so the steady-state loop spends no VALU on addressing, and rows past the
  // end of the context read as 0 (hardware range check) instead of needing clamped indices.
  const unsigned kbytes = (unsigned)(((long)(a.N - 1) * a.ldk + DP) * 4), vbytes = (unsigned)(((long)(a.N - 1) * a.ldv + DP) * 4);
  const i32x4 krs = make_rsrc(kbase, kbytes), vrs = make_rsrc(vbase, vbytes);
  const int koff = (j * a.ldk + 4 * g) * 4;
  int voff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) voff[r] = ((4 * g + r) * a.ldv + j) * 4;

  auto load_kv = [&](int t0, float4 (&kf)[DT], float (&vf)[DT][4]) {
    const int ks = t0 * a.ldk * 4, vs = t0 * a.ldv * 4;
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      const f32x4 w = hn_buffer_load_x4(krs, koff + 64 * s, ks, 0);
      kf[s] = make_float4(w.x, w.y, w.z, w.w);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
        vf[d][r] = hn_buffer_load_x1(vrs, voff[r] + 64 * d, vs, 0);
    }
  };

  // one 16-token step on the (kf, vf) fragments; prefetches the following tile into (kn, vn)
  // BND (a std::integral_constant<int, ...> tag): the fixed-reference mode as a COMPILE-TIME property of the loop (image core).  As a run-time
  // scalar the wave-uniform switch still cost two vector instructions per step (the `rescale` flag's i1 copy through a VGPR:
  // v_cndmask + v_cmp in front of the branch) plus the branch itself, in a loop where every vector cycle is matrix time lost; the
  // token loop below is therefore instantiated twice and the switch taken once, in front of it.
  auto step = [&](auto BND, int t0, float4 (&kf)[DT], float (&vf)[DT][4], float4 (&kn)[DT], float (&vn)[DT][4]) __attribute__((always_inline)) {
    constexpr int kMode = decltype(BND)::value;      // 0: running reference, 1: fixed reference, 2: decided per launch (run-time switch)
    // Unconditional: past the split the rows belong to the next split (read and never used), past the context they read 0
    // through the descriptor.  Behind a branch the request count differs between the two paths into the join, and the
    // compiler then waits vmcnt(0) -- for the prefetch it has just issued -- in front of the tile's first MFMA.
    load_kv(t0 + 16, kn, vn);
    if (DROP && a.drop_rowsum && !a.ones_in_mem) {   // shared-context binding under dropout: accumulator column DP-1 = sum_t p'_t, the row sum
      if (j == 15) {                    // of the THINNED probabilities (no longer 1), needed by the beta term of the values
#pragma unroll
        for (int r = 0; r < 4; ++r) vf[DT - 1][r] = 1.0f;
      }
    }
    if (ONES && !a.ones_in_mem) {      // hn_fusion_forward has K1 write the ones column into z itself
      asm volatile("" ::: "memory");   // keeps this a scalar BRANCH: if-converted it is five v_cndmask per step for every caller
      if (g == 3) kf[DT - 1].w = 1.0f;
      if (j == 15) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vf[DT - 1][r] = 1.0f;
      }
    }

    // ---- S^T tile = K Q^T  (KS chained MFMAs per query tile, NQ independent chains; C operand = -m with ONES)
    f32x4 S[NQ];
#pragma unroll
    for (int st = 0; st < KS; ++st) {
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        S[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(kf[st >> 2], st & 3), f4c(qf[i][st >> 2], st & 3),
                                                    st == 0 ? (ONES ? negm[i] : (f32x4){0.f, 0.f, 0.f, 0.f}) : S[i], 0, 0, 0);
    }

    // ---- mask / ragged tail: lane (g, j) holds tokens t0 + 4 g + r
    bool any_live = true;
    if (mrow != nullptr || t0 + 16 > t_end) {
      bool live = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tok = t0 + 4 * g + r;
        bool ok = tok < t_end;
        if (ok && mrow) ok = mrow[tok] != 0;
        live |= ok;
        if (!ok) {
#pragma unroll
          for (int i = 0; i < NQ; ++i) S[i][r] = -__builtin_inff();
        }
      }
      any_live = __any(live);
    }

    f32x4 P[NQ];
    if (ONES) {
      // S already holds s - m.  Guard: rescale when some p would exceed 2^threshold (or on the first live tile).
      // The guard tests the SUM of the step's p values (p >= 0, so sum >= max): packed adds cost about half of a
      // max3 chain, and with an up-to-date reference the sum of 16 values stays <= 16.
      float ps0 = 0.0f, ps1 = 0.0f;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) P[i][r] = fast_exp2(S[i][r]);
      }
      // (one scalar branch on `bounded`: folded into the per-lane `need` it came back as v_cndmask / v_cmp pairs in front of two
      // branches -- four VALU instructions per step in a loop where every VALU cycle is MFMA time lost)
      // (all sixteen exponentials in front of the first P V product: with nothing between them the scheduler pairs each with its
      // MFMA -- exp, wait states, MFMA, sixteen times over)
      if constexpr (kMode == 1) __builtin_amdgcn_sched_barrier(0);
      bool rescale = false;
      bool guard = kMode == 0;
      if constexpr (kMode == 2) guard = unbounded != 0;
      if (guard) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          ps0 += P[i][0] + P[i][2];
          ps1 += P[i][1] + P[i][3];
        }
        const bool need = !(ps0 + ps1 <= 256.0f) || unset;
        rescale = __any(need) && any_live;
      }
      if (rescale) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          float tm = fmaxf(fmaxf(S[i][0], S[i][1]), fmaxf(S[i][2], S[i][3]));
          tm = fmaxf(tm, __shfl_xor(tm, 16));
          tm = fmaxf(tm, __shfl_xor(tm, 32));
          float delta = unset ? tm : fmaxf(tm, 0.0f);
          if (!(delta > -3.0e38f)) delta = 0.0f;          // row saw only -inf scores: keep the reference
          const float alpha = unset ? 1.0f : fast_exp2(-delta);
          m[i] += delta;
          if (DROP) l[i] *= alpha;           // (with dropout the denominator lives on the VALU, see below)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha, 4 * g + r);
#pragma unroll
            for (int d = 0; d < DT; ++d) O[i][d][r] *= ar;
            P[i][r] = fast_exp2(S[i][r] - delta);
          }
          negm[i] = (f32x4){-m[i], -m[i], -m[i], -m[i]};
        }
        unset = false;
      }
      if (DROP) {      // the ones column of V will sum the THINNED probabilities (the row-sum channel): the full sum is added here
#pragma unroll
        for (int i = 0; i < NQ; ++i) l[i] += (P[i][0] + P[i][1]) + (P[i][2] + P[i][3]);
      }
    } else {
      bool need = false;
      float tmax[NQ];
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        tmax[i] = fmaxf(fmaxf(S[i][0], S[i][1]), fmaxf(S[i][2], S[i][3]));
        need |= tmax[i] > m[i] + kRescaleThreshold;
      }
      if (__any(need)) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          float tm = tmax[i];
          tm = fmaxf(tm, __shfl_xor(tm, 16));
          tm = fmaxf(tm, __shfl_xor(tm, 32));
          const float mn = fmaxf(m[i], tm);
          const float alpha = fast_exp2(m[i] - mn);
          l[i] *= alpha;
          // accumulator reg r of lane (g, d) belongs to query row 4 g + r, whose alpha lives in lane 4 g + r
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha, 4 * g + r);
#pragma unroll
            for (int d = 0; d < DT; ++d) O[i][d][r] *= ar;
          }
          m[i] = mn;
        }
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp2(S[i][r] - m[i]);
          l[i] += p;
          P[i][r] = p;
        }
      }
    }
    if (DROP) {   // nn.Dropout on the NORMALISED probabilities (:421): l keeps the full sum, the P V operand is thinned
      // (kept or zero here; the factor 1 / (1 - p) multiplies the accumulators once, in the epilogue)
      // 16-bit decisions, rows 16 apart share a generator call (common.h): four tiles per call when the wave's tiles come in
      // fours and the (b, h) block starts on a multiple of 64 rows (wave-uniform), two per call for pairs on a multiple of 32
      const uint32_t row0 = (uint32_t)(bh * L + qg * NQ * 16 + j), quad = (uint32_t)(t0 + 4 * g) >> 2;
      if (NQ % 4 == 0 && ((bh * L) & 63) == 0) {
#pragma unroll
        for (int i = 0; i + 3 < NQ; i += 4) {
          bool keep[4][4];
          drop_rows4(a.drop, quad, row0 + 16 * i, keep);
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[i + k][r] = keep[k][r] ? P[i + k][r] : 0.0f;
        }
      } else if (NQ % 2 == 0 && ((bh * L) & 31) == 0) {
#pragma unroll
        for (int i = 0; i + 1 < NQ; i += 2) {
          bool lo[4], hi[4];
          drop_pair(a.drop, quad, row0 + 16 * i, lo, hi);
#pragma unroll
          for (int r = 0; r < 4; ++r) { P[i][r] = lo[r] ? P[i][r] : 0.0f; P[i + 1][r] = hi[r] ? P[i + 1][r] : 0.0f; }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          bool keep[4];
          drop_quad_attn(a.drop, quad, row0 + 16 * i, keep);
#pragma unroll
          for (int r = 0; r < 4; ++r) P[i][r] = keep[r] ? P[i][r] : 0.0f;
        }
      }
    }

    // ---- O += P V   (A = P straight from registers, B = V rows 4 g + r)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int d = 0; d < DT; ++d) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
          O[i][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(P[i][r], vf[d][r], O[i][d], 0, 0, 0);
      }
    }
  };

  float4 kA[DT], kB[DT];
  float vA[DT][4], vB[DT][4];
  if (t_begin < t_end) load_kv(t_begin, kA, vA);
  // (both lambdas are force-inlined: left to the inliner's budget, the dropout variants -- whose step carries the generator -- kept `step`
  // as a function of its own, and the fragment / accumulator arrays it takes by reference moved to scratch: 832 bytes per lane, 4x the time)
  auto walk = [&](auto BND) __attribute__((always_inline)) {
    for (int t0 = t_begin; t0 < t_end; t0 += 32) {
      step(BND, t0, kA, vA, kB, vB);
      if (t0 + 16 < t_end) step(BND, t0 + 16, kB, vB, kA, vA);
    }
  };
  // Two instances only where it was measured to pay and the registers are there: the dp = 16 image core with four tiles per wave.
  // Everywhere else ONE loop with the run-time switch, as before -- the second instance costs every variant ~20 VGPRs (the
  // allocation is the maximum over both), which takes the dp = 32 cores with 7 / 8 k-steps from four waves per SIMD to three.
  if constexpr (ONES && DT == 1 && NQ == 4 && !DROP) {
    if (bounded) walk(std::integral_constant<int, 1>{}); else walk(std::integral_constant<int, 0>{});
  } else if constexpr (ONES) {
    walk(std::integral_constant<int, 2>{});
  } else {
    walk(std::integral_constant<int, 0>{});
  }

  // ---- single split (latent self-attention, short contexts): normalise and write O in its final layout
  if (!ONES && a.Ofinal != nullptr) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int tile = qg * NQ + i;
      float li = l[i];
      li += __shfl_xor(li, 16);
      li += __shfl_xor(li, 32);
      const float inv = 1.0f / li;                         // lane (g, j): row j of the tile
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ir = __shfl(inv, 4 * g + r);           // accumulator reg r of lane (g, d) belongs to row 4 g + r
        const int q = tile * 16 + 4 * g + r;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const int col = 16 * d + j;
          if (q < L && col < a.dh) a.Ofinal[((long)bi * L + q) * a.ldo + hi * a.dh + col] = O[i][d][r] * (DROP ? ir * a.drop.scale : ir);
        }
      }
      if (a.stats && g == 0 && tile * 16 + j < L) {
        a.stats[((long)bh * L + tile * 16 + j) * 2 + 0] = m[i];
        a.stats[((long)bh * L + tile * 16 + j) * 2 + 1] = li;
      }
    }
    return;
  }

  // ---- write the partial (O, m, l) of this split.  ONES: l sits in accumulator column DP-1.
  const long prow = ((long)bh * a.nsplit + split) * a.Lp;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int tile = qg * NQ + i;
    if (tile * 16 < a.Lp) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          a.Opart[(prow + tile * 16 + 4 * g + r) * DP + 16 * d + j] = DROP ? O[i][d][r] * a.drop.scale : O[i][d][r];
      if (ONES && !DROP) {
        if (g == 0) a.Mpart[prow + tile * 16 + j] = m[i];
        if (j == 15) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a.Lpart[prow + tile * 16 + 4 * g + r] = O[i][DT - 1][r];
        }
      } else {
        float li = l[i];
        li += __shfl_xor(li, 16);
        li += __shfl_xor(li, 32);
        if (g == 0) {
          a.Mpart[prow + tile * 16 + j] = m[i];
          a.Lpart[prow + tile * 16 + j] = li;
        }
      }
    }
  }
}

// development knob: HN_CORE_NQ overrides the query tiles per wave of the dp = 16 binding (2, 4 or 8)
static int nq_dt1() {
  static int v = -1;
  if (v < 0) {
    const char *e = tuning_env("HN_CORE_NQ");
    v = e ? atoi(e) : 4;
    if (v != 2 && v != 4 && v != 8) v = 4;
  }
  return v;
}
static int nq_for(int dt) { return dt == 1 ? nq_dt1() : (dt == 2 ? 2 : (dt == 4 ? 2 : 1)); }

// Forward fp32 core, dp = 16: the default of 4 query tiles per wave leaves b * h * ceil(tiles / 4) work items -- 16 at b = 1 --
// and the token split then has to make up the waves: 242 splits at b = 1, 64 at b = 4, whose merge costs more than the core
// (profiles/r03_c: merge_vproj_kernel 49 us against 34 us for the core at b = 1).  With 2 or 1 tiles per wave the same waves come
// from 12 splits or fewer, which the chain behind the block merges itself.  0 = keep the default.
static int sb_env(const char *name) { const char *e = getenv(name); return e ? atoi(e) : 0; }
int attn_core_nq_small_batch(int dp, int b, int h, int Lp) {
  static const bool off = tuning_env("HN_CORE_NQ") != nullptr || tuning_env("HN_NO_SMALL_BATCH_GEOMETRY") != nullptr;
  static const int force_nq = sb_env("HN_SB_NQ");       // development knobs: fixed tiles per wave / split cap of the small-batch plan
  if (off || dp != 16) return 0;
  if (force_nq > 0) return force_nq;
  const int tiles = Lp / 16;
  const int cand[3] = {4, 2, 1};
  // measured (cfg1, b = 1 .. 16, every (tiles per wave, split cap) pair: gpurun_out/r03u): the fastest plan has >= 128 work items
  // (b * h * query-tile groups) in front of the token split -- 1 tile per wave at b <= 2, 2 at b = 4, the default 4 from b = 8 on --
  // and as many splits (<= 48, folded by the chain's merge head in groups of 12) as fill the wave slots
  for (int i = 0; i < 3; ++i)
    if ((long)b * h * ceil_div(tiles, cand[i]) >= 128) return cand[i];
  return (long)b * h * tiles * CHAIN_MERGE_MAX_SPLITS >= 512 ? 1 : 0;
}
int attn_core_small_batch_split_cap() {
  static const int cap = sb_env("HN_SB_SPLITS");
  return cap > 0 && cap <= CHAIN_MERGE_MAX_SPLITS ? cap : CHAIN_MERGE_MAX_SPLITS;
}

void attn_core_geometry(int b, int h, int Lp, int N, int dp, int *nsplit, int *chunk, int waves_per_simd, int nq) {
  const int dt = dp / 16;
  const int ngroups = ceil_div(Lp / 16, nq > 0 ? nq : nq_for(dt));
  static long target_waves = 0;   // development knob HN_CORE_WAVES: resident waves the token split aims for
  if (target_waves == 0) { const char *e = tuning_env("HN_CORE_WAVES"); target_waves = e ? atol(e) : 256L * 4 * 4; if (target_waves < 64) target_waves = 4096; }
  const long tw = waves_per_simd > 0 ? 256L * 4 * waves_per_simd : target_waves;
  static int geom_floor = -1;      // development knob HN_GEOM_CEIL=1: the old rounding
  if (geom_floor < 0) { const char *e = tuning_env("HN_GEOM_CEIL"); geom_floor = (e && e[0] == '1') ? 0 : 1; }
  // floor: all waves resident in ONE round (a ceil that overshoots the slots by a few waves costs a whole second round)
  long want = geom_floor ? tw / ((long)b * h * ngroups) : ceil_div_ll(tw, (long)b * h * ngroups);
  long max_splits = N / 128;
  if (max_splits < 1) max_splits = 1;
  if (nq > 0 && want > attn_core_small_batch_split_cap()) want = attn_core_small_batch_split_cap();      // planned for the chain's merge head (attn_core_nq_small_batch)
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  int c = (int)ceil_div_ll(N, want);
  c = ceil_div(c, 16) * 16;
  *chunk = c;
  *nsplit = ceil_div(N, c);
}

int launch_attn_core(const AttnCoreArgs &a, hipStream_t s) {
  HN_REQUIRE(a.dp == 16 || a.dp == 32 || a.dp == 64 || a.dp == 128, HN_E_UNSUPPORTED, "attn_core: dp=%d", a.dp);
  HN_REQUIRE(a.Lp % 16 == 0 && a.chunk % 16 == 0 && a.nsplit >= 1, HN_E_SHAPE, "attn_core: Lp=%d chunk=%d", a.Lp, a.chunk);
  HN_REQUIRE((a.ldq % 4) == 0 && (a.ldk % 4) == 0, HN_E_SHAPE, "attn_core: ldq=%d ldk=%d must be multiples of 4", a.ldq, a.ldk);
  HN_REQUIRE(((long)a.N * a.ldk + a.dp) * 4 < (1L << 31) && ((long)a.N * a.ldv + a.dp) * 4 < (1L << 31), HN_E_UNSUPPORTED,
             "attn_core: one sample's K/V rows must span < 2 GiB (N=%d ld=%d)", a.N, a.ldk);
  HN_REQUIRE(a.Ofinal == nullptr || (a.nsplit == 1 && !a.ones_col), HN_E_SHAPE, "attn_core: direct output needs a single split");
  HN_REQUIRE(!a.ones_col || (a.dp <= 32 && a.Kp == a.Vp), HN_E_UNSUPPORTED, "attn_core: ones column needs the shared-context binding");
  HN_REQUIRE(a.drop.thr == 0 || !a.ones_col || a.dp == 16 || a.dp == 32, HN_E_SHAPE,
             "attn_core: dropout with the ones column (the row-sum channel) runs on dp = 16 / 32");
  if (self_core_lds_eligible(a)) return launch_self_core_lds(a, s);
  const int dt = a.dp / 16;
  int nq = a.nq > 0 ? a.nq : nq_for(dt);
  // latent self-attention (dp = 64, one split) at small batches: fewer than one wave per two SIMDs with 2 tiles per wave;
  // one tile per wave doubles the resident waves (b = 1: 0.92 -> 0.90 ms per forward; no gain from b = 32 on)
  if (dt == 4 && a.nsplit == 1 && !a.ones_col && a.drop.thr == 0 && (long)a.b * a.h * ceil_div(a.Lp / 16, 2) <= 512) nq = 1;
  const int ngroups = ceil_div(a.Lp / 16, nq);
  const int wpb = ngroups < 4 ? ngroups : 4;
  const int gy = ceil_div(ngroups, wpb);
  const long blocks = (long)a.nsplit * gy * a.b * a.h;
  HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "attn_core: grid too large");
  dim3 grid((unsigned)blocks), block(64 * wpb);
  const int ks = a.qk_steps > 0 ? a.qk_steps : 4 * dt;
  HN_REQUIRE(ks >= 1 && ks <= 4 * dt && (ks == 4 * dt || (a.ones_col && a.ones_in_mem)), HN_E_SHAPE, "attn_core: qk_steps=%d", a.qk_steps);
#define HN_CORE(DT_, NQ_, ONES_, KS_) hipLaunchKernelGGL((attn_core_kernel<DT_, NQ_, ONES_, KS_>), grid, block, 0, s, a, ngroups, gy, wpb)
  if (a.drop.thr != 0 && a.ones_col) {
    // dropout on the shared-context binding with the score-bound softmax: the reference of a row is fixed (no running maximum, no
    // rescale), the ones column -- injected in registers -- is the row-sum channel, the denominator is summed on the VALU
    HN_REQUIRE((dt == 1 && nq == 4) || (dt == 2 && nq == 2), HN_E_UNSUPPORTED, "attn_core: dropout variant dp=%d nq=%d", a.dp, nq);
    // (round 4: the packed context -- ks < 4 dt k-steps, the ones column in memory -- under dropout as well; the thinned row sum
    // rides in the ones column's accumulator either way)
#define HN_CORE_DROP_ONES(DT_, NQ_, KS_) hipLaunchKernelGGL((attn_core_kernel<DT_, NQ_, true, KS_, true>), grid, block, 0, s, a, ngroups, gy, wpb)
    if (dt == 1) {
      if (ks == 1) HN_CORE_DROP_ONES(1, 4, 1); else if (ks == 2) HN_CORE_DROP_ONES(1, 4, 2); else if (ks == 3) HN_CORE_DROP_ONES(1, 4, 3);
      else HN_CORE_DROP_ONES(1, 4, 4);
    } else {
      if (ks == 5) HN_CORE_DROP_ONES(2, 2, 5); else if (ks == 6) HN_CORE_DROP_ONES(2, 2, 6); else if (ks == 7) HN_CORE_DROP_ONES(2, 2, 7);
      else { HN_REQUIRE(ks == 8, HN_E_SHAPE, "attn_core: dropout variant qk_steps=%d dp=32", ks); HN_CORE_DROP_ONES(2, 2, 8); }
    }
#undef HN_CORE_DROP_ONES
  } else if (dt == 1 && a.ones_col && (nq == 2 || nq == 1) && ks != 4) {
    HN_REQUIRE(ks >= 1 && ks <= 3, HN_E_SHAPE, "attn_core: qk_steps=%d", ks);
    if (nq == 2) { if (ks == 1) HN_CORE(1, 2, true, 1); else if (ks == 2) HN_CORE(1, 2, true, 2); else HN_CORE(1, 2, true, 3); }
    else { if (ks == 1) HN_CORE(1, 1, true, 1); else if (ks == 2) HN_CORE(1, 1, true, 2); else HN_CORE(1, 1, true, 3); }
  } else if (dt == 1 && a.ones_col) {
    if (nq == 8) HN_CORE(1, 8, true, 4);
    else if (nq == 2) HN_CORE(1, 2, true, 4);
    else if (nq == 1) HN_CORE(1, 1, true, 4);
    else if (ks == 1) HN_CORE(1, 4, true, 1);
    else if (ks == 2) HN_CORE(1, 4, true, 2);
    else if (ks == 3) HN_CORE(1, 4, true, 3);
    else HN_CORE(1, 4, true, 4);
  } else if (a.drop.thr != 0) {
#define HN_CORE_DROP(DT_, NQ_) hipLaunchKernelGGL((attn_core_kernel<DT_, NQ_, false, 4 * DT_, true>), grid, block, 0, s, a, ngroups, gy, wpb)
    HN_REQUIRE(dt != 1 || nq == 4, HN_E_UNSUPPORTED, "attn_core: dropout variant is built for NQ = 4");
    if (dt == 1) HN_CORE_DROP(1, 4);
    else if (dt == 2) HN_CORE_DROP(2, 2);
    else if (dt == 4) HN_CORE_DROP(4, 2);
    else HN_CORE_DROP(8, 1);
#undef HN_CORE_DROP
  } else if (dt == 1) {
    if (nq == 8) HN_CORE(1, 8, false, 4);
    else if (nq == 2) HN_CORE(1, 2, false, 4);
    else if (nq == 1) HN_CORE(1, 1, false, 4);
    else HN_CORE(1, 4, false, 4);
  } else if (dt == 2 && a.ones_col) {
    if (ks == 4) HN_CORE(2, 2, true, 4);
    else if (ks == 5) HN_CORE(2, 2, true, 5);
    else if (ks == 6) HN_CORE(2, 2, true, 6);
    else if (ks == 7) HN_CORE(2, 2, true, 7);
    else HN_CORE(2, 2, true, 8);
  } else if (dt == 2) {
    HN_CORE(2, 2, false, 8);
  } else if (dt == 4) {
    if (nq == 1) HN_CORE(4, 1, false, 16);
    else HN_CORE(4, 2, false, 16);
  } else {
    HN_CORE(8, 1, false, 32);
  }
#undef HN_CORE
  HN_LAUNCH_CHECK("attn_core");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// rank-D path, query side: Qf[b,h,q,d] = cscale * gamma[d] * sum_e Q[b,q,h*dh+e] * W_k[h*dh+e, d]
// (S = Q_h K_h^T with K = (z*gamma + beta) W_k^T; the beta term is constant along the softmax axis and
// cancels).  Tiny: L*dp outputs of a dh-long dot product per (sample, head).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qfold_kernel(const float *__restrict__ Q, int ldq_row, const float *__restrict__ w_k,
                                                    int D, const float *__restrict__ gamma, float cscale,
                                                    float *__restrict__ Qf, int h, int L, int Lp, int dh, int dp, int pack_ks,
                                                    float *__restrict__ bound, int *__restrict__ bound_flag) {
  extern __shared__ float wk[];  // [dh][dp]
  const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
  for (int idx = threadIdx.x; idx < dh * dp; idx += blockDim.x) {
    const int e = idx / dp, d = idx % dp;
    const float *wr = w_k + (long)(hi * dh + e) * D;
    float w = 0.0f;
    if (pack_ks == 0) {
      if (d < D) w = wr[d] * (gamma ? gamma[d] : 1.0f);
    } else {
      // packed layout: column d holds kept channel c with weight (gamma_c W_c - gamma_{D-1} W_{D-1}): the context
      // rows sum to zero, so the last channel's score contribution moves onto the others
      const int c = packed_chan(d, pack_ks);
      if (c >= 0 && c < D - 1) w = wr[c] * (gamma ? gamma[c] : 1.0f) - wr[D - 1] * (gamma ? gamma[D - 1] : 1.0f);
    }
    wk[idx] = w * cscale;
  }
  __syncthreads();
  float *dst = Qf + (long)bh * Lp * dp;
  const int dlim = pack_ks ? dp : D;
  for (int idx = threadIdx.x; idx < Lp * dp; idx += blockDim.x) {
    const int q = idx / dp, d = idx % dp;
    float acc = 0.0f;
    if (q < L && d < dlim) {
      const float *qr = Q + ((long)bi * L + q) * ldq_row + hi * dh;
      for (int e = 0; e < dh; ++e) acc = fmaf(qr[e], wk[e * dp + d], acc);
    }
    dst[idx] = acc;
  }
  // Score bound per query row (Cauchy-Schwarz): the context rows are affine-free LayerNorm outputs, |z|^2 = D var/(var+eps)
  // <= D, so every score s = q . z of this row obeys |s| <= |q| sqrt(D).  With that bound as the softmax reference the
  // attention core needs neither a running max nor an overflow guard (p <= 1 always).  The bound is only used while
  // 2^(-2 bound) stays far inside the fp32 range: rows beyond kMaxBound raise a flag and the core falls back to the
  // running reference for the whole launch (deterministic: the flag is an OR).
  if (bound != nullptr) {
    __syncthreads();                                   // this workgroup's Qf rows are complete (and visible to it)
    for (int q = threadIdx.x; q < Lp; q += blockDim.x) {
      float ss = 0.0f;
      for (int d = 0; d < dp; ++d) { const float v = dst[q * dp + d]; ss = fmaf(v, v, ss); }
      const float bq = sqrtf(ss * (float)D) * 1.00002f + 1e-6f;
      bound[(long)bh * Lp + q] = bq;
      if (bq > 60.0f) atomicOr(bound_flag, 1);
    }
  }
}

// Matrix-core form of the same fold for the aligned case (dh a multiple of 16, 16-byte aligned query rows): per (sample,
// head) a (Lp x dh) . (dh x dp) product -- 128 MFMAs instead of 2048 x 64 scalar FMAs fed by uncoalesced global reads
// (17 -> ~6 us at cfg2).  Contraction order permuted like the attention core's (step (c, r) <-> channel 16 c + 4 g + r), so
// every A fragment is one 16-byte load; B comes from the staged weight image; the row bound falls out of the accumulators.
// BF16: the folded queries leave as bf16 (round to nearest even) for the bf16 core -- 32 slots per row, natural channel order --
// with the 1 % slack on the row bound that covers the bf16 rounding of q and z (attention_bf16.hip).
template <int CT, bool BF16 = false>      // 16-column tiles of the folded query (dp = 16 CT)
__global__ __launch_bounds__(256) void qfold_mfma_kernel(const float *__restrict__ Q, int ldq_row, const float *__restrict__ w_k,
                                                         int D, const float *__restrict__ gamma, float cscale,
                                                         float *__restrict__ Qf, int h, int L, int Lp, int dh, int pack_ks,
                                                         float *__restrict__ bound, int *__restrict__ bound_flag) {
  constexpr int dp = 16 * CT;
  extern __shared__ float wk[];  // [dh][dp]
  const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
  for (int idx = threadIdx.x; idx < dh * dp; idx += blockDim.x) {
    const int e = idx / dp, d = idx % dp;
    const float *wr = w_k + (long)(hi * dh + e) * D;
    float w = 0.0f;
    if (pack_ks == 0) {
      if (d < D) w = wr[d] * (gamma ? gamma[d] : 1.0f);
    } else {
      const int c = packed_chan(d, pack_ks);
      if (c >= 0 && c < D - 1) w = wr[c] * (gamma ? gamma[c] : 1.0f) - wr[D - 1] * (gamma ? gamma[D - 1] : 1.0f);
    }
    wk[idx] = w * cscale;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  float *dst = Qf + (long)bh * Lp * dp;
  const int nc = dh / 16;
  for (int tile = wave; tile * 16 < Lp; tile += 4) {
    const int row = tile * 16 + j;
    const bool live = row < L;
    const float *qr = Q + ((long)bi * L + (live ? row : 0)) * ldq_row + hi * dh + 4 * g;
    f32x4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nc; ++c) {
      float4 a = *(const float4 *)(qr + 16 * c);
      if (!live) a = make_float4(0.f, 0.f, 0.f, 0.f);
      const float *wb = wk + (16 * c + 4 * g) * dp + j;       // B[k = g][col j] of step (c, r) = wk[16 c + 4 g + r][col]
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wb[16 * t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wb[dp + 16 * t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wb[2 * dp + 16 * t], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wb[3 * dp + 16 * t], acc[t], 0, 0, 0);
      }
    }
    // accumulator reg r of lane (g, j): row tile*16 + 4 g + r, column 16 t + j
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = tile * 16 + 4 * g + r;
        if (BF16) {
          unsigned u = __float_as_uint(acc[t][r]);
          u += 0x7fffu + ((u >> 16) & 1u);
          if (orow < Lp) ((uint16_t *)Qf)[((long)bh * Lp + orow) * dp + 16 * t + j] = (uint16_t)(u >> 16);
        } else if (orow < Lp) {
          dst[(long)orow * dp + 16 * t + j] = acc[t][r];
        }
        ss[r] = fmaf(acc[t][r], acc[t][r], ss[r]);
      }
    if (bound != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = ss[r];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        const int orow = tile * 16 + 4 * g + r;
        if (j == 0 && orow < Lp) {
          const float bq = sqrtf(v * (float)D) * (BF16 ? 1.01f : 1.00002f) + 1e-6f;
          bound[(long)bh * Lp + orow] = bq;
          if (bq > 60.0f) atomicOr(bound_flag, 1);
        }
      }
    }
  }
}

// bf16 folded queries on the matrix cores (plain bf16 core, one operand plane): false when the shape needs qfold_bf16_kernel
bool launch_qfold_mfma_bf16(const float *Q, int ldq_row, const float *w_k, int D, const float *gamma, float cscale, uint16_t *Qf,
                            int b, int h, int L, int Lp, int dh, hipStream_t s, float *bound, int *bound_flag) {
  if (!(dh % 16 == 0 && (ldq_row & 3) == 0 && ((uintptr_t)Q & 15) == 0 && D <= 32)) return false;
  hipLaunchKernelGGL((qfold_mfma_kernel<2, true>), dim3(b * h), dim3(256), (size_t)dh * 32 * sizeof(float), s, Q, ldq_row, w_k, D, gamma, cscale,
                     (float *)Qf, h, L, Lp, dh, 0, bound, bound_flag);
  return true;
}

int launch_qfold(const float *Q, int ldq_row, const float *w_k, int D, const float *gamma, float cscale, float *Qf,
                 int b, int h, int L, int Lp, int dh, int dp, hipStream_t s, int pack_ks, float *bound, int *bound_flag) {
  size_t lds = (size_t)dh * dp * sizeof(float);
  HN_REQUIRE((bound == nullptr) == (bound_flag == nullptr), HN_E_NULL, "qfold: bound and bound_flag go together");
  if ((dp == 16 || dp == 32) && dh % 16 == 0 && (ldq_row & 3) == 0 && ((uintptr_t)Q & 15) == 0) {
    if (dp == 16) hipLaunchKernelGGL(qfold_mfma_kernel<1>, dim3(b * h), dim3(256), lds, s, Q, ldq_row, w_k, D, gamma, cscale, Qf, h, L, Lp, dh,
                                     pack_ks, bound, bound_flag);
    else hipLaunchKernelGGL(qfold_mfma_kernel<2>, dim3(b * h), dim3(256), lds, s, Q, ldq_row, w_k, D, gamma, cscale, Qf, h, L, Lp, dh,
                            pack_ks, bound, bound_flag);
    HN_LAUNCH_CHECK("qfold_mfma");
    return HN_OK;
  }
  hipLaunchKernelGGL(qfold_kernel, dim3(b * h), dim3(256), lds, s, Q, ldq_row, w_k, D, gamma, cscale, Qf, h, L, Lp, dh, dp, pack_ks,
                     bound, bound_flag);
  HN_LAUNCH_CHECK("qfold");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// merge of the token splits (fixed order) and, for the rank-D path, the value projection
//   O'[q,:] = sum_s 2^(m_s - M) O_s[q,:] / sum_s 2^(m_s - M) l_s ;   O_h = (O' * gamma + beta) W_v,h^T
// ------------------------------------------------------------------------------------------------
constexpr int MERGE_ROWS = 32;

// Split merge in two phases per chunk of MERGE_MAXS splits: 8 lanes per latent row turn the per-split (m, l) pairs into
// weights w_s = 2^(m_s - M) (staged in LDS) and the merged (M, l), then every output element is one pass of independent
// coalesced loads over the chunk.  (The one-phase form re-read m and l for every element: 4 * nsplit dependent loads.)
// R rows per workgroup: 32, or 8 when that would leave most of the chip idle (small batches: few (b, h) pairs, hundreds of
// splits -- at b = 1 the merge of 256 splits took 121 us per image block, a third of the forward).
constexpr int MERGE_MAXS = 64;
struct MergeWeights {
  float w[MERGE_ROWS][MERGE_MAXS + 1];
  float M[MERGE_ROWS], invl[MERGE_ROWS], l[MERGE_ROWS];
};

// Accumulates acc[k] (element idx = tid + 256 k of the R x width block; element (qq, d) is taken when keep(d)) over all
// splits, unnormalised; leaves M / l / 1/l of the block's rows in mw.  All 256 threads must call it.
// Every load is unconditional on a clamped address (rows up to Lp exist in the partials) and issued in groups of PF that are in
// flight together; the first group leaves BEFORE the weight phase.  The partials were written by the core on other XCDs, so each
// load is an HBM / Infinity-Cache round trip: the per-split loop with one dependent load per iteration this replaces spent
// nsplit + 4 of them in sequence (merge_vproj 19.5 us at 9 splits, merge_explicit 42 us at cfg4).
template <int EPT, typename Keep>
__device__ __forceinline__ void merge_splits(MergeWeights &mw, const float *Opart, const float *Mpart, const float *Lpart,
                                             long pbase, int nsplit, int Lp, int dp, int q0, int L, int R, int width, Keep keep,
                                             float (&acc)[EPT]) {
  constexpr int PF = EPT == 1 ? 16 : (EPT <= 4 ? 8 : 4);   // splits per group (EPT * PF values in registers)
  const int tid = threadIdx.x;
  const long sstride = (long)Lp * dp;
  long eoff[EPT];
  int eq[EPT];
  bool live[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int idx = tid + 256 * k;
    const bool inb = idx < R * width;
    const int qq = inb ? idx / width : 0, d = inb ? idx % width : 0;
    eq[k] = qq;
    live[k] = inb && q0 + qq < L && keep(d);
    eoff[k] = (pbase + min(q0 + qq, Lp - 1)) * dp + d;
    acc[k] = 0.0f;
  }
  float pre[EPT][PF];
  auto request = [&](int s) {                            // splits s .. s + PF - 1 (past the end: the last one again, weight 0)
#pragma unroll
    for (int k = 0; k < EPT; ++k)
#pragma unroll
      for (int u = 0; u < PF; ++u) pre[k][u] = Opart[eoff[k] + (long)min(s + u, nsplit - 1) * sstride];
  };
  request(0);

  const int wq = tid >> 3, j = tid & 7;                  // weight phase: 8 lanes per row, rows 0 .. R-1
  const bool wlive = wq < R && q0 + wq < L;
  const long wrow = pbase + min(q0 + wq, Lp - 1);
  float M = kNegBig;
  if (wlive) for (int s = j; s < nsplit; s += 32) {      // four independent loads per trip (clamped: a repeated split does not move a maximum)
    float m4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) m4[u] = Mpart[wrow + (long)min(s + 8 * u, nsplit - 1) * Lp];
#pragma unroll
    for (int u = 0; u < 4; ++u) M = fmaxf(M, m4[u]);
  }
  M = fmaxf(M, __shfl_xor(M, 1));
  M = fmaxf(M, __shfl_xor(M, 2));
  M = fmaxf(M, __shfl_xor(M, 4));
  float lsum = 0.0f;
  for (int s0 = 0; s0 < nsplit; s0 += MERGE_MAXS) {
    const int s1 = min(nsplit, s0 + MERGE_MAXS);
    const int s1p = min(s0 + MERGE_MAXS, (s1 + PF - 1) / PF * PF);       // weights up to the end of the last group: 0
    const bool last = s1 == nsplit;
    if (wq < R) {                                        // the lane's <= 8 splits of the chunk: all (m, l) pairs requested before the first use
      float mv[MERGE_MAXS / 8], lv[MERGE_MAXS / 8];
#pragma unroll
      for (int u = 0; u < MERGE_MAXS / 8; ++u) {
        const long at = wrow + (long)min(s0 + j + 8 * u, nsplit - 1) * Lp;
        mv[u] = Mpart[at];
        lv[u] = Lpart[at];
      }
#pragma unroll
      for (int u = 0; u < MERGE_MAXS / 8; ++u) {
        const int s = s0 + j + 8 * u;
        if (s < s1p) {
          float w = 0.0f;
          if (wlive && s < s1) {
            w = fast_exp2(mv[u] - M);
            lsum = fmaf(w, lv[u], lsum);
          }
          mw.w[wq][s - s0] = w;
        }
      }
    }
    if (last) {           // the merged (M, l) ride on the barrier the last chunk needs anyway: one barrier in all for <= 64 splits
      lsum += __shfl_xor(lsum, 1);
      lsum += __shfl_xor(lsum, 2);
      lsum += __shfl_xor(lsum, 4);
      if (wq < R && j == 0) { mw.M[wq] = M; mw.l[wq] = lsum; mw.invl[wq] = 1.0f / lsum; }
    }
    __syncthreads();
    for (int s = s0; s < s1; s += PF) {
      if (s != 0) request(s);
#pragma unroll
      for (int k = 0; k < EPT; ++k)
#pragma unroll
        for (int u = 0; u < PF; ++u) acc[k] = fmaf(mw.w[eq[k]][s + u - s0], pre[k][u], acc[k]);
    }
    if (!last) __syncthreads();      // the next chunk overwrites the staged weights
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) acc[k] = live[k] ? acc[k] : 0.0f;
}

static int merge_rows_per_block(int b, int h, int L) {       // 32 rows, or 8 while 32 would give fewer workgroups than CUs
  return (long)b * h * ceil_div(L, MERGE_ROWS) >= 256 ? MERGE_ROWS : 8;
}
// The explicit merge has no per-workgroup staging (merge_vproj_kernel loads the folded value projection into LDS first: fewer,
// fatter workgroups pay there), so it takes the 8-row form up to FOUR workgroups per CU: a merge is a few groups of dependent
// loads, and at one workgroup per CU -- cfg4: 256 workgroups of 32 rows -- nothing hides them (13.8 us for 8 splits).
static int merge_rows_per_block_explicit(int b, int h, int L) {
  return (long)b * h * ceil_div(L, MERGE_ROWS) >= 1024 ? MERGE_ROWS : 8;
}

__global__ __launch_bounds__(256) void merge_vproj_kernel(const float *__restrict__ Opart, const float *__restrict__ Mpart,
                                                          const float *__restrict__ Lpart, int nsplit, int h, int L, int Lp,
                                                          int dp, int D, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, const float *__restrict__ w_v,
                                                          int dh, float *__restrict__ O, int ldo, float *__restrict__ stats,
                                                          float *__restrict__ oprime_save, int pack_ks, int srow, int R) {
  extern __shared__ float sm[];
  float *oh = sm;                        // [MERGE_ROWS][dp + 1]
  float *wv = sm + MERGE_ROWS * (dp + 1);  // [dh][dp + 1]: gamma folded in; [dp] = the beta term
  const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
  const int q0 = blockIdx.y * R;
  const long pbase = (long)bh * nsplit * Lp;
  const int dlim = pack_ks ? dp : D;
  __shared__ MergeWeights mw;
  // the head's W_v block (dh x D, contiguous) and the affine vectors come in with coalesced loads FIRST -- they overlap the split
  // merge -- and the folded projection image is built from LDS afterwards (a per-element loop over D with dependent global
  // loads used to be this kernel's longest chain)
  float *raw = wv + dh * (dp + 1);       // [dh][D]
  float *gb = raw + dh * D;              // gamma[D], beta[D]
  for (int idx = threadIdx.x; idx < dh * D; idx += blockDim.x) raw[idx] = w_v[(long)hi * dh * D + idx];
  for (int idx = threadIdx.x; idx < D; idx += blockDim.x) { gb[idx] = gamma ? gamma[idx] : 1.0f; gb[D + idx] = beta ? beta[idx] : 0.0f; }
  float acc[4];                          // R * dp <= 32 * 32 elements
  // srow: column dp-1 carries the dropped row sum (see attn_core)
  merge_splits<4>(mw, Opart, Mpart, Lpart, pbase, nsplit, Lp, dp, q0, L, R, dp,
                  [&](int d) { return d < dlim || (srow && d == dp - 1); }, acc);
  if (stats && threadIdx.x < R && q0 + threadIdx.x < L) {
    stats[((long)bh * L + q0 + threadIdx.x) * 2 + 0] = mw.M[threadIdx.x];
    stats[((long)bh * L + q0 + threadIdx.x) * 2 + 1] = mw.l[threadIdx.x];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = threadIdx.x + 256 * k;
    if (idx < R * dp) {
      const int qq = idx / dp, d = idx % dp, q = q0 + qq;
      const bool kept = q < L && (d < dlim || (srow && d == dp - 1));
      const float v = kept ? acc[k] * mw.invl[qq] : 0.0f;
      if (oprime_save && !pack_ks && q < L) oprime_save[((long)bi * L + q) * (h * dp) + hi * dp + d] = v;   // training: normalised P z, padding columns 0
      oh[qq * (dp + 1) + d] = v;
    }
  }
  for (int idx = threadIdx.x; idx < dh * (dp + 1); idx += blockDim.x) {
    const int e = idx / (dp + 1), d = idx % (dp + 1);
    const float *wr = raw + e * D;
    float w = 0.0f;
    if (d == dp) {
      for (int c = 0; c < D; ++c) w = fmaf(gb[D + c], wr[c], w);
    } else if (pack_ks == 0) {
      if (d < D) w = wr[d] * gb[d];
    } else {
      // packed layout: the dropped channel of the averaged row is minus the sum of the kept ones
      const int c = packed_chan(d, pack_ks);
      if (c >= 0 && c < D - 1) w = wr[c] * gb[c] - wr[D - 1] * gb[D - 1];
    }
    wv[idx] = w;
  }
  __syncthreads();
  if (oprime_save && pack_ks) {     // the tape keeps the natural channel layout: kept channels from their slots, the dropped one = -sum
    for (int idx = threadIdx.x; idx < R * dp; idx += blockDim.x) {
      const int qq = idx / dp, c = idx % dp, q = q0 + qq;
      if (q >= L) continue;
      float v = 0.0f;
      if (c < D - 1) v = oh[qq * (dp + 1) + packed_slot(c, pack_ks)];
      else if (c == D - 1) {
        for (int k = 0; k < D - 1; ++k) v -= oh[qq * (dp + 1) + packed_slot(k, pack_ks)];
      } else if (srow && c == dp - 1) v = oh[qq * (dp + 1) + dp - 1];      // the thinned row sum keeps its column
      oprime_save[((long)bi * L + q) * (h * dp) + hi * dp + c] = v;
    }
  }
  for (int idx = threadIdx.x; idx < R * dh; idx += blockDim.x) {
    const int qq = idx / dh, e = idx % dh, q = q0 + qq;
    if (q >= L) continue;
    float a = wv[e * (dp + 1) + dp] * (srow ? oh[qq * (dp + 1) + dp - 1] : 1.0f);      // beta term * sum_t p'_t (1 without dropout)
    for (int d = 0; d < dlim; ++d) a = fmaf(oh[qq * (dp + 1) + d], wv[e * (dp + 1) + d], a);
    O[((long)bi * L + q) * ldo + hi * dh + e] = a;
  }
}

int launch_merge_vproj(const float *Opart, const float *Mpart, const float *Lpart, int nsplit, int b, int h, int L,
                       int Lp, int dp, int D, const float *gamma, const float *beta, const float *w_v, int dh,
                       float *O, int ldo, float *stats, float *oprime_save, hipStream_t s, int pack_ks, int srow) {
  size_t lds = ((size_t)MERGE_ROWS * (dp + 1) + (size_t)dh * (dp + 1) + (size_t)dh * D + 2 * (size_t)D) * sizeof(float);
  HN_REQUIRE(dp <= 32, HN_E_UNSUPPORTED, "merge_vproj: dp=%d", dp);
  const int R = merge_rows_per_block(b, h, L);
  hipLaunchKernelGGL(merge_vproj_kernel, dim3(b * h, ceil_div(L, R)), dim3(256), lds, s, Opart, Mpart, Lpart,
                     nsplit, h, L, Lp, dp, D, gamma, beta, w_v, dh, O, ldo, stats, oprime_save, pack_ks, srow, R);
  HN_LAUNCH_CHECK("merge_vproj");
  return HN_OK;
}

// EPT = output elements per thread (R * dh <= 256 EPT): one narrow head over few rows (R = 8, dim_head 27: 216 elements) takes
// EPT = 1 and sixteen splits in flight per trip -- with the 16-element form it issued 15 dead loads per live one and spent
// nsplit / 4 dependent round trips (14 us at 64 splits)
template <int EPT>
__global__ __launch_bounds__(256) void merge_explicit_kernel(const float *__restrict__ Opart, const float *__restrict__ Mpart,
                                                             const float *__restrict__ Lpart, int nsplit, int h, int L,
                                                             int Lp, int dp, int dh, float *__restrict__ O, int ldo,
                                                             float *__restrict__ stats, int R) {
  const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
  const int q0 = blockIdx.y * R;
  const long pbase = (long)bh * nsplit * Lp;
  __shared__ MergeWeights mw;
  float acc[EPT];                        // R * dh <= 32 * 128 elements
  merge_splits<EPT>(mw, Opart, Mpart, Lpart, pbase, nsplit, Lp, dp, q0, L, R, dh, [](int) { return true; }, acc);
  if (stats && threadIdx.x < R && q0 + threadIdx.x < L) {
    stats[((long)bh * L + q0 + threadIdx.x) * 2 + 0] = mw.M[threadIdx.x];
    stats[((long)bh * L + q0 + threadIdx.x) * 2 + 1] = mw.l[threadIdx.x];
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int idx = threadIdx.x + 256 * k;
    if (idx < R * dh) {
      const int qq = idx / dh, e = idx % dh, q = q0 + qq;
      if (q < L) O[((long)bi * L + q) * ldo + hi * dh + e] = acc[k] * mw.invl[qq];
    }
  }
}

int launch_merge_explicit(const float *Opart, const float *Mpart, const float *Lpart, int nsplit, int b, int h, int L,
                          int Lp, int dp, int dh, float *O, int ldo, float *stats, hipStream_t s) {
  HN_REQUIRE(dh <= 128, HN_E_UNSUPPORTED, "merge_explicit: dh=%d", dh);
  const int R = merge_rows_per_block_explicit(b, h, L);
  const dim3 grid(b * h, ceil_div(L, R));
  if (R * dh <= 256)
    hipLaunchKernelGGL(merge_explicit_kernel<1>, grid, dim3(256), 0, s, Opart, Mpart, Lpart, nsplit, h, L, Lp, dp, dh, O, ldo, stats, R);
  else if (R * dh <= 1024)
    hipLaunchKernelGGL(merge_explicit_kernel<4>, grid, dim3(256), 0, s, Opart, Mpart, Lpart, nsplit, h, L, Lp, dp, dh, O, ldo, stats, R);
  else
    hipLaunchKernelGGL(merge_explicit_kernel<16>, grid, dim3(256), 0, s, Opart, Mpart, Lpart, nsplit, h, L, Lp, dp, dh, O, ldo, stats, R);
  HN_LAUNCH_CHECK("merge_explicit");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// Attention.attn_weights (:420) on demand:  P[bh, q, t] = 2^(Q[q,:].K[t,:] - M_q) / L_q  (0 when masked)
// Opt-in export path (the reference always keeps this (b*h, L, N) tensor; 6.6 GB per image block at
// b = 32).  Plain VALU kernel, 64 tokens x all rows per workgroup.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void probs_kernel(const float *__restrict__ Q, long q_b, long q_h, int ldq, int dp,
                                                    const float *__restrict__ Kp, long k_b, long k_h, int ldk,
                                                    const uint8_t *__restrict__ mask,
                                                    const float *__restrict__ stats, float *__restrict__ P, int h, int L,
                                                    int N) {
  extern __shared__ float ks[];  // [64][dp + 1]
  const int bh = blockIdx.y, bi = bh / h, hi = bh % h;
  const int t0 = blockIdx.x * 64;
  const float *kbase = Kp + (long)bi * k_b + (long)hi * k_h;
  for (int idx = threadIdx.x; idx < 64 * dp; idx += blockDim.x) {
    const int tt = idx / dp, d = idx % dp;
    const int t = min(t0 + tt, N - 1);
    ks[tt * (dp + 1) + d] = kbase[(long)t * ldk + d];
  }
  __syncthreads();
  const int tt = threadIdx.x & 63, t = t0 + tt;
  const bool live = t < N && (!mask || mask[(long)bi * N + t] != 0);
  const float *qb = Q + (long)bi * q_b + (long)hi * q_h;
  for (int q = threadIdx.x >> 6; q < L; q += 4) {
    float acc = 0.0f;
    for (int d = 0; d < dp; ++d) acc = fmaf(qb[(long)q * ldq + d], ks[tt * (dp + 1) + d], acc);
    const float M = stats[((long)bh * L + q) * 2 + 0], Ls = stats[((long)bh * L + q) * 2 + 1];
    if (t < N) P[((long)bh * L + q) * N + t] = live ? fast_exp2(acc - M) / Ls : 0.0f;
  }
}

int launch_probs(const float *Q, long q_b, long q_h, int ldq, int dp, const float *Kp, long k_b, long k_h, int ldk,
                 const uint8_t *mask, const float *stats, float *P, int b, int h, int L, int N, hipStream_t s) {
  size_t lds = (size_t)64 * (dp + 1) * sizeof(float);
  hipLaunchKernelGGL(probs_kernel, dim3(ceil_div(N, 64), b * h), dim3(256), lds, s, Q, q_b, q_h, ldq, dp, Kp, k_b, k_h, ldk,
                     mask, stats, P, h, L, N);
  HN_LAUNCH_CHECK("probs");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// Reduced export for the explainer: every consumer of Attention.attn_weights in the reference
// takes the mean over the latent rows first (healnet/models/explainer.py:161-164, :209-211: `torch.mean(w, dim=1)`),
//   I[bh, t] = 1/L * sum_q P[bh, q, t]
// so this kernel emits the (b*h, N) vector straight from the saved softmax statistics instead of the
// (b*h, L, N) matrix (6.6 GB per image block at b = 32).  Same tiling as probs_kernel; the four row groups of a
// workgroup are summed in a fixed order through LDS (deterministic).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void importance_kernel(const float *__restrict__ Q, long q_b, long q_h, int ldq, int dp,
                                                         const float *__restrict__ Kp, long k_b, long k_h, int ldk,
                                                         const uint8_t *__restrict__ mask, const float *__restrict__ stats,
                                                         float *__restrict__ I, int h, int L, int N) {
  extern __shared__ float ks[];  // [64][dp + 1] + [4][64]
  float *part = ks + 64 * (dp + 1);
  const int bh = blockIdx.y, bi = bh / h, hi = bh % h;
  const int t0 = blockIdx.x * 64;
  const float *kbase = Kp + (long)bi * k_b + (long)hi * k_h;
  for (int idx = threadIdx.x; idx < 64 * dp; idx += blockDim.x) {
    const int tt = idx / dp, d = idx % dp;
    const int t = min(t0 + tt, N - 1);
    ks[tt * (dp + 1) + d] = kbase[(long)t * ldk + d];
  }
  __syncthreads();
  const int tt = threadIdx.x & 63, t = t0 + tt, grp = threadIdx.x >> 6;
  const bool live = t < N && (!mask || mask[(long)bi * N + t] != 0);
  const float *qb = Q + (long)bi * q_b + (long)hi * q_h;
  float sum = 0.0f;
  for (int q = grp; q < L; q += 4) {
    float acc = 0.0f;
    for (int d = 0; d < dp; ++d) acc = fmaf(qb[(long)q * ldq + d], ks[tt * (dp + 1) + d], acc);
    const float M = stats[((long)bh * L + q) * 2 + 0], Ls = stats[((long)bh * L + q) * 2 + 1];
    sum += live ? fast_exp2(acc - M) / Ls : 0.0f;
  }
  part[grp * 64 + tt] = sum;
  __syncthreads();
  if (grp == 0 && t < N) I[(long)bh * N + t] = (((part[tt] + part[64 + tt]) + part[128 + tt]) + part[192 + tt]) / (float)L;
}

int launch_importance(const float *Q, long q_b, long q_h, int ldq, int dp, const float *Kp, long k_b, long k_h, int ldk,
                      const uint8_t *mask, const float *stats, float *I, int b, int h, int L, int N, hipStream_t s) {
  size_t lds = ((size_t)64 * (dp + 1) + 256) * sizeof(float);
  hipLaunchKernelGGL(importance_kernel, dim3(ceil_div(N, 64), b * h), dim3(256), lds, s, Q, q_b, q_h, ldq, dp, Kp, k_b, k_h,
                     ldk, mask, stats, I, h, L, N);
  HN_LAUNCH_CHECK("importance");
  return HN_OK;
}

}  // namespace hn
