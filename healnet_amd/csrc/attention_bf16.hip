// bf16-MFMA attention core for the shared-context (rank-D) binding: the image / volume cross-attention of
// Attention.forward (healnet/models/healnet.py:400-426) when the model runs with core_precision = bf16
// (BASELINE.json configs[2]: "3-modality ... 1 MI355X bf16").
//
// Same algorithm and partial-result format as attn_core_kernel<.., ONES> in attention.hip (split-KV flash
// attention, scores in log2 units, -m delivered through the C operand of the QK^T MFMA, softmax denominator
// accumulated by a ones row of V, lazy sum-guarded rescale); what changes is the matrix instruction and the
// operand images it wants:
//
//   v_mfma_f32_16x16x32_bf16: A lane (g, j) = row j, k = 8 g .. 8 g + 7   (one 16-byte register quad)
//                             B lane (g, j) = col j, k = 8 g .. 8 g + 7
//                             C/D lane (g, j) = col j, rows 4 g + r        (as the fp32 16x16x4)
//
//   * QK^T contracts the channel slots: one MFMA covers 32 slots (D <= 32), two MFMAs (tiles X, Y) cover a
//     32-token step.  A = context rows from the token-major image zb (Np, 32): row m of tile X is token
//     t0 + 8 (m / 4) + m % 4, of tile Y that + 4, so that lane (g, q) ends up with the scores of the eight
//     CONSECUTIVE tokens t0 + 8 g .. 8 g + 7 of its query q (X: regs 0-3, Y: regs 4-7).
//   * P V contracts those tokens: the eight probabilities, converted pairwise with v_cvt_pk_bf16_f32, ARE the
//     A operand (row q, k = 8 g + s) -- no shuffle, no LDS -- and B = eight consecutive tokens of one channel,
//     one 16-byte load from the fragment-major image zT (Np / 32, DV / 16, 4, 16, 8): a wave's load is one contiguous
//     1 KB tile.  Channel DV - 1 of zT is 1.0 on valid tokens, so
//     accumulator column DV - 1 is the softmax denominator of the SAME rounded probabilities the numerator used.
//
// NS = 2 ("bf16x3"): every operand is carried as a bf16 pair hi + lo (hi = bf16(v), lo = bf16(v - hi), 16 mantissa
// bits together) and every product as hi*hi + hi*lo + lo*hi with fp32 accumulation in the MFMA -- fp32-class results
// (operand error 2^-17, the dropped lo*lo term 2^-18) from the bf16 pipe.  QK^T: the three partial products are laid
// side by side along the contraction index (NKQ 32-slot blocks: [zh | zl] . [qh | qh] and [zh | 0] . [ql | 0] for
// D <= 16; [zh].[qh], [zl].[qh], [zh].[ql] for D <= 32).  P V: p is split in registers (cvt_pk, shift/mask, subtract,
// cvt_pk), V comes as two channel-major planes; the ones row lives in the hi plane only, so the denominator is
// sum(p_hi + p_lo).
//
// Per 32-token step and 16-query tile (NS = 1): 2 + DV/16 MFMAs of 16 cycles (fp32 path: 14 .. 26 of 32 cycles), 8 v_exp,
// 4 cvt_pk, 4 packed adds: the loop is bound by the exp / VALU rate, not by the matrix pipe.
#include "common.h"
#include <stdlib.h>

namespace hn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ float fast_exp2_b(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ bf16x8 as_bf16x8(const f32x4 &v) { return __builtin_bit_cast(bf16x8, v); }

// EXPL: the explicit K / V binding of a patch bag under core_precision = bf16 (dim_head 64): K is the token-major bf16 image the
// projection wrote ((b, N, inner): row pitch a.k_pitch bytes, head h at column 64 h), V its fragment-major image per (b, head)
// ((Np / 32, 4, 4, 16, 8): the same 1 KB tiles as zT), the query rows are the scaled projections themselves (64 slots), and the
// softmax denominator is summed on the vector pipe (V has no ones column); running reference (no score bound); the V image holds
// Np = roundup32(N) token slots per (sample, head), the pad slots zero.
template <int DTV, int NQ, int NS, bool EXPL = false>
__global__ __launch_bounds__(256) void attn_core_bf16_kernel(AttnCoreBf16Args a, int ngroups, int gy, int waves_per_block) {
  constexpr int DV = 16 * DTV;
  constexpr int NKQ = EXPL ? 2 : (NS == 1 ? 1 : (DTV == 1 ? 2 : 3));      // 32-slot blocks of the QK^T contraction
  constexpr int ZP = 32 * NKQ;                                // slots per context / query row
  const int L = a.Lq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;

  long total = (long)gridDim.x;
  long id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);      // XCD-aware remap (see attention.hip)
  const int split = (int)(id % a.nsplit);
  const int yb = (int)((id / a.nsplit) % gy);
  const int bh = (int)(id / ((long)a.nsplit * gy));
  const int qg = yb * waves_per_block + wave;
  if (qg >= ngroups) return;
  const int bi = bh / a.h;

  // ---- query fragments (B operand): lane (g, j) holds Qf[row = tile*16 + j][8 g .. 8 g + 7]
  bf16x8 qf[NQ][NKQ];
  const uint16_t *qbase = a.Qf + (long)bh * a.Lp * ZP;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int row = (qg * NQ + i) * 16 + j;
#pragma unroll
    for (int k = 0; k < NKQ; ++k) {
      f32x4 w = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (row < a.Lp) w = *(const f32x4 *)(qbase + (long)row * ZP + 32 * k + 8 * g);
      qf[i][k] = as_bf16x8(w);
    }
  }

  f32x4 O[NQ][DTV];
  f32x4 negm[NQ];
  float m[NQ];
  float lsum[NQ];      // EXPL: this lane's share (8 of every 32 tokens) of its query row's denominator
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    negm[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m[i] = 0.0f;
    lsum[i] = 0.0f;
#pragma unroll
    for (int d = 0; d < DTV; ++d) O[i][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  bool unset = true;
  // fixed Cauchy-Schwarz reference (see attention.hip); a scalar INTEGER: a bool carried into the loop costs VALU copies there
  const int unbounded = a.bound != nullptr ? __builtin_amdgcn_readfirstlane(*a.bound_flag) : 1;
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
  const uint8_t *mrow = a.mask ? a.mask + (long)bi * a.N : nullptr;

  // (EXPL: the descriptor ends with this head's last row, so token rows past N read 0 whatever the other heads hold there)
  const int kpitch = EXPL ? a.k_pitch : ZP * 2;      // bytes per token row of the QK^T image
  const int hi = bh - bi * a.h;
  const i32x4 krs = EXPL ? make_rsrc(a.zb + ((long)bi * a.N * kpitch) / 2 + hi * 64, (unsigned)((long)a.N * kpitch - hi * 128))
                         : make_rsrc(a.zb + (long)bi * a.Np * ZP, (unsigned)((long)a.Np * ZP * 2));
  const i32x4 vrs = EXPL ? make_rsrc(a.zT + (long)bh * DV * a.Np, (unsigned)((long)DV * a.Np * 2))
                         : make_rsrc(a.zT + (long)bi * NS * DV * a.Np, (unsigned)((long)NS * DV * a.Np * 2));
  const int koff = (8 * (j >> 2) + (j & 3)) * kpitch + 16 * g;
  int voff[NS][DTV];
#pragma unroll
  for (int p = 0; p < NS; ++p)
#pragma unroll
    for (int d = 0; d < DTV; ++d) voff[p][d] = (p * DV * a.Np + (d * 64 + g * 16 + j) * 8) * 2;      // fragment-major tiles (encode.hip)

  auto load_kv = [&](int t0, f32x4 (&kf)[2][NKQ], f32x4 (&vf)[NS][DTV]) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int k = 0; k < NKQ; ++k) kf[x][k] = hn_buffer_load_x4(krs, koff + x * 4 * kpitch + 64 * k, t0 * kpitch, 0);
#pragma unroll
    for (int p = 0; p < NS; ++p)
#pragma unroll
      for (int d = 0; d < DTV; ++d) vf[p][d] = hn_buffer_load_x4(vrs, voff[p][d], t0 * (DTV * 32), 0);      // 32 tokens = DTV tiles of 1 KB
  };

  // one 32-token step on (kf, vf); prefetches the following step into (kn, vn)
  auto step = [&](int t0, f32x4 (&kf)[2][NKQ], f32x4 (&vf)[NS][DTV], f32x4 (&kn)[2][NKQ], f32x4 (&vn)[NS][DTV]) {
    load_kv(t0 + 32, kn, vn);     // unconditional (see attention.hip: behind a branch the compiler waits vmcnt(0) for it at once); rows past the end read 0

    f32x4 S[NQ][2];
#pragma unroll
    for (int k = 0; k < NKQ; ++k)
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < NQ; ++i)
          S[i][x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kf[x][k]), qf[i][k], k == 0 ? negm[i] : S[i][x], 0, 0, 0);

    // ---- mask / ragged tail: lane (g, j) holds tokens t0 + 8 g + 4 x + r
    bool any_live = true;
    if (mrow != nullptr || t0 + 32 > t_end) {
      bool live = false;
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tok = t0 + 8 * g + 4 * x + r;
          bool ok = tok < t_end;
          if (ok && mrow) ok = mrow[tok] != 0;
          live |= ok;
          if (!ok) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) S[i][x][r] = -__builtin_inff();
          }
        }
      any_live = __any(live);
    }

    float ps0 = 0.0f, ps1 = 0.0f;
    f32x4 P[NQ][2];
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
      for (int x = 0; x < 2; ++x) {
#pragma unroll
        for (int r = 0; r < 4; ++r) P[i][x][r] = fast_exp2_b(S[i][x][r]);
        if (unbounded != 0) {
          ps0 += P[i][x][0] + P[i][x][2];
          ps1 += P[i][x][1] + P[i][x][3];
        }
      }
    bool rescale = false;
    if (unbounded != 0) {
      const bool need = !(ps0 + ps1 <= 512.0f) || unset;
      rescale = __any(need) && any_live;
    }
    if (rescale) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        float tm = fmaxf(fmaxf(fmaxf(S[i][0][0], S[i][0][1]), fmaxf(S[i][0][2], S[i][0][3])),
                         fmaxf(fmaxf(S[i][1][0], S[i][1][1]), fmaxf(S[i][1][2], S[i][1][3])));
        tm = fmaxf(tm, __shfl_xor(tm, 16));
        tm = fmaxf(tm, __shfl_xor(tm, 32));
        float delta = unset ? tm : fmaxf(tm, 0.0f);
        if (!(delta > -3.0e38f)) delta = 0.0f;          // row saw only -inf scores: keep the reference
        const float alpha = unset ? 1.0f : fast_exp2_b(-delta);
        m[i] += delta;
        if (EXPL) lsum[i] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ar = __shfl(alpha, 4 * g + r);    // accumulator reg r of lane (g, d) belongs to query row 4 g + r
#pragma unroll
          for (int d = 0; d < DTV; ++d) O[i][d][r] *= ar;
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int r = 0; r < 4; ++r) P[i][x][r] = fast_exp2_b(S[i][x][r] - delta);
        negm[i] = (f32x4){-m[i], -m[i], -m[i], -m[i]};
      }
      unset = false;
    }

    if (EXPL) {
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        lsum[i] += ((P[i][0][0] + P[i][0][1]) + (P[i][0][2] + P[i][0][3])) + ((P[i][1][0] + P[i][1][1]) + (P[i][1][2] + P[i][1][3]));
    }
    // ---- O += P V: A = the eight probabilities of this lane's query, B = eight consecutive tokens of a channel
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 pk;
      pk.x = cvt_pk_bf16(P[i][0][0], P[i][0][1]);
      pk.y = cvt_pk_bf16(P[i][0][2], P[i][0][3]);
      pk.z = cvt_pk_bf16(P[i][1][0], P[i][1][1]);
      pk.w = cvt_pk_bf16(P[i][1][2], P[i][1][3]);
      const bf16x8 pa = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
      for (int d = 0; d < DTV; ++d)
        O[i][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, as_bf16x8(vf[0][d]), O[i][d], 0, 0, 0);
      if (NS == 2) {
        // residual plane: p_lo = bf16(p - p_hi); unpack p_hi from the packed pairs (low half << 16, high half masked)
        u32x4 pl;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const unsigned u = pk[w];
          const float h0 = __uint_as_float(u << 16), h1 = __uint_as_float(u & 0xffff0000u);
          const int x = w >> 1, r = (w & 1) * 2;
          pl[w] = cvt_pk_bf16(P[i][x][r] - h0, P[i][x][r + 1] - h1);
        }
        const bf16x8 pb = __builtin_bit_cast(bf16x8, pl);
#pragma unroll
        for (int d = 0; d < DTV; ++d) {
          O[i][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, as_bf16x8(vf[1][d]), O[i][d], 0, 0, 0);
          O[i][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pb, as_bf16x8(vf[0][d]), O[i][d], 0, 0, 0);
        }
      }
    }
  };

  f32x4 kA[2][NKQ], kB[2][NKQ], vA[NS][DTV], vB[NS][DTV];
  if (!EXPL && NS == 1 && bounded && mrow == nullptr && !a.no_pipeline) {
    // ---- software-pipelined loop for the bounded reference without a key mask (the inference forward of BASELINE configs[2]).
    // tools/ubench/mfma_valu_overlap.hip: on gfx950 a block of v_mfma_f32_16x16x32_bf16 and a block of v_exp_f32 of the waves
    // of one SIMD do NOT overlap (both = 0.92 x the sum), but with the exponentials PINNED between the MFMAs of the same wave
    // a quarter of the vector time hides (4 waves per SIMD: 495 -> 396 ns per 8 MFMAs + 16 exp + 16 mul; the fp32 16x16x4
    // MFMA shows no such effect: 781 -> 816).  So the QK^T MFMAs of step t + 1 are issued one by one with the four
    // exponentials of a score quad of step t behind each; the probabilities overwrite their scores, the next step's scores
    // land in the other register set (same register count as the S / P pair of the general loop).  No tail masking is needed:
    // splits end on multiples of 32 tokens and the rows of zT past N (ones row included) are zero, so tokens past the end add
    // nothing to O or to the denominator.
    auto load_k = [&](int t0, f32x4 (&kf)[2][NKQ]) {
#pragma unroll
      for (int x = 0; x < 2; ++x) kf[x][0] = hn_buffer_load_x4(krs, koff + x * 4 * (ZP * 2), t0 * (ZP * 2), 0);
    };
    auto load_v = [&](int t0, f32x4 (&vf)[NS][DTV]) {
#pragma unroll
      for (int d = 0; d < DTV; ++d) vf[0][d] = hn_buffer_load_x4(vrs, voff[0][d], t0 * (DTV * 32), 0);
    };
    f32x4 S0[NQ][2], S1[NQ][2];
    auto pstep = [&](f32x4 (&Sc)[NQ][2], f32x4 (&Sn)[NQ][2], const f32x4 (&kn)[2][NKQ], const f32x4 (&vf)[NS][DTV]) {
#define BF_SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
      for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          Sn[i][x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kn[x][0]), qf[i][0], negm[i], 0, 0, 0); BF_SB;
#pragma unroll
          for (int r = 0; r < 4; ++r) Sc[i][x][r] = fast_exp2_b(Sc[i][x][r]);
          BF_SB;
        }
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 pk;
        pk.x = cvt_pk_bf16(Sc[i][0][0], Sc[i][0][1]);
        pk.y = cvt_pk_bf16(Sc[i][0][2], Sc[i][0][3]);
        pk.z = cvt_pk_bf16(Sc[i][1][0], Sc[i][1][1]);
        pk.w = cvt_pk_bf16(Sc[i][1][2], Sc[i][1][3]);
        const bf16x8 pa = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
        for (int d = 0; d < DTV; ++d)
          O[i][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, as_bf16x8(vf[0][d]), O[i][d], 0, 0, 0);
      }
#undef BF_SB
    };
    if (t_begin < t_end) {
      load_k(t_begin, kA);
      load_v(t_begin, vA);
      load_k(t_begin + 32, kB);
#pragma unroll
      for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int x = 0; x < 2; ++x)
          S0[i][x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(kA[x][0]), qf[i][0], negm[i], 0, 0, 0);
    }
    for (int t0 = t_begin; t0 < t_end; t0 += 64) {
      load_k(t0 + 64, kA);                    // unconditional prefetches: rows past the end read 0
      load_v(t0 + 32, vB);
      pstep(S0, S1, kB, vA);
      if (t0 + 32 < t_end) {
        load_k(t0 + 96, kB);
        load_v(t0 + 64, vA);
        pstep(S1, S0, kA, vB);
      }
    }
  } else {
    if (t_begin < t_end) load_kv(t_begin, kA, vA);
    for (int t0 = t_begin; t0 < t_end; t0 += 64) {
      step(t0, kA, vA, kB, vB);
      if (t0 + 32 < t_end) step(t0 + 32, kB, vB, kA, vA);
    }
  }

  // ---- partial (O, m, l) of this split; l sits in accumulator column DV-1
  const long prow = ((long)bh * a.nsplit + split) * a.Lp;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int tile = qg * NQ + i;
    if (tile * 16 < a.Lp) {
#pragma unroll
      for (int d = 0; d < DTV; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          a.Opart[(prow + tile * 16 + 4 * g + r) * DV + 16 * d + j] = O[i][d][r];
      if (g == 0) a.Mpart[prow + tile * 16 + j] = m[i];
      if (EXPL) {
        float l = lsum[i];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        if (g == 0) a.Lpart[prow + tile * 16 + j] = l;
      } else if (j == 15) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.Lpart[prow + tile * 16 + 4 * g + r] = O[i][DTV - 1][r];
      }
    }
  }
  (void)L;
}

int launch_attn_core_bf16(const AttnCoreBf16Args &a, hipStream_t s) {
  if (a.expl) {
    HN_REQUIRE(a.DV == 64 && a.ns == 1 && a.Np % 32 == 0 && a.Np >= a.N && a.Np < a.N + 32 && a.k_pitch >= 128 && a.k_pitch % 16 == 0 && !a.bound,
               HN_E_UNSUPPORTED, "attn_core_bf16 (explicit binding): DV=%d ns=%d N=%d Np=%d k_pitch=%d", a.DV, a.ns, a.N, a.Np, a.k_pitch);
    HN_REQUIRE(a.Lp % 16 == 0 && a.chunk % 32 == 0 && a.nsplit >= 1 && (long)a.N * a.k_pitch < (1L << 31), HN_E_SHAPE,
               "attn_core_bf16 (explicit binding): Lp=%d chunk=%d N=%d", a.Lp, a.chunk, a.N);
    // query tiles per wave: 2 (180 VGPRs, two waves per SIMD; the 4 query groups of a (sample, head) each stream its K / V tiles) or
    // 4 (279 VGPRs, one wave per SIMD, half the K / V traffic): HN_BF16_EXPL_NQ, development knob
    static const int nqe = tuning_env("HN_BF16_EXPL_NQ") ? atoi(tuning_env("HN_BF16_EXPL_NQ")) : 2;
    const int NQE = nqe == 4 ? 4 : 2;
    const int ngroups = ceil_div(a.Lp / 16, NQE);
    const int wpb = ngroups < 4 ? ngroups : 4;
    const int gy = ceil_div(ngroups, wpb);
    const long blocks = (long)a.nsplit * gy * a.b * a.h;
    HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "attn_core_bf16: grid too large");
    if (NQE == 4) hipLaunchKernelGGL((attn_core_bf16_kernel<4, 4, 1, true>), dim3((unsigned)blocks), dim3(64 * wpb), 0, s, a, ngroups, gy, wpb);
    else hipLaunchKernelGGL((attn_core_bf16_kernel<4, 2, 1, true>), dim3((unsigned)blocks), dim3(64 * wpb), 0, s, a, ngroups, gy, wpb);
    HN_LAUNCH_CHECK("attn_core_bf16(explicit)");
    return HN_OK;
  }
  HN_REQUIRE(a.DV == 16 || a.DV == 32, HN_E_UNSUPPORTED, "attn_core_bf16: DV=%d", a.DV);
  HN_REQUIRE(a.ns == 1 || a.ns == 2, HN_E_UNSUPPORTED, "attn_core_bf16: ns=%d", a.ns);
  HN_REQUIRE(a.Lp % 16 == 0 && a.chunk % 32 == 0 && a.nsplit >= 1 && a.Np % 32 == 0 && a.Np >= a.N, HN_E_SHAPE,
             "attn_core_bf16: Lp=%d chunk=%d Np=%d", a.Lp, a.chunk, a.Np);
  HN_REQUIRE((long)a.Np * bf16_row_slots(a.DV, a.ns) * 2 < (1L << 31) && (long)a.ns * a.DV * a.Np * 2 < (1L << 31), HN_E_UNSUPPORTED,
             "attn_core_bf16: one sample's context must span < 2 GiB (N=%d)", a.N);
  constexpr int NQ = 4;
  const int ngroups = ceil_div(a.Lp / 16, NQ);
  const int wpb = ngroups < 4 ? ngroups : 4;
  const int gy = ceil_div(ngroups, wpb);
  const long blocks = (long)a.nsplit * gy * a.b * a.h;
  HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "attn_core_bf16: grid too large");
  dim3 grid((unsigned)blocks), block(64 * wpb);
  static const bool no_pipeline = tuning_env("HN_BF16_NO_PIPELINE") != nullptr;
  AttnCoreBf16Args ap = a;
  ap.no_pipeline = no_pipeline ? 1 : 0;
#define HN_CORE16(DT_, NS_) hipLaunchKernelGGL((attn_core_bf16_kernel<DT_, NQ, NS_>), grid, block, 0, s, ap, ngroups, gy, wpb)
  if (a.DV == 16 && a.ns == 1) HN_CORE16(1, 1);
  else if (a.DV == 16) HN_CORE16(1, 2);
  else if (a.ns == 1) HN_CORE16(2, 1);
  else HN_CORE16(2, 2);
#undef HN_CORE16
  HN_LAUNCH_CHECK("attn_core_bf16");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// query side: Qf[b,h,q,c] = bf16( cscale * gamma[c] * sum_e Q[b,q,h*dh+e] * W_k[h*dh+e, c] ), 32 slots per row
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t f2bf(float f) {       // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((unsigned)h << 16); }

// q-side slot image (see the kernel header): ns == 1: [qh(32)];  ns == 2, DV == 16: [qh(16) qh(16)] [ql(16) 0];
// ns == 2, DV == 32: [qh(32)] [qh(32)] [ql(32)]
__global__ __launch_bounds__(256) void qfold_bf16_kernel(const float *__restrict__ Q, int ldq_row, const float *__restrict__ w_k,
                                                         int D, const float *__restrict__ gamma, float cscale,
                                                         uint16_t *__restrict__ Qf, int h, int L, int Lp, int dh, int DV, int ns,
                                                         float *__restrict__ bound, int *__restrict__ bound_flag) {
  extern __shared__ float wk[];  // [dh][32] + [Lp] row sums of squares
  const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
  for (int idx = threadIdx.x; idx < dh * 32; idx += blockDim.x) {
    const int e = idx >> 5, d = idx & 31;
    wk[idx] = d < D ? w_k[(long)(hi * dh + e) * D + d] * (gamma ? gamma[d] : 1.0f) * cscale : 0.0f;
  }
  __syncthreads();
  const int zp = bf16_row_slots(DV, ns);
  uint16_t *dst = Qf + (long)bh * Lp * zp;
  float *ssq = wk + dh * 32;
  for (int q = threadIdx.x; q < Lp; q += blockDim.x) ssq[q] = 0.0f;
  __syncthreads();
  for (int idx = threadIdx.x; idx < Lp * 32; idx += blockDim.x) {
    const int q = idx >> 5, d = idx & 31;
    float acc = 0.0f;
    if (q < L && d < D) {
      const float *qr = Q + ((long)bi * L + q) * ldq_row + hi * dh;
      for (int e = 0; e < dh; ++e) acc = fmaf(qr[e], wk[e * 32 + d], acc);
    }
    if (bound != nullptr) {      // row sum of squares: the 32 lanes of a row are one half-wave
      float sq = acc * acc;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (d == 0) ssq[q] = sq;
    }
    const uint16_t hi16 = f2bf(acc);
    uint16_t *row = dst + (long)q * zp;
    if (ns == 1) {
      row[d] = hi16;
    } else {
      const uint16_t lo16 = f2bf(acc - bf2f(hi16));
      if (DV == 16) {
        if (d < 16) { row[d] = hi16; row[16 + d] = hi16; row[32 + d] = lo16; row[48 + d] = 0; }
      } else {
        row[d] = hi16; row[32 + d] = hi16; row[64 + d] = lo16;
      }
    }
  }
  if (bound != nullptr) {        // Cauchy-Schwarz score bound (see qfold_kernel); 1 % slack covers the bf16 rounding of q and z
    __syncthreads();
    for (int q = threadIdx.x; q < Lp; q += blockDim.x) {
      const float bq = sqrtf(ssq[q] * (float)D) * 1.01f + 1e-6f;
      bound[(long)bh * Lp + q] = bq;
      if (bq > 60.0f) atomicOr(bound_flag, 1);
    }
  }
}

int launch_qfold_bf16(const float *Q, int ldq_row, const float *w_k, int D, const float *gamma, float cscale, uint16_t *Qf,
                      int b, int h, int L, int Lp, int dh, int DV, int ns, hipStream_t s, float *bound, int *bound_flag) {
  HN_REQUIRE(D <= DV - 1 && (ns == 1 || ns == 2), HN_E_SHAPE, "qfold_bf16: D=%d DV=%d ns=%d", D, DV, ns);
  // one operand plane: the fold runs on the matrix cores (33 -> ~9 us per call at cfg3: 6 calls per forward)
  if (ns == 1 && launch_qfold_mfma_bf16(Q, ldq_row, w_k, D, gamma, cscale, Qf, b, h, L, Lp, dh, s, bound, bound_flag)) {
    HN_LAUNCH_CHECK("qfold_mfma_bf16");
    return HN_OK;
  }
  hipLaunchKernelGGL(qfold_bf16_kernel, dim3(b * h), dim3(256), ((size_t)dh * 32 + Lp) * sizeof(float), s, Q, ldq_row, w_k, D, gamma,
                     cscale, Qf, h, L, Lp, dh, DV, ns, bound, bound_flag);
  HN_LAUNCH_CHECK("qfold_bf16");
  return HN_OK;
}

}  // namespace hn
