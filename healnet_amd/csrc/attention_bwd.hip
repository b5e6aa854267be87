// Backward of the split-KV attention core (autograd of healnet/models/healnet.py Attention.forward :409-424), exact
// fp32 on the matrix cores, recompute-based: the forward keeps only the per-row softmax statistics (max, sum) and
// the normalised output; P is re-formed tile by tile from Q K^T and those statistics.
//
// With  P = softmax(s),  O = P V,  D_q = sum_d dO[q,d] O[q,d]  (flash-attention identity):
//     dP = dO V^T          dS = P * (dP - D)          dQ = dS K          dK = dS^T Q          dV = P^T dO
// Scores are kept in log2 units (S = Qs K^T with Qs = 2 dh^-1/2 log2(e) q), the kernels accumulate the plain sums
// sum dS K / sum dS^T Qs / sum P^T dO and the callers apply the constant factors.
//
// Two kernels, mirroring the two kinds of outputs:
//   attn_bwd_dq_kernel   query-parallel, token range split across waves exactly like the forward core; output
//                        partial dQ (reduced over splits by dq_reduce_kernel).  This is all the rank-D binding
//                        needs (K = V = normalised context z, which has no gradient).
//   attn_bwd_dkv_kernel  token-parallel (explicit K/V only): a wave owns one 16-token tile, walks all query tiles
//                        and keeps dK / dV of its tokens in registers: no cross-wave reduction, deterministic.
#include "common.h"
#include <stdlib.h>
#pragma clang diagnostic ignored "-Winline-asm"

namespace hn {

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ------------------------------------------------------------------------------------------------
// dQ.  Register layout as in the forward core: S^T = K Q^T (A = K tile, B = Q tile) leaves lane (g, j) with the
// scores of query row j for tokens 4 g + r;  dP^T = V dO^T has the same shape with the dO fragment in the place of
// the Q fragment;  dQ += dS K uses dS straight from registers as the A operand and K rows 4 g + r as B.
// ------------------------------------------------------------------------------------------------
// KS > 0: packed context (common.h packed_slot) -- the channel contractions of S and dP run KS steps instead of 4 DT, the last
// 16-column block only its first KS - 4 (DT - 1) k-steps (the skipped ones hold unused columns and the ones column).
// (the body is a device function so that attn_bwd_self_pair_kernel below can run it beside the dK/dV body in one launch; `id` of
// `total` = the workgroup's index in the dQ part of the grid)
template <int DT, int NQ, bool SHARED_KV, bool DROP = false, int KS = 0>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnBwdArgs &a, int ngroups, int gy, int waves_per_block, long id, long total) {
  constexpr int DP = 16 * DT;
  constexpr int LAST = KS > 0 ? KS - 4 * (DT - 1) : 4;     // k-steps of the last 16-column block
  const int L = a.Lq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int split = (int)(id % a.nsplit);
  const int yb = (int)((id / a.nsplit) % gy);
  const int bh = (int)(id / ((long)a.nsplit * gy));
  const int qg = yb * waves_per_block + wave;
  if (qg >= ngroups) return;
  const int bi = bh / a.h, hi = bh % a.h;

  const i32x4 rsQ = make_rsrc(a.Q + (long)bi * a.q_b + (long)hi * a.q_h, rsrc_bytes(L, a.ldq, DP));
  const i32x4 rsG = make_rsrc(a.dO + (long)bi * a.do_b + (long)hi * a.do_h, rsrc_bytes(L, a.lddo, DP));
  float4 qf[NQ][DT], gf[NQ][DT];
  float mrow[NQ], invl[NQ], drow[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int row = (qg * NQ + i) * 16 + j;
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      qf[i][s] = buf4(rsQ, (row * a.ldq + 16 * s + 4 * g) * 4);
      gf[i][s] = buf4(rsG, (row * a.lddo + 16 * s + 4 * g) * 4);
    }
    const int rc = min(row, L - 1);
    mrow[i] = a.stats[((long)bh * L + rc) * 2 + 0];
    invl[i] = 1.0f / a.stats[((long)bh * L + rc) * 2 + 1];
    drow[i] = a.delta[(long)bh * L + rc];
  }

  f32x4 dQ[NQ][DT], negm[NQ], negd[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    negm[i] = (f32x4){-mrow[i], -mrow[i], -mrow[i], -mrow[i]};
    negd[i] = (f32x4){-drow[i], -drow[i], -drow[i], -drow[i]};
    if (DROP && KS > 0) {
      // packed context under dropout: the dP chain skips the k-step of the ones column, so the row-sum channel's gradient
      // ds = dO'[row j][DP - 1] (held by lane (3, j)) starts the chain instead -- dP = dO' V^T + ds, thinned afterwards
      const float ds = a.drop_rowsum ? __shfl(gf[i][DT - 1].w, 48 + j) : 0.0f;
      negd[i] = (f32x4){ds, ds, ds, ds};
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) dQ[i][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const int t_begin = split * a.chunk;
  const int t_end = min(a.N, t_begin + a.chunk);
  const float *kbase = a.Kp + (long)bi * a.k_b + (long)hi * a.k_h;
  const float *vbase = a.Vp + (long)bi * a.v_b + (long)hi * a.v_h;
  const uint8_t *mask = a.mask ? a.mask + (long)bi * a.N : nullptr;
  const i32x4 krs = make_rsrc(kbase, rsrc_bytes(a.N, a.ldk, DP)), vrs = make_rsrc(vbase, rsrc_bytes(a.N, a.ldv, DP));
  const int koff = (j * a.ldk + 4 * g) * 4, vkoff = (j * a.ldv + 4 * g) * 4;
  int kroff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) kroff[r] = ((4 * g + r) * a.ldk + j) * 4;

  // fragments of one 16-token tile: K (and V) rows j as A operands of the S / dP chains, K rows 4 g + r as the B operand of dQ.
  // Rows past the context read 0 through the descriptors, so a request needs no guard.
  // TWO fragment sets: the next tile is requested -- unconditionally: a guarded request makes the compiler wait vmcnt(0) for it at
  // once -- before the current one is multiplied.  Measured per launch: dp = 16 (cfg2 image, N = 50 176) 1143 -> 1073 us;
  // dp = 64 (cfg4 patch bag, N = 4096: 48 more VGPRs, three -> two resident waves) 184 -> 132 us.
  constexpr int NSET = 2;
  float4 kfs[NSET][DT], vfs[NSET][DT];
  float krs_[NSET][DT][4];
  auto load_frags = [&](int t, float4 (&kf)[DT], float4 (&vf)[DT], float (&kr)[DT][4]) {
    const int ks = t * a.ldk * 4, vs = t * a.ldv * 4;
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      kf[s] = buf4s(krs, koff + 64 * s, ks);
      if (!SHARED_KV) vf[s] = buf4s(vrs, vkoff + 64 * s, vs);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < DT; ++d) kr[d][r] = hn_buffer_load_x1(krs, kroff[r] + 64 * d, ks, 0);
  };
  if (NSET == 2) load_frags(t_begin, kfs[0], vfs[0], krs_[0]);
  for (int tb = t_begin; tb < t_end; tb += 16 * NSET) {
#pragma unroll
  for (int ps = 0; ps < NSET; ++ps) {
    const int t0 = tb + 16 * ps;
    if (t0 >= t_end) break;
    if (NSET == 2) load_frags(t0 + 16, kfs[ps ^ 1], vfs[ps ^ 1], krs_[ps ^ 1]);
    else load_frags(t0, kfs[0], vfs[0], krs_[0]);
    float4 (&kf)[DT] = kfs[ps];
    float4 (&vf)[DT] = vfs[ps];
    float (&kr)[DT][4] = krs_[ps];

    // S and dP start from -m and -D (persistent register quads) as the C operands of their MFMA chains, so the chains deliver
    // s - m and dP - D directly; 1/l is applied once to the finished dQ rows.  (With dropout dP must be thinned BEFORE D is
    // subtracted, so that variant starts dP from 0.)
    f32x4 S[NQ], dP[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) { S[i] = negm[i]; dP[i] = (DROP && KS == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : negd[i]; }
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      float4 vv = SHARED_KV ? kf[s] : vf[s];
      if (DROP && KS == 0 && SHARED_KV && s == DT - 1 && a.drop_rowsum && g == 3) vv.w = 1.0f;      // the values' ones column dp-1 (row-sum channel)
      const int steps = s == DT - 1 ? LAST : 4;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        S[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].x, qf[i][s].x, S[i], 0, 0, 0);
        dP[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.x, gf[i][s].x, dP[i], 0, 0, 0);
      }
      if (steps > 1) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          S[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].y, qf[i][s].y, S[i], 0, 0, 0);
          dP[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.y, gf[i][s].y, dP[i], 0, 0, 0);
        }
      }
      if (steps > 2) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          S[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].z, qf[i][s].z, S[i], 0, 0, 0);
          dP[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.z, gf[i][s].z, dP[i], 0, 0, 0);
        }
      }
      if (steps > 3) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          S[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].w, qf[i][s].w, S[i], 0, 0, 0);
          dP[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.w, gf[i][s].w, dP[i], 0, 0, 0);
        }
      }
    }
    if (DROP) {   // O = (P * d) V with the forward's keep/scale factors d:  dP <- d * (dO V^T) - D
      const uint32_t row0 = (uint32_t)(bh * L + qg * NQ * 16 + j), quad = (uint32_t)(t0 + 4 * g) >> 2;
      if (NQ % 4 == 0 && ((bh * L) & 63) == 0) {      // rows 16 apart share a generator call (common.h; as in the forward core)
#pragma unroll
        for (int i = 0; i + 3 < NQ; i += 4) {
          bool keep[4][4];
          drop_rows4(a.drop, quad, row0 + 16 * i, keep);
          const float sc = a.drop.scale;
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) dP[i + k][r] = keep[k][r] ? fmaf(dP[i + k][r], sc, -drow[i + k]) : -drow[i + k];
        }
      } else if (NQ % 2 == 0 && ((bh * L) & 31) == 0) {
#pragma unroll
        for (int i = 0; i + 1 < NQ; i += 2) {
          bool lo[4], hi[4];
          drop_pair(a.drop, quad, row0 + 16 * i, lo, hi);
          const float sc = a.drop.scale;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dP[i][r] = lo[r] ? fmaf(dP[i][r], sc, -drow[i]) : -drow[i];
            dP[i + 1][r] = hi[r] ? fmaf(dP[i + 1][r], sc, -drow[i + 1]) : -drow[i + 1];
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          bool keep[4];
          drop_quad_attn(a.drop, quad, row0 + 16 * i, keep);
          const float sc = a.drop.scale;
#pragma unroll
          for (int r = 0; r < 4; ++r) dP[i][r] = keep[r] ? fmaf(dP[i][r], sc, -drow[i]) : -drow[i];
        }
      }
    }
    // token validity of lane (g, j): tokens t0 + 4 g + r -- only the ragged tail / masked launches pay for it
    if (mask != nullptr || t0 + 16 > t_end) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tok = t0 + 4 * g + r;
        bool ok = tok < t_end;
        if (ok && mask) ok = mask[tok] != 0;
        if (!ok) {
#pragma unroll
          for (int i = 0; i < NQ; ++i) S[i][r] = -__builtin_inff();
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {          // l * dS  (2^-inf = 0 kills invalid tokens); the product as a vector op: two v_pk_mul_f32
      const f32x4 e = {fexp2(S[i][0]), fexp2(S[i][1]), fexp2(S[i][2]), fexp2(S[i][3])};
      S[i] = e * dP[i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < NQ; ++i)
          dQ[i][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(S[i][r], kr[d][r], dQ[i][d], 0, 0, 0);
  }
  }

  const long prow = ((long)bh * a.nsplit + split) * a.Lp;
  if (a.dQfinal != nullptr) {      // one split: the finished rows, as dq_reduce would write them (wave-uniform)
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int tile = qg * NQ + i;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = tile * 16 + 4 * g + r, c = 16 * d + j;
          const float v = dQ[i][d][r] * __shfl(invl[i], 4 * g + r);
          if (q < L && c < a.dq_width) a.dQfinal[((long)bi * L + q) * a.dq_ld + (long)hi * a.dq_pitch + c] = v * a.dq_scale;
        }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int tile = qg * NQ + i;
    if (tile * 16 < a.Lp) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r)     // accumulator reg r of lane (g, d) belongs to query row 4 g + r, whose 1/l lives in lane 4 g + r
          a.dQpart[(prow + tile * 16 + 4 * g + r) * DP + 16 * d + j] = dQ[i][d][r] * __shfl(invl[i], 4 * g + r);
    }
  }
}

template <int DT, int NQ, bool SHARED_KV, bool DROP = false, int KS = 0>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs a, int ngroups, int gy, int waves_per_block) {
  attn_bwd_dq_body<DT, NQ, SHARED_KV, DROP, KS>(a, ngroups, gy, waves_per_block, (long)blockIdx.x, (long)gridDim.x);
}

static int bwd_nq(int dt) { return dt == 1 ? 4 : (dt == 2 ? 2 : (dt == 4 ? 2 : 1)); }

int launch_attn_bwd_dq(const AttnBwdArgs &a, hipStream_t s) {
  HN_REQUIRE(a.dp == 16 || a.dp == 32 || a.dp == 64 || a.dp == 128, HN_E_UNSUPPORTED, "attn_bwd_dq: dp=%d", a.dp);
  HN_REQUIRE(a.dQfinal == nullptr || (a.nsplit == 1 && !attn_bwd_dq_lds_eligible(a)), HN_E_SHAPE, "attn_bwd_dq: direct rows need one split (nsplit=%d)", a.nsplit);
  if (attn_bwd_dq_lds_eligible(a)) return launch_attn_bwd_dq_lds(a, s);
  const int dt = a.dp / 16, nq = bwd_nq(dt);
  const int ngroups = ceil_div(a.Lp / 16, nq);
  const int wpb = ngroups < 4 ? ngroups : 4;
  const int gy = ceil_div(ngroups, wpb);
  const long blocks = (long)a.nsplit * gy * a.b * a.h;
  HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "attn_bwd_dq: grid too large");
  dim3 grid((unsigned)blocks), block(64 * wpb);
  const bool shared = a.Kp == a.Vp;
  const bool drop = a.drop.thr != 0;
#define HN_DQ(DT_, NQ_)                                                                                                          \
  if (shared && drop) hipLaunchKernelGGL((attn_bwd_dq_kernel<DT_, NQ_, true, true>), grid, block, 0, s, a, ngroups, gy, wpb);     \
  else if (shared) hipLaunchKernelGGL((attn_bwd_dq_kernel<DT_, NQ_, true, false>), grid, block, 0, s, a, ngroups, gy, wpb);      \
  else if (drop) hipLaunchKernelGGL((attn_bwd_dq_kernel<DT_, NQ_, false, true>), grid, block, 0, s, a, ngroups, gy, wpb);        \
  else hipLaunchKernelGGL((attn_bwd_dq_kernel<DT_, NQ_, false, false>), grid, block, 0, s, a, ngroups, gy, wpb);
#define HN_DQ_PACKED(DT_, NQ_, KS_) hipLaunchKernelGGL((attn_bwd_dq_kernel<DT_, NQ_, true, false, KS_>), grid, block, 0, s, a, ngroups, gy, wpb)
  if (a.qk_steps > 0) {     // packed shared context (rank-D binding; round 4: under dropout as well)
    HN_REQUIRE(shared && (dt == 1 || dt == 2) && a.qk_steps > 4 * (dt - 1) && a.qk_steps < 4 * dt, HN_E_SHAPE,
               "attn_bwd_dq: qk_steps=%d dp=%d", a.qk_steps, a.dp);
#define HN_DQ_PACKED_DROP(DT_, NQ_, KS_) hipLaunchKernelGGL((attn_bwd_dq_kernel<DT_, NQ_, true, true, KS_>), grid, block, 0, s, a, ngroups, gy, wpb)
    if (drop) {
      switch (a.qk_steps) {
        case 1: HN_DQ_PACKED_DROP(1, 4, 1); break;
        case 2: HN_DQ_PACKED_DROP(1, 4, 2); break;
        case 3: HN_DQ_PACKED_DROP(1, 4, 3); break;
        case 5: HN_DQ_PACKED_DROP(2, 2, 5); break;
        case 6: HN_DQ_PACKED_DROP(2, 2, 6); break;
        default: HN_DQ_PACKED_DROP(2, 2, 7); break;
      }
      HN_LAUNCH_CHECK("attn_bwd_dq(packed, dropout)");
      return HN_OK;
    }
#undef HN_DQ_PACKED_DROP
    switch (a.qk_steps) {
      case 1: HN_DQ_PACKED(1, 4, 1); break;
      case 2: HN_DQ_PACKED(1, 4, 2); break;
      case 3: HN_DQ_PACKED(1, 4, 3); break;
      case 5: HN_DQ_PACKED(2, 2, 5); break;
      case 6: HN_DQ_PACKED(2, 2, 6); break;
      default: HN_DQ_PACKED(2, 2, 7); break;
    }
    HN_LAUNCH_CHECK("attn_bwd_dq(packed)");
    return HN_OK;
  }
  switch (dt) {
    case 1: HN_DQ(1, 4) break;
    case 2: HN_DQ(2, 2) break;
    case 4: HN_DQ(4, 2) break;
    default: HN_DQ(8, 1) break;
  }
#undef HN_DQ_PACKED
#undef HN_DQ
  HN_LAUNCH_CHECK("attn_bwd_dq");
  return HN_OK;
}

// sum of the split partials in fixed order, scaled, written as (b*L rows, h heads, `width` of dp columns) with row pitch ld_out
__global__ __launch_bounds__(256) void dq_reduce_kernel(const float *__restrict__ part, int nsplit, int h, int L, int Lp, int dp,
                                                        int width, float scale, float *__restrict__ out, int ld_out, int head_pitch,
                                                        long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % width);
    const long t = i / width;
    const int q = (int)(t % L);
    const long bh = t / L;
    float acc = 0.0f;
    for (int s = 0; s < nsplit; ++s) acc += part[((bh * nsplit + s) * Lp + q) * dp + d];
    const long bi = bh / h;
    const int hi = (int)(bh % h);
    out[(bi * L + q) * ld_out + (long)hi * head_pitch + d] = acc * scale;
  }
}

int launch_dq_reduce(const float *part, int nsplit, int b, int h, int L, int Lp, int dp, int width, float scale, float *out,
                     int ld_out, int head_pitch, hipStream_t s) {
  const long total = (long)b * h * L * width;
  long blocks = ceil_div_ll(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dq_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, part, nsplit, h, L, Lp, dp, width, scale, out,
                     ld_out, head_pitch, total);
  HN_LAUNCH_CHECK("dq_reduce");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// dK, dV (explicit K/V binding).  S is formed the other way round -- A = Q tile (M = query rows), B = K tile
// (N = tokens) -- so lane (g, j) holds S[q = 4 g + r][t = j]: P^T and dS^T are then directly the A operands of
// dV += P^T dO and dK += dS^T Q with k-chunk r = query rows {4 g + r}.
// ------------------------------------------------------------------------------------------------
template <int DT, bool DROP = false>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdArgs a, int ntiles, int dh, int inner) {
  constexpr int DP = 16 * DT;
  const int L = a.Lq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int tile = blockIdx.x * 4 + wave;
  const int bh = blockIdx.y, bi = bh / a.h, hi = bh % a.h;
  if (tile >= ntiles) return;
  const int t0 = tile * 16;

  const i32x4 rsQ = make_rsrc(a.Q + (long)bi * a.q_b + (long)hi * a.q_h, rsrc_bytes(L, a.ldq, DP));
  const i32x4 rsG = make_rsrc(a.dO + (long)bi * a.do_b + (long)hi * a.do_h, rsrc_bytes(L, a.lddo, DP));
  const i32x4 krs = make_rsrc(a.Kp + (long)bi * a.k_b + (long)hi * a.k_h, rsrc_bytes(a.N, a.ldk, DP));
  const i32x4 vrs = make_rsrc(a.Vp + (long)bi * a.v_b + (long)hi * a.v_h, rsrc_bytes(a.N, a.ldv, DP));
  float4 kf[DT], vf[DT];                         // B operands: lane (g, j = t) holds K[t][16 s + 4 g ..]
#pragma unroll
  for (int s = 0; s < DT; ++s) {
    kf[s] = buf4(krs, ((t0 + j) * a.ldk + 16 * s + 4 * g) * 4);
    vf[s] = buf4(vrs, ((t0 + j) * a.ldv + 16 * s + 4 * g) * 4);
  }
  float live = (t0 + j) < a.N ? 1.0f : 0.0f;
  if (a.mask) live *= a.mask[(long)bi * a.N + min(t0 + j, a.N - 1)] ? 1.0f : 0.0f;

  f32x4 dK[DT], dV[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) { dK[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // Per-lane byte offsets of every fragment are loop invariant; the query tile advances in the SGPR offset of the buffer loads
  // and the row statistics come through descriptors too (rows past L read 0): no vector instruction computes an address in
  // the loop, and 1 / l is a v_rcp_f32 (1 ulp) instead of a division (ten instructions), masked only in a ragged last tile.
  // Every vector instruction here is matrix time lost (the fp32 MFMA shares the SIMD's issue with the VALU).
  const int qa_off = (j * a.ldq + 4 * g) * 4, ga_off = (j * a.lddo + 4 * g) * 4;
  int qb_off[4], gb_off[4], st_off[4], dl_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    qb_off[r] = ((4 * g + r) * a.ldq + j) * 4;
    gb_off[r] = ((4 * g + r) * a.lddo + j) * 4;
    st_off[r] = (4 * g + r) * 8;
    dl_off[r] = (4 * g + r) * 4;
  }
  const i32x4 rsS = make_rsrc(a.stats + (long)bh * L * 2, (unsigned)L * 8u);
  const i32x4 rsD = make_rsrc(a.delta + (long)bh * L, (unsigned)L * 4u);
  for (int q0 = 0; q0 < L; q0 += 16) {
    float4 qa[DT], ga[DT];                       // A operands: lane (g, i = q) holds Q[q][16 s + 4 g ..]
    float qb[DT][4], gb[DT][4];                  // B operands of the second products: rows 4 g + r, column 16 d + j
    const int qs = q0 * a.ldq * 4, gs = q0 * a.lddo * 4;
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      qa[s] = buf4s(rsQ, qa_off + 64 * s, qs);
      ga[s] = buf4s(rsG, ga_off + 64 * s, gs);
    }
    float mr[4], il[4], dl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        qb[d][r] = hn_buffer_load_x1(rsQ, qb_off[r] + 64 * d, qs, 0);
        gb[d][r] = hn_buffer_load_x1(rsG, gb_off[r] + 64 * d, gs, 0);
      }
      mr[r] = hn_buffer_load_x1(rsS, st_off[r], q0 * 8, 0);
      il[r] = __builtin_amdgcn_rcpf(hn_buffer_load_x1(rsS, st_off[r] + 4, q0 * 8, 0));
      dl[r] = hn_buffer_load_x1(rsD, dl_off[r], q0 * 4, 0);
    }
    if (q0 + 16 > L) {                           // ragged last tile (wave-uniform): rows past L contribute nothing (their l read 0)
#pragma unroll
      for (int r = 0; r < 4; ++r) il[r] = q0 + 4 * g + r < L ? il[r] : 0.0f;
    }
    f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      S = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].x, kf[s].x, S, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s].x, vf[s].x, dP, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].y, kf[s].y, S, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s].y, vf[s].y, dP, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].z, kf[s].z, S, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s].z, vf[s].z, dP, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].w, kf[s].w, S, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s].w, vf[s].w, dP, 0, 0, 0);
    }
    f32x4 P, dS;
    float dm[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    // lanes j .. j + 3 of a column quad hold the same four rows: 4 Philox calls per quad of lanes instead of 16
    if (DROP) drop_quad_transposed(a.drop, (uint32_t)(t0 + j) >> 2, (uint32_t)(bh * L + q0 + 4 * g), (uint32_t)(j & 3), dm);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pn = fexp2(S[r] - mr[r]) * il[r] * live;
      dS[r] = pn * (dP[r] * dm[r] - dl[r]);
      P[r] = pn * dm[r];                                     // dV takes the thinned probabilities
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        dV[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(P[r], gb[d][r], dV[d], 0, 0, 0);
        dK[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(dS[r], qb[d][r], dK[d], 0, 0, 0);
      }
  }
  // C/D map: col = lane & 15 -> d, row = 4 g + r -> token.  Compact (b*N, 2*inner) layout: [dK | dV]
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tok = t0 + 4 * g + r, col = 16 * d + j;
      if (tok < a.N && col < dh) {
        float *row = a.dKV + ((long)bi * a.N + tok) * (2 * inner) + hi * dh + col;
        row[0] = dK[d][r] * a.dk_scale;
        row[inner] = dV[d][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV with the query side in LDS (dp = 64, L * 64 floats x 2 <= 64 KB: the patch-bag cross block and the latent self block of the
// default model).  The kernel above re-fetches Q and dO of its (sample, head) from L2 for every 16-token tile -- 44 fragment-shaped
// loads (16 rows x 64 B each) per 64 MFMAs, the pattern that runs at a third of the full-line rate -- and sits at 0.49
// of the fp32 MFMA peak at cfg4 (225 us).  Here a workgroup lands Q and dO ONCE by LDS-DMA in full 256-byte rows (8 KB per 16-row
// query tile, requested up front in tile order, consumed behind counted waits + one barrier per query tile) and its four waves own
// TWO token tiles each, so one set of fragment reads feeds 128 MFMAs.
//   LDS image: row r of Q / dO is 16 slots of 16 bytes, slot c stored at c ^ sw(r & 15), sw(x) = ((x & 3) << 2) | (x >> 2).  With
//   the S / dP contraction ordered k = 16 g + 4 s + e (lane group g reads slot 4 g + s) both access patterns are conflict-free:
//   the A fragments (16 rows, slots 4 g + s) and the B fragments of dK += dS^T Q / dV += P^T dO (rows 4 g + r, ALL 16 slots: lane j
//   takes slot j, i.e. columns 4 j .. 4 j + 3 -- MFMA e of a quad then produces the output columns {4 j + e}, and a lane ends up with
//   four CONSECUTIVE columns of its token rows: float4 stores).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dkv_glds16(const i32x4 &rsrc, unsigned lds_byte, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_byte), "v"(voffset), "s"(rsrc), "s"(soffset)
               : "memory", "m0");
}
// two fp32 -> three bf16 pairs (h, m, l) with x = h + m + l exactly (gemm_x6.hip's split)
__device__ __forceinline__ void dkv_split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(ra), "v"(rb));
  const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l) : "v"(sa), "v"(sb));
}
template <int N> __device__ __forceinline__ void dkv_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TPW>
__device__ __forceinline__ void attn_bwd_dkv_lds_body(const AttnBwdArgs &a, int ntiles, int dh, int inner, int bx, int by) {
  constexpr int DT = 4, DP = 64, MAXL = 128;
  __shared__ __attribute__((aligned(16))) float lds[2 * MAXL * DP + 4 * 256];          // Q image, dO image, (max, sum) rows, delta
  const int L = a.Lq, nq = (L + 15) >> 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = lane & 15;
  const int bh = by, bi = bh / a.h, hi = bh % a.h;
  const int tile0 = (bx * 4 + wave) * TPW;

  const i32x4 rsQ = make_rsrc(a.Q + (long)bi * a.q_b + (long)hi * a.q_h, rsrc_bytes(L, a.ldq, DP));
  const i32x4 rsG = make_rsrc(a.dO + (long)bi * a.do_b + (long)hi * a.do_h, rsrc_bytes(L, a.lddo, DP));
  const i32x4 krs = make_rsrc(a.Kp + (long)bi * a.k_b + (long)hi * a.k_h, rsrc_bytes(a.N, a.ldk, DP));
  const i32x4 vrs = make_rsrc(a.Vp + (long)bi * a.v_b + (long)hi * a.v_h, rsrc_bytes(a.N, a.ldv, DP));

  // ---- the query side by LDS-DMA: per 16-row tile 4 + 4 one-KB pieces (4 rows each), wave w takes piece w of Q and of dO.  Lane
  // (r4 = lane >> 4, p = lane & 15) fetches slot p ^ sw(row & 15) of row 16 t + 4 w + r4 and lands at slot p.  Wave 0 also lands the
  // row statistics (max, sum) and delta of the (sample, head): (L, 2) + (L) floats behind the two images.
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds);
  const int r4 = lane >> 4, p16 = lane & 15, rl = 4 * wave + r4;            // row within a query tile
  const int swl = ((rl & 3) << 2) | (rl >> 2);
  const int voq = (rl * a.ldq + ((p16 ^ swl) << 2)) * 4, vog = (rl * a.lddo + ((p16 ^ swl) << 2)) * 4;
  auto request_tile = [&](int t) {
    const unsigned dst = lds_base + (unsigned)((t * 16 + 4 * wave) * DP * 4);
    dkv_glds16(rsQ, dst, voq, t * 16 * a.ldq * 4);
    dkv_glds16(rsG, dst + MAXL * DP * 4, vog, t * 16 * a.lddo * 4);
  };
  constexpr int STATS = 2 * MAXL * DP;                                       // float offset of the statistics block
  {
    const i32x4 rsS = make_rsrc(a.stats + (long)bh * L * 2, (unsigned)L * 8u);
    const i32x4 rsD = make_rsrc(a.delta + (long)bh * L, (unsigned)L * 4u);
    // every wave issues the same number of pieces (the counted waits below are per wave): waves 0 / 1 carry the statistics and
    // delta, waves 2 / 3 an out-of-range dummy (lands zeros in an unused slab)
    const unsigned sdst = lds_base + (unsigned)((STATS + wave * 256) * 4);
    if (wave == 0) dkv_glds16(rsS, sdst, lane * 16, 0);
    else if (wave == 1) dkv_glds16(rsD, sdst, lane * 16, 0);
    else dkv_glds16(rsD, sdst, 0x7ffffff0, 0);
  }
  request_tile(0);

  // token side: B operands of S / dP, lane (g, j = token) holds K[t][16 g + 4 s ..] (the contraction order of the header)
  float4 kf[TPW][DT], vf[TPW][DT];
  float live[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int t0 = (tile0 + u) * 16;
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      kf[u][s] = buf4(krs, ((t0 + j) * a.ldk + 16 * g + 4 * s) * 4);
      vf[u][s] = buf4(vrs, ((t0 + j) * a.ldv + 16 * g + 4 * s) * 4);
    }
    live[u] = (t0 + j) < a.N ? 1.0f : 0.0f;
    if (a.mask) live[u] *= a.mask[(long)bi * a.N + min(t0 + j, a.N - 1)] ? 1.0f : 0.0f;
  }
  // Everything requested so far -- statistics, query tile 0, the K / V fragments -- is what the first tile needs: ONE full wait,
  // made here by hand AND through a use of the last fragment (the compiler counts only its own loads: without the use it would
  // put its vmcnt(0) in front of the first MFMA, behind the requests below, and drain them all).  The other query tiles are requested
  // behind it and land under tile 0's MFMAs; no ordinary load is issued after this point.
  dkv_wait_vmcnt<0>();
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    asm volatile("" ::"v"(vf[u][DT - 1].w), "v"(kf[u][DT - 1].w), "v"(live[u]));
  }
  for (int t = 1; t < nq; ++t) request_tile(t);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  f32x4 dK[TPW][DT], dV[TPW][DT];
#pragma unroll
  for (int u = 0; u < TPW; ++u)
#pragma unroll
    for (int d = 0; d < DT; ++d) { dK[u][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV[u][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const int swj = ((j & 3) << 2) | (j >> 2);
  const float *st_ms = lds + STATS, *st_dl = lds + STATS + 256;

  for (int t = 0; t < nq; ++t) {
    const int q0 = t * 16;
    if (t > 0) {
      // this wave's two pieces of tile t have landed once at most the 2 (nq - 1 - t) pieces behind them are outstanding
      const int behind = nq - 1 - t;
      if (behind >= 6) dkv_wait_vmcnt<12>();
      else if (behind == 5) dkv_wait_vmcnt<10>();
      else if (behind == 4) dkv_wait_vmcnt<8>();
      else if (behind == 3) dkv_wait_vmcnt<6>();
      else if (behind == 2) dkv_wait_vmcnt<4>();
      else if (behind == 1) dkv_wait_vmcnt<2>();
      else dkv_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const float *Qs = lds + q0 * DP, *Gs = lds + MAXL * DP + q0 * DP;
    float4 qa[DT], ga[DT], qb[4], gb[4];
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      qa[s] = *(const float4 *)&Qs[j * DP + (((4 * g + s) ^ swj) << 2)];
      ga[s] = *(const float4 *)&Gs[j * DP + (((4 * g + s) ^ swj) << 2)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                // row 4 g + r: sw = (r << 2) | g
      qb[r] = *(const float4 *)&Qs[(4 * g + r) * DP + ((j ^ ((r << 2) | g)) << 2)];
      gb[r] = *(const float4 *)&Gs[(4 * g + r) * DP + ((j ^ ((r << 2) | g)) << 2)];
    }
    float il[4], mc[4], dc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 ms = *(const float2 *)&st_ms[(q0 + 4 * g + r) * 2];
      mc[r] = ms.x;
      il[r] = (q0 + 4 * g + r < L) ? __builtin_amdgcn_rcpf(ms.y) : 0.0f;
      dc[r] = st_dl[q0 + 4 * g + r];
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < DT; ++s) {
        S = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].x, kf[u][s].x, S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s].x, vf[u][s].x, dP, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].y, kf[u][s].y, S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s].y, vf[u][s].y, dP, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].z, kf[u][s].z, S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s].z, vf[u][s].z, dP, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s].w, kf[u][s].w, S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s].w, vf[u][s].w, dP, 0, 0, 0);
      }
      float P[4], dS[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pn = fexp2(S[r] - mc[r]) * il[r] * live[u];
        dS[r] = pn * (dP[r] - dc[r]);
        P[r] = pn;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float qv[4] = {qb[r].x, qb[r].y, qb[r].z, qb[r].w}, gv[4] = {gb[r].x, gb[r].y, gb[r].z, gb[r].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dV[u][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(P[r], gv[e], dV[u][e], 0, 0, 0);
          dK[u][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(dS[r], qv[e], dK[u][e], 0, 0, 0);
        }
      }
    }
  }
  // accumulator e, element r: token 4 g + r, column 4 j + e.
  if (TPW == 2 && a.dkv3 != nullptr) {
    // Straight into the transposed three-plane image of gemm_x6.hip (the weight-gradient product G = dKV^T z contracts over the
    // tokens): a fragment slot is one column's eight k-values of a 16-token k-step; in PAIR order (x6_pair_order, common.h) those
    // are this lane's rows 4 g .. 4 g + 3 of BOTH its tiles, so a lane owns whole 16-byte slots and dKV itself is never stored.
    // The slots of a wave (2 k-steps x 2 column tiles x 3 planes per half = twelve 1 KB fragments) are assembled in the wave's own
    // 12 KB of the Q / dO images' LDS (free behind the barrier) and leave as full, lane-linear 1 KB stores: written slot by slot
    // from the accumulator layout the same bytes were 64 partial lines per instruction and cost the kernel 55 us of 145.
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tile0 + 1 >= ntiles) return;
    typedef unsigned dkv_u32x4 __attribute__((ext_vector_type(4)));
    unsigned char *stg = (unsigned char *)lds + wave * 12288;
    const long ktb = (((long)bi * a.N) >> 4) + tile0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {               // dK columns, then dV columns
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * j + e;                          // column within the head: 0 .. 63
        unsigned hh[4], mm[4], ll[4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 v = half == 0 ? dK[u][e] * a.dk_scale : dV[u][e];
          dkv_split2(v[0], v[1], hh[2 * u], mm[2 * u], ll[2 * u]);
          dkv_split2(v[2], v[3], hh[2 * u + 1], mm[2 * u + 1], ll[2 * u + 1]);
        }
        dkv_u32x4 *dst = (dkv_u32x4 *)(stg + ((((g >> 1) * 2 + (c >> 5)) * 3) * 64 + 32 * (g & 1) + (c & 31)) * 16);
        dst[0] = (dkv_u32x4){hh[0], hh[1], hh[2], hh[3]};
        dst[64] = (dkv_u32x4){mm[0], mm[1], mm[2], mm[3]};
        dst[128] = (dkv_u32x4){ll[0], ll[1], ll[2], ll[3]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int ct0 = (half * inner + hi * 64) >> 5;
#pragma unroll
      for (int f = 0; f < 12; ++f) {                       // f = (k-step * 2 + column tile) * 3 + plane
        const dkv_u32x4 v = *(const dkv_u32x4 *)(stg + f * 1024 + lane * 16);
        const int ktl = f / 6, ctl = (f / 3) & 1, pl = f % 3;
        *(dkv_u32x4 *)(a.dkv3 + ((((ktb + ktl) * a.dkv3_ct + ct0 + ctl) * 3 + pl) * 64 + lane) * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the reads above have left the LDS before the next half overwrites it)
    }
    return;
  }
  // Compact (b*N, 2*inner) layout: [dK | dV]
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    if (tile0 + u >= ntiles) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tok = (tile0 + u) * 16 + 4 * g + r, col = 4 * j;
      if (tok < a.N && col < dh) {
        float *row = a.dKV + ((long)bi * a.N + tok) * (2 * inner) + hi * dh + col;
        *(f32x4 *)row = (f32x4){dK[u][0][r], dK[u][1][r], dK[u][2][r], dK[u][3][r]} * a.dk_scale;
        *(f32x4 *)(row + inner) = (f32x4){dV[u][0][r], dV[u][1][r], dV[u][2][r], dV[u][3][r]};
      }
    }
  }
}

template <int TPW>
__global__ __launch_bounds__(256) void attn_bwd_dkv_lds_kernel(AttnBwdArgs a, int ntiles, int dh, int inner) {
  attn_bwd_dkv_lds_body<TPW>(a, ntiles, dh, inner, (int)blockIdx.x, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// Latent self-attention backward (L = N <= 128, dim_head 64, explicit binding): dQ and dK/dV are independent once delta exists,
// and at small batches each of the two launches leaves most of the chip idle (cfg4 b = 8: 18 + 15 us for 64 (sample, head)
// pairs).  One launch runs both bodies side by side: the first `n_dq` workgroups are the dQ kernel's grid, the rest the dK/dV
// kernel's (bx, by) grid flattened.  Same code, same bits as the two launches.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_self_pair_kernel(AttnBwdArgs a, int ngroups, int gy, int wpb, int n_dq, int ntiles, int dh,
                                                                 int inner, int dkv_bx) {
  if ((int)blockIdx.x < n_dq) {
    attn_bwd_dq_body<4, 2, false, false, 0>(a, ngroups, gy, wpb, (long)blockIdx.x, (long)n_dq);
  } else {
    const int r = (int)blockIdx.x - n_dq;
    attn_bwd_dkv_lds_body<1>(a, ntiles, dh, inner, r % dkv_bx, r / dkv_bx);
  }
}

// true: both products were launched (the caller still runs dq_reduce); false: not this shape, nothing launched
bool launch_attn_bwd_self_pair(const AttnBwdArgs &a, int dh, int inner, hipStream_t s, int *rc_out) {
  static const bool off = getenv("HN_NO_SELF_BWD_PAIR") != nullptr || tuning_env("HN_NO_DKV_LDS") != nullptr;
  *rc_out = HN_OK;
  auto al16p = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  const int ntiles = ceil_div(a.N, 16);
  const bool dkv_ok = a.dp == 64 && a.Lq <= 128 && a.drop.thr == 0 && dh % 4 == 0 && inner % 4 == 0 && a.ldq % 4 == 0 && a.lddo % 4 == 0 &&
                      a.q_b % 4 == 0 && a.q_h % 4 == 0 && a.do_b % 4 == 0 && a.do_h % 4 == 0 && al16p(a.Q) && al16p(a.dO) && al16p(a.dKV) &&
                      a.N >= 64 && ntiles < 64;
  const int nq = bwd_nq(4), ngroups = ceil_div(a.Lp / 16, nq), wpb = ngroups < 4 ? ngroups : 4;
  // (measured: 64 (sample, head) pairs, cfg4 b = 8: 18 + 15 us -> the step 5.20 -> 5.11 ms; 256 pairs, cfg2 b = 32: each launch fills
  // the chip on its own and the pair is 0.2 % slower -- up to 128 pairs it is)
  if ((long)a.b * a.h > 128) return false;
  if (off || !dkv_ok || a.Kp == a.Vp || a.qk_steps != 0 || a.mask != nullptr || attn_bwd_dq_lds_eligible(a) || wpb != 4 || a.Lp % 16 != 0 ||
      a.chunk % 16 != 0)
    return false;
  const int gy = ceil_div(ngroups, wpb);
  const long n_dq = (long)a.nsplit * gy * a.b * a.h;
  const int dkv_bx = ceil_div(ntiles, 4);
  const long n_dkv = (long)dkv_bx * a.b * a.h;
  if (n_dq + n_dkv >= (1L << 31) || (long)a.b * a.h > 65535) return false;
  hipLaunchKernelGGL(attn_bwd_self_pair_kernel, dim3((unsigned)(n_dq + n_dkv)), dim3(256), 0, s, a, ngroups, gy, wpb, (int)n_dq, ntiles, dh, inner,
                     dkv_bx);
  if (hipGetLastError() != hipSuccess) { *rc_out = HN_E_HIP; }
  return true;
}

int launch_attn_bwd_dkv(const AttnBwdArgs &a_in, int dh, int inner, hipStream_t s, bool *wrote_planes) {
  AttnBwdArgs a = a_in;
  if (wrote_planes) *wrote_planes = false;
  HN_REQUIRE(a.dp == 16 || a.dp == 32 || a.dp == 64 || a.dp == 128, HN_E_UNSUPPORTED, "attn_bwd_dkv: dp=%d", a.dp);
  const int ntiles = ceil_div(a.N, 16);
  dim3 grid(ceil_div(ntiles, 4), a.b * a.h), block(256);
  HN_REQUIRE(grid.y <= 65535, HN_E_UNSUPPORTED, "attn_bwd_dkv: b*h too large");
  // query side in LDS: dp = 64, <= 128 query rows, no dropout, 16-byte aligned rows on both sides of the DMA and of the stores
  static const bool no_lds = tuning_env("HN_NO_DKV_LDS") != nullptr;      // development switch: the register-only kernel
  auto al16p = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  if (!no_lds && a.dp == 64 && a.Lq <= 128 && a.drop.thr == 0 && dh % 4 == 0 && inner % 4 == 0 && a.ldq % 4 == 0 && a.lddo % 4 == 0 &&
      a.q_b % 4 == 0 && a.q_h % 4 == 0 && a.do_b % 4 == 0 && a.do_h % 4 == 0 && al16p(a.Q) && al16p(a.dO) && al16p(a.dKV) && a.N >= 64) {
    // the transposed three-plane image instead of dKV: two token tiles per wave, whole pairs inside every sample, every image row
    // (column of dKV) owned by some lane -- else the caller's x6_split_t builds it from the compact layout
    static const bool no_planes = getenv("HN_NO_DKV_PLANES") != nullptr;      // route switch (A/B)
    const bool planes = a.dkv3 != nullptr && !no_planes && ntiles >= 64 && a.N % 32 == 0 && (2 * inner) % 256 == 0 && a.h * dh == inner && dh == 64 &&
                        a.dkv3_ct * 32 == 2 * inner;
    if (!planes) a.dkv3 = nullptr;
    if (ntiles >= 64) hipLaunchKernelGGL(attn_bwd_dkv_lds_kernel<2>, dim3(ceil_div(ntiles, 8), a.b * a.h), block, 0, s, a, ntiles, dh, inner);
    else hipLaunchKernelGGL(attn_bwd_dkv_lds_kernel<1>, dim3(ceil_div(ntiles, 4), a.b * a.h), block, 0, s, a, ntiles, dh, inner);
    HN_LAUNCH_CHECK("attn_bwd_dkv_lds");
    if (wrote_planes) *wrote_planes = planes;
    return HN_OK;
  }
  a.dkv3 = nullptr;
#define HN_DKV(DT_)                                                                                              \
  if (a.drop.thr != 0) hipLaunchKernelGGL((attn_bwd_dkv_kernel<DT_, true>), grid, block, 0, s, a, ntiles, dh, inner); \
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<DT_, false>), grid, block, 0, s, a, ntiles, dh, inner);
  switch (a.dp / 16) {
    case 1: HN_DKV(1) break;
    case 2: HN_DKV(2) break;
    case 4: HN_DKV(4) break;
    default: HN_DKV(8) break;
  }
#undef HN_DKV
  HN_LAUNCH_CHECK("attn_bwd_dkv");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// small per-head elementwise helpers on "head-pitched" matrices: rows = b*L, head hi at column hi*pitch, `width`
// valid columns per head
// ------------------------------------------------------------------------------------------------
// delta[b, h, q] = sum_d X[row, h, d] * Y[row, h, d]
// P lanes per (row, head) pair (P = 16 / 32 / 64, a power of two >= min(width, 64)): coalesced along d, shuffle reduction.
// (One thread per pair walked `width` strided elements serially: 18 us per call.)
template <int P>
__global__ __launch_bounds__(256) void rowdot_heads_kernel(const float *__restrict__ X, int ldx, int xpitch, const float *__restrict__ Y,
                                                           int ldy, int ypitch, int h, int L, int width, long rows,
                                                           float *__restrict__ delta) {
  const int sub = threadIdx.x % P;
  const long pair = ((long)blockIdx.x * blockDim.x + threadIdx.x) / P;
  const bool live = pair < rows * h;
  const long row = live ? pair / h : 0;
  const int hi = live ? (int)(pair % h) : 0;
  const float *x = X + row * ldx + (long)hi * xpitch, *y = Y + row * ldy + (long)hi * ypitch;
  float s = 0.0f;
  if (live)
    for (int d = sub; d < width; d += P) s = fmaf(x[d], y[d], s);
#pragma unroll
  for (int o = P / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (live && sub == 0) delta[((row / L) * h + hi) * L + (row % L)] = s;
}

int launch_rowdot_heads(const float *X, int ldx, int xpitch, const float *Y, int ldy, int ypitch, int h, int L, int width,
                        long rows, float *delta, hipStream_t s) {
  const int P = width <= 16 ? 16 : (width <= 32 ? 32 : 64);
  const unsigned blocks = (unsigned)ceil_div_ll(rows * h * P, 256);
  if (P == 16) hipLaunchKernelGGL(rowdot_heads_kernel<16>, dim3(blocks), dim3(256), 0, s, X, ldx, xpitch, Y, ldy, ypitch, h, L, width, rows, delta);
  else if (P == 32) hipLaunchKernelGGL(rowdot_heads_kernel<32>, dim3(blocks), dim3(256), 0, s, X, ldx, xpitch, Y, ldy, ypitch, h, L, width, rows, delta);
  else hipLaunchKernelGGL(rowdot_heads_kernel<64>, dim3(blocks), dim3(256), 0, s, X, ldx, xpitch, Y, ldy, ypitch, h, L, width, rows, delta);
  HN_LAUNCH_CHECK("rowdot_heads");
  return HN_OK;
}

// dst[row, h, d] = (src[row, h, d] (* mul[row, h, d])) * colscale[d] * scale + coladd[d]   for d < width, 0 for width <= d < pitch
__global__ __launch_bounds__(256) void head_affine_kernel(const float *__restrict__ src, int lds, int spitch, const float *mul, int ldm,
                                                          int mpitch, const float *colscale, const float *coladd, float scale,
                                                          int h, int width, int dpitch, int ldd, long rows, float *__restrict__ dst) {
  const long total = rows * h * dpitch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % dpitch);
    const long t = i / dpitch;
    const int hi = (int)(t % h);
    const long row = t / h;
    float v = 0.0f;
    if (d < width) {
      v = src[row * lds + (long)hi * spitch + d];
      if (mul) v *= mul[row * ldm + (long)hi * mpitch + d];
      v = v * (colscale ? colscale[d] : 1.0f) * scale + (coladd ? coladd[d] : 0.0f);
    }
    dst[row * ldd + (long)hi * dpitch + d] = v;
  }
}

int launch_head_affine(const float *src, int lds, int spitch, const float *mul, int ldm, int mpitch, const float *colscale,
                       const float *coladd, float scale, int h, int width, int dpitch, int ldd, long rows, float *dst,
                       hipStream_t s) {
  long blocks = ceil_div_ll(rows * h * dpitch, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(head_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, lds, spitch, mul, ldm, mpitch, colscale, coladd,
                     scale, h, width, dpitch, ldd, rows, dst);
  HN_LAUNCH_CHECK("head_affine");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// Packed shared context (common.h packed_slot): the token side stores D - 1 channels, the dropped one being minus their sum
// (LayerNorm rows sum to zero).  For any row vector u contracted with a context row z:
//     u . z = sum_{c < D-1} (u_c - u_{D-1}) z_c                       -> fold (mode 0): natural -> packed operand
// and a gradient w.r.t. the packed operand goes back to the natural one as
//     du_c = dup_{slot(c)}  (c < D-1),   du_{D-1} = - sum_c dup_{slot(c)}   -> unfold (mode 1): packed -> natural
// One thread per (row, head), in place, (rows, h, dp) head-pitched.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_fold_kernel(float *__restrict__ x, int ld, int h, int D, int dp, int ks, int mode, long total, int srow) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float *row = x + (i / h) * ld + (i % h) * dp;
  float v[32];                      // dp <= 32 on this path
  for (int c = 0; c < dp; ++c) v[c] = row[c];
  if (mode == 0) {
    const float last = v[D - 1];
    for (int sl = 0; sl < dp; ++sl) {
      const int c = packed_chan(sl, ks);
      row[sl] = (c >= 0 && c < D - 1) ? v[c] - last : 0.0f;
    }
    if (srow) row[dp - 1] = v[dp - 1];      // dropout: the row-sum channel's gradient keeps its column (never a packed slot)
  } else {
    float sum = 0.0f;
    for (int c = 0; c < dp; ++c) {   // channels in order: every kept one is in `sum` when c reaches D - 1
      float o = 0.0f;
      if (c < D - 1) { o = v[packed_slot(c, ks)]; sum += o; }
      else if (c == D - 1) o = -sum;
      row[c] = o;
    }
  }
}

int launch_pack_fold(float *x, int ld, int h, int D, int dp, int ks, int mode, long rows, hipStream_t s, int srow) {
  HN_REQUIRE(dp <= 32 && ks > 0 && D >= 2 && D <= dp - 1, HN_E_SHAPE, "pack_fold: D=%d dp=%d ks=%d", D, dp, ks);
  const long total = rows * h;
  hipLaunchKernelGGL(pack_fold_kernel, dim3((unsigned)ceil_div_ll(total, 256)), dim3(256), 0, s, x, ld, h, D, dp, ks, mode, total, srow);
  HN_LAUNCH_CHECK("pack_fold");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// Shared-context (rank-D) binding under dropout.  Without dropout sum_t p_t = 1 and the value bias folds to a constant:
// A = (P z) gamma + beta.  With thinned probabilities p' the row sum s = sum_t p'_t is a random variable; the forward keeps
// it in column dp-1 of the saved average (the values get a ones column), and
//     A_c = o_c gamma_c + beta_c s        d o_c = dA_c gamma_c        d s = sum_c dA_c beta_c        d beta_c = sum dA_c s
// mode 0: dst = A   mode 1: dst_c = dA_c * s (-> colsum = d beta)   mode 2: dst = d(saved): [dA gamma | 0 .. | sum dA beta]
// Layout of every operand: (rows, h, dp) with D valid channels and the row-sum channel at dp-1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srow_affine_kernel(const float *__restrict__ saved, const float *__restrict__ dA,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta, int mode,
                                                          int D, int dp, long total, float *__restrict__ dst) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % dp);
    const long base = i - c;
    float v = 0.0f;
    if (mode == 0) {
      if (c < D) v = saved[i] * (gamma ? gamma[c] : 1.0f) + (beta ? beta[c] : 0.0f) * saved[base + dp - 1];
    } else if (mode == 1) {
      if (c < D) v = dA[i] * saved[base + dp - 1];
    } else {
      if (c < D) v = dA[i] * (gamma ? gamma[c] : 1.0f);
      else if (c == dp - 1 && beta) {
        for (int k = 0; k < D; ++k) v = fmaf(dA[base + k], beta[k], v);
      }
    }
    dst[i] = v;
  }
}

int launch_srow_affine(const float *saved, const float *dA, const float *gamma, const float *beta, int mode, int h, int D, int dp,
                       long rows, float *dst, hipStream_t s) {
  const long total = rows * h * dp;
  long blocks = ceil_div_ll(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(srow_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, saved, dA, gamma, beta, mode, D, dp, total, dst);
  HN_LAUNCH_CHECK("srow_affine");
  return HN_OK;
}

}  // namespace hn
