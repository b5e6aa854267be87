// Fused fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
//   C[m, col(n)] = act( alpha * sum_k pro(A[m,k]) * W[n,k] + bias[n] ) (+ R[m,n])
//
// Every dense projection of the fusion stack goes through this one kernel (reference:
// healnet/models/healnet.py  to_q :403, to_kv :405, to_out :426, FeedForward.net :343-348), with the
// surrounding elementwise work fused in:
//   prologue on A : LayerNorm over k with in-kernel row moments (PreNorm.norm :314), or the affine
//                   half of LayerNorm on an already normalised operand (PreNorm.norm_context :316-319)
//   epilogue      : bias, LeakyReLU(0.01) (:385), SELU/GELU gated linear unit (:323-331),
//                   residual add (:236-245), per-head column re-pitching.
//
// Tile: 64x64 per 256-thread workgroup, BK = 32; four waves in a 2x2 grid, each owning one 32x32
// accumulator (16 VGPRs).  Operands are staged through LDS with a 36-float row pitch, which makes the
// ds_read_b128 fragment reads bank-conflict free (row*36 mod 64 hits 16 distinct 4-bank slots per
// 16-lane service group).  Global loads are one dword per lane with lanes running along k, so every
// wave-level load covers two full 128-byte row segments regardless of K / leading-dimension alignment
// (the tuned HEALNet configs use odd sizes: l_d 119, D 13/773/2005).  The kernel is bound by the fp32
// MFMA rate (16 x 64-cycle MFMAs per wave per k-tile vs. 16 dword loads), not by the loader.
#include "common.h"

namespace hn {

constexpr int BM = 64, BN = 64, BK = 32, LDS_PITCH = 36;

__device__ __forceinline__ float selu_exact(float x) {
  const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
  return scale * (x > 0.0f ? x : alpha * expm1f(x));
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <bool GLU>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ float As[BM * LDS_PITCH];
  __shared__ float Bs[(GLU ? 2 : 1) * BN * LDS_PITCH];
  __shared__ float row_mu[BM], row_rs[BM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const float *__restrict__ A = g.A + (long)z * g.strideA;
  const float *__restrict__ W = g.W + (long)z * g.strideW;
  float *__restrict__ C = g.C + (long)z * g.strideC;
  const float *bias = g.bias ? g.bias + (long)z * g.strideBias : nullptr;
  const float *R = g.R ? g.R + (long)z * g.strideR : nullptr;

  // ---- LayerNorm prologue: per-row mean / rstd over the whole K extent (two-pass, 4 lanes per row)
  if (g.pro == PRO_LAYERNORM) {
    const int r = tid >> 2, part = tid & 3;
    const int m = m0 + r;
    float s = 0.0f;
    if (m < g.M)
      for (int k = part; k < g.K; k += 4) s += A[(long)m * g.lda + k];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    const float mu = s / (float)g.K;
    float q = 0.0f;
    if (m < g.M)
      for (int k = part; k < g.K; k += 4) { float d = A[(long)m * g.lda + k] - mu; q += d * d; }
    q += __shfl_xor(q, 1);
    q += __shfl_xor(q, 2);
    if (part == 0) { row_mu[r] = mu; row_rs[r] = 1.0f / sqrtf(q / (float)g.K + g.eps); }
    __syncthreads();
  }

  // ---- loader geometry: lane runs along k, 8 row groups
  const int lk = tid & 31, lr = tid >> 5;
  float ra[8], rb[8], rg[GLU ? 8 : 1];

  auto load_tile = [&](int k0) {
    const int k = k0 + lk;
    const bool kin = k < g.K;
    float gam = 1.0f, bet = 0.0f;
    if (g.pro != PRO_NONE && kin) { gam = g.gamma ? g.gamma[k] : 1.0f; bet = g.beta ? g.beta[k] : 0.0f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = lr + 8 * j;
      const int m = m0 + r, n = n0 + r;
      float a = 0.0f;
      if (kin && m < g.M) {
        a = A[(long)m * g.lda + k];
        if (g.pro == PRO_LAYERNORM) a = (a - row_mu[r]) * row_rs[r] * gam + bet;
        else if (g.pro == PRO_AFFINE) a = a * gam + bet;
      }
      ra[j] = a;
      rb[j] = (kin && n < g.N) ? W[(long)n * g.ldw + k] : 0.0f;
      if (GLU) rg[j] = (kin && n < g.N) ? W[(long)(n + g.glu_offset) * g.ldw + k] : 0.0f;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = lr + 8 * j;
      As[r * LDS_PITCH + lk] = ra[j];
      Bs[r * LDS_PITCH + lk] = rb[j];
      if (GLU) Bs[(BN + r) * LDS_PITCH + lk] = rg[j];
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, fhalf = lane >> 5;
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 accg = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  const int nk = (g.K + BK - 1) / BK;
  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
    // fragments: lane (row, half) holds k = 16*half + 0..15 of its row; A and B use the same k order
    const float4 *ap = (const float4 *)(As + (wm * 32 + frow) * LDS_PITCH + 16 * fhalf);
    const float4 *bp = (const float4 *)(Bs + (wn * 32 + frow) * LDS_PITCH + 16 * fhalf);
    float4 a4[4], b4[4], g4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a4[i] = ap[i]; b4[i] = bp[i]; }
    if (GLU) {
      const float4 *gp = (const float4 *)(Bs + (BN + wn * 32 + frow) * LDS_PITCH + 16 * fhalf);
#pragma unroll
      for (int i = 0; i < 4; ++i) g4[i] = gp[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[i].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[i].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[i].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[i].w, acc, 0, 0, 0);
      if (GLU) {
        accg = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, g4[i].x, accg, 0, 0, 0);
        accg = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, g4[i].y, accg, 0, 0, 0);
        accg = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, g4[i].z, accg, 0, 0, 0);
        accg = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, g4[i].w, accg, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
  const int n = n0 + wn * 32 + frow;
  if (n < g.N) {
    const float bv = bias ? bias[n] : 0.0f;
    const float bgv = (GLU && bias) ? bias[n + g.glu_offset] : 0.0f;
    const long ocol = g.col_group > 0 ? (long)(n / g.col_group) * g.col_group_pitch + (n % g.col_group) : n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      if (m < g.M) {
        float v = g.alpha * acc[r] + bv;
        if (GLU) {
          const float gate = g.alpha * accg[r] + bgv;
          v *= (g.act == ACT_GLU_SELU) ? selu_exact(gate) : gelu_f(gate);
        } else if (g.act == ACT_LEAKY) {
          v = v > 0.0f ? v : 0.01f * v;
        }
        if (R) v += R[(long)m * g.ldr + n];
        C[(long)m * g.ldc + ocol] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Register-direct variant for 16-byte aligned operands with K % 32 == 0 (every latent-side projection of
// the default model: K = l_d = 128 or K = inner / 4*l_d = 512).
//
// No LDS, no barriers: one wave owns a 32-row block and walks `nt` 32-column tiles of it; lane
// (r = lane & 31, h = lane >> 5) loads its MFMA fragments straight from global memory as 16-byte vectors
// (A[m0 + r][k0 + 16 h ..], W[n0 + r][k0 + 16 h ..]: each wave-level load covers 32 full 64-byte half
// rows).  The latent-side GEMMs are small (M = b*l_c rows, 0.5-1 GFLOP) and latency-bound, so what matters
// is many independent waves with their loads in flight, not operand reuse through LDS: the LDS-staged
// kernel above needs 74 us for M=4096, N=128, K=512 (128 workgroups, one barrier pair per k-tile).
// With K <= 128 the wave keeps its whole (LayerNorm-ed) A block in registers and only streams W.
// ------------------------------------------------------------------------------------------------
template <bool GLU>
__device__ __forceinline__ void direct_epilogue(const GemmArgs &g, const f32x16 &acc, const f32x16 &accg, int m0, int n,
                                                int fhalf, const float *bias, const float *R, float *C) {
  if (n >= g.N) return;
  const float bv = bias ? bias[n] : 0.0f;
  const float bgv = (GLU && bias) ? bias[n + g.glu_offset] : 0.0f;
  const long ocol = g.col_group > 0 ? (long)(n / g.col_group) * g.col_group_pitch + (n % g.col_group) : n;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
    if (m < g.M) {
      float v = g.alpha * acc[r] + bv;
      if (GLU) {
        const float gate = g.alpha * accg[r] + bgv;
        v *= (g.act == ACT_GLU_SELU) ? selu_exact(gate) : gelu_f(gate);
      } else if (g.act == ACT_LEAKY) {
        v = v > 0.0f ? v : 0.01f * v;
      }
      if (R) v += R[(long)m * g.ldr + n];
      C[(long)m * g.ldc + ocol] = v;
    }
  }
}

#define HN_MFMA4(ACC, AV, BV)                                                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, (BV).x, ACC, 0, 0, 0);     \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, (BV).y, ACC, 0, 0, 0);     \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, (BV).z, ACC, 0, 0, 0);     \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, (BV).w, ACC, 0, 0, 0);

// KS > 0: K == 32*KS, A resident in registers.  KS == 0: A streamed, any K % 32 == 0.
template <int KS, bool GLU>
__global__ __launch_bounds__(256) void gemm_direct_kernel(GemmArgs g, int nt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int m0 = blockIdx.x * 32, z = blockIdx.z;
  const int tile0 = (blockIdx.y * 4 + wave) * nt;
  if (tile0 * 32 >= g.N) return;
  const float *__restrict__ A = g.A + (long)z * g.strideA;
  const float *__restrict__ W = g.W + (long)z * g.strideW;
  float *__restrict__ C = g.C + (long)z * g.strideC;
  const float *bias = g.bias ? g.bias + (long)z * g.strideBias : nullptr;
  const float *R = g.R ? g.R + (long)z * g.strideR : nullptr;
  const int mrow = min(m0 + fr, g.M - 1);
  const float *arow = A + (long)mrow * g.lda + 16 * fh;

  constexpr int NA = KS > 0 ? KS * 4 : 1;
  float4 areg[NA];
  if constexpr (KS > 0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) areg[i] = *(const float4 *)(arow + 32 * (i >> 2) + 4 * (i & 3));
    if (g.pro == PRO_LAYERNORM) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < NA; ++i) s += (areg[i].x + areg[i].y) + (areg[i].z + areg[i].w);
      s += __shfl_xor(s, 32);
      const float mu = s / (float)g.K;
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const float dx = areg[i].x - mu, dy = areg[i].y - mu, dz = areg[i].z - mu, dw = areg[i].w - mu;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      q += __shfl_xor(q, 32);
      const float rs = 1.0f / sqrtf(q / (float)g.K + g.eps);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        areg[i].x = (areg[i].x - mu) * rs; areg[i].y = (areg[i].y - mu) * rs;
        areg[i].z = (areg[i].z - mu) * rs; areg[i].w = (areg[i].w - mu) * rs;
      }
    }
    if (g.pro != PRO_NONE) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int k = 32 * (i >> 2) + 4 * (i & 3) + 16 * fh;
        const float4 gm = g.gamma ? *(const float4 *)(g.gamma + k) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 bt = g.beta ? *(const float4 *)(g.beta + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        areg[i].x = areg[i].x * gm.x + bt.x; areg[i].y = areg[i].y * gm.y + bt.y;
        areg[i].z = areg[i].z * gm.z + bt.z; areg[i].w = areg[i].w * gm.w + bt.w;
      }
    }
  }

  for (int t = 0; t < nt; ++t) {
    const int n0 = (tile0 + t) * 32;
    if (n0 >= g.N) break;
    const int nrow = min(n0 + fr, g.N - 1);
    const float *wrow = W + (long)nrow * g.ldw + 16 * fh;
    const float *grow = W + (long)(nrow + (GLU ? g.glu_offset : 0)) * g.ldw + 16 * fh;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 accg = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (KS > 0) {
      float4 wc[4], gc[4], wn[4], gn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { wc[i] = *(const float4 *)(wrow + 4 * i); if (GLU) gc[i] = *(const float4 *)(grow + 4 * i); }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            wn[i] = *(const float4 *)(wrow + 32 * (s + 1) + 4 * i);
            if (GLU) gn[i] = *(const float4 *)(grow + 32 * (s + 1) + 4 * i);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          HN_MFMA4(acc, areg[4 * s + i], wc[i])
          if (GLU) { HN_MFMA4(accg, areg[4 * s + i], gc[i]) }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { wc[i] = wn[i]; if (GLU) gc[i] = gn[i]; }
      }
    } else {
      const int nsteps = g.K >> 5;
      float4 ac[4], wc[4], gc[4], an[4], wn[4], gn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ac[i] = *(const float4 *)(arow + 4 * i);
        wc[i] = *(const float4 *)(wrow + 4 * i);
        if (GLU) gc[i] = *(const float4 *)(grow + 4 * i);
      }
      for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            an[i] = *(const float4 *)(arow + 32 * (s + 1) + 4 * i);
            wn[i] = *(const float4 *)(wrow + 32 * (s + 1) + 4 * i);
            if (GLU) gn[i] = *(const float4 *)(grow + 32 * (s + 1) + 4 * i);
          }
        }
        if (g.pro == PRO_AFFINE) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int k = 32 * s + 4 * i + 16 * fh;
            const float4 gm = g.gamma ? *(const float4 *)(g.gamma + k) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 bt = g.beta ? *(const float4 *)(g.beta + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            ac[i].x = ac[i].x * gm.x + bt.x; ac[i].y = ac[i].y * gm.y + bt.y;
            ac[i].z = ac[i].z * gm.z + bt.z; ac[i].w = ac[i].w * gm.w + bt.w;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          HN_MFMA4(acc, ac[i], wc[i])
          if (GLU) { HN_MFMA4(accg, ac[i], gc[i]) }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { ac[i] = an[i]; wc[i] = wn[i]; if (GLU) gc[i] = gn[i]; }
      }
    }
    direct_epilogue<GLU>(g, acc, accg, m0, n0 + fr, fh, bias, R, C);
  }
}

// ------------------------------------------------------------------------------------------------
// Skinny variant for M <= 32 rows (the tabular / omic modality: one context token per sample, so the
// K/V projection is b rows x 2005 features against an 8.2 MB weight -- a weight-streaming, HBM-bound
// GEMV batch, not MFMA work).  One workgroup produces TN = 4 output columns for all rows: lanes run
// along k (coalesced dword loads of the weight rows, no alignment requirement), the four waves split
// the k range, every lane keeps 32 x 4 partial sums which are folded across the wave with a
// reduce-scatter butterfly (126 shuffles instead of 768) and across the waves through LDS.
// ------------------------------------------------------------------------------------------------
template <int CNT>
__device__ __forceinline__ void fold_half(float (&v)[128], int lane, int mask) {
  const bool up = (lane & mask) != 0;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const float keep = up ? v[i + CNT] : v[i];
    const float send = up ? v[i] : v[i + CNT];
    v[i] = keep + __shfl_xor(send, mask);
  }
}

__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g) {
  constexpr int TN = 4, TM = 32;
  __shared__ float part[4][TM * TN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * TN, z = blockIdx.z;
  const float *__restrict__ A = g.A + (long)z * g.strideA;
  const float *__restrict__ W = g.W + (long)z * g.strideW;
  float acc[TM * TN];
#pragma unroll
  for (int i = 0; i < TM * TN; ++i) acc[i] = 0.0f;
  for (int k0 = wave * 64; k0 < g.K; k0 += 256) {
    const int k = k0 + lane;
    const bool kin = k < g.K;
    float w[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) w[j] = (kin && n0 + j < g.N) ? W[(long)(n0 + j) * g.ldw + k] : 0.0f;
    float gam = 1.0f, bet = 0.0f;
    if (g.pro == PRO_AFFINE && kin) { gam = g.gamma ? g.gamma[k] : 1.0f; bet = g.beta ? g.beta[k] : 0.0f; }
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      float a = 0.0f;
      if (kin && m < g.M) a = A[(long)m * g.lda + k] * gam + bet;
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[m * TN + j] = fmaf(a, w[j], acc[m * TN + j]);
    }
  }
  fold_half<64>(acc, lane, 32);
  fold_half<32>(acc, lane, 16);
  fold_half<16>(acc, lane, 8);
  fold_half<8>(acc, lane, 4);
  fold_half<4>(acc, lane, 2);
  fold_half<2>(acc, lane, 1);
  part[wave][2 * lane] = acc[0];          // lane now owns flattened outputs 2*lane, 2*lane + 1 (index = m*TN + j)
  part[wave][2 * lane + 1] = acc[1];
  __syncthreads();
  if (threadIdx.x < TM * TN) {
    const int idx = threadIdx.x, m = idx / TN, n = n0 + idx % TN;
    if (m < g.M && n < g.N) {
      float v = g.alpha * (part[0][idx] + part[1][idx] + part[2][idx] + part[3][idx]);
      if (g.bias) v += g.bias[(long)z * g.strideBias + n];
      if (g.act == ACT_LEAKY) v = v > 0.0f ? v : 0.01f * v;
      if (g.R) v += g.R[(long)z * g.strideR + (long)m * g.ldr + n];
      const long ocol = g.col_group > 0 ? (long)(n / g.col_group) * g.col_group_pitch + (n % g.col_group) : n;
      g.C[(long)z * g.strideC + (long)m * g.ldc + ocol] = v;
    }
  }
}

static bool direct_eligible(const GemmArgs &g) {
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  if (g.K % 32 != 0 || g.lda % 4 != 0 || g.ldw % 4 != 0) return false;
  if (g.strideA % 4 != 0 || g.strideW % 4 != 0) return false;
  if (!al16(g.A) || !al16(g.W)) return false;
  if (g.pro != PRO_NONE && ((g.gamma && !al16(g.gamma)) || (g.beta && !al16(g.beta)))) return false;
  if (g.pro == PRO_LAYERNORM && g.K > 128) return false;   // LN needs the row resident in registers
  return true;
}

template <bool GLU>
static void launch_direct(const GemmArgs &g, hipStream_t s) {
  const int mt = ceil_div(g.M, 32), ntiles = ceil_div(g.N, 32);
  // enough waves to cover the chip a few times over, but let a wave reuse its A block when N is large
  int nt = 1;
  while (nt < 4 && (long)mt * ceil_div(ntiles, nt * 2) >= 2048) nt *= 2;
  dim3 grid(mt, ceil_div(ntiles, 4 * nt), g.batch);
  const int ks = g.K <= 128 ? g.K / 32 : 0;
  switch (ks) {
    case 1: hipLaunchKernelGGL((gemm_direct_kernel<1, GLU>), grid, dim3(256), 0, s, g, nt); break;
    case 2: hipLaunchKernelGGL((gemm_direct_kernel<2, GLU>), grid, dim3(256), 0, s, g, nt); break;
    case 3: hipLaunchKernelGGL((gemm_direct_kernel<3, GLU>), grid, dim3(256), 0, s, g, nt); break;
    case 4: hipLaunchKernelGGL((gemm_direct_kernel<4, GLU>), grid, dim3(256), 0, s, g, nt); break;
    default: hipLaunchKernelGGL((gemm_direct_kernel<0, GLU>), grid, dim3(256), 0, s, g, nt); break;
  }
}

int launch_gemm(const GemmArgs &g, hipStream_t s) {
  HN_REQUIRE(g.A && g.W && g.C, HN_E_NULL, "gemm: NULL operand");
  HN_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0, HN_E_SHAPE, "gemm: M=%d N=%d K=%d batch=%d", g.M, g.N,
             g.K, g.batch);
  dim3 grid(ceil_div(g.M, BM), ceil_div(g.N, BN), g.batch);
  HN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, HN_E_UNSUPPORTED, "gemm: grid too large (N=%d batch=%d)", g.N, g.batch);
  const bool glu = g.act == ACT_GLU_SELU || g.act == ACT_GLU_GELU;
  if (g.M <= 32 && !glu && g.pro != PRO_LAYERNORM && g.K >= 512) {
    hipLaunchKernelGGL(gemm_skinny_kernel, dim3(ceil_div(g.N, 4), 1, g.batch), dim3(256), 0, s, g);
    HN_LAUNCH_CHECK("gemm_skinny");
    return HN_OK;
  }
  if (direct_eligible(g) && ceil_div(g.N, 128) <= 65535) {
    if (glu) launch_direct<true>(g, s); else launch_direct<false>(g, s);
    HN_LAUNCH_CHECK("gemm_direct");
    return HN_OK;
  }
  if (glu)
    hipLaunchKernelGGL(gemm_kernel<true>, grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(gemm_kernel<false>, grid, dim3(256), 0, s, g);
  HN_LAUNCH_CHECK("gemm");
  return HN_OK;
}

}  // namespace hn
