// Fused fp32 GEMMs on the CDNA4 matrix cores (exact fp32 MFMA: v_mfma_f32_32x32x2_f32 / 16x16x4_f32).
//
//   C[m, col(n)] = act( alpha * sum_k pro(A[m,k]) * W[n,k] + bias[n] ) (+ R[m,n])          W in nn.Linear layout
//
// Every dense projection of the fusion stack goes through launch_gemm() (reference:
// healnet/models/healnet.py  to_q :403, to_kv :405, to_out :426, FeedForward.net :343-348), with the
// surrounding elementwise work fused in:
//   prologue on A : LayerNorm over k with in-kernel row moments (PreNorm.norm :314), or the affine
//                   half of LayerNorm on an already normalised operand (PreNorm.norm_context :316-319)
//   epilogue      : bias, LeakyReLU(0.01) (:385), SELU/GELU gated linear unit (:323-331),
//                   residual add (:236-245), per-head column re-pitching.
//
// Four kernels behind one launcher:
//   gemm_kernel         64x64 tile, dword loads with lanes along k: any alignment (context-side projections with
//                       odd K / leading dimension: D = 13, 773, 2005; tuned configs with l_d = 119)
//   gemm_t32_kernel     32x64 tile, 16-byte loads, operands streamed (aligned operands, K > 512)
//   gemm_t32a_kernel    32x64 tile, A block resident (aligned, K <= 512): all latent-side GEMMs of the default model
//   gemm_skinny_kernel  M <= 32 rows: weight-streaming GEMV batch (one-token tabular context)
//
// All operand loads are raw buffer loads (see common.h): rows past the end read as 0 in hardware, so the loaders
// carry no per-lane predicates (hipcc turns "load or 0" into branch + load + s_waitcnt vmcnt(0) per element).
#include "common.h"
#include <stdlib.h>

namespace hn {

__device__ __forceinline__ float selu_exact(float x) {
  const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
  return scale * (x > 0.0f ? x : alpha * expm1f(x));
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <bool GLU>
__device__ __forceinline__ float epilogue_value(const GemmArgs &g, float acc, float accg, float bv, float bgv) {
  float v = g.alpha * acc + bv;
  if (GLU) {
    const float gate = g.alpha * accg + bgv;
    // a scalar branch (kept one by the empty asm): as a select BOTH gates are evaluated for every element
    if (g.act == ACT_GLU_SELU) {
      asm volatile("" ::: "memory");
      v *= selu_exact(gate);
    } else {
      asm volatile("" ::: "memory");
      v *= gelu_f(gate);
    }
  } else if (g.act == ACT_LEAKY) {
    v = v > 0.0f ? v : 0.01f * v;
  }
  return v;
}

__device__ __forceinline__ long out_col(const GemmArgs &g, int n) {
  return g.col_group > 0 ? (long)(n / g.col_group) * g.col_group_pitch + (n % g.col_group) : n;
}

// ------------------------------------------------------------------------------------------------
// General kernel.  64x64 tile per 256-thread workgroup, BK = 32, four waves in a 2x2 grid, one 32x32 accumulator
// each.  LDS pitch 36 floats: ds_read_b128 fragment reads are conflict free (row*36 mod 64 hits 16 distinct
// 4-bank slots per 16-lane service group).  One dword per lane with lanes along k: every wave-level load covers
// two full 128-byte row segments regardless of alignment.
// ------------------------------------------------------------------------------------------------
constexpr int BM = 64, BN = 64, BK = 32, LDS_PITCH = 36;

template <bool GLU>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_PITCH];
  __shared__ __attribute__((aligned(16))) float Bs[(GLU ? 2 : 1) * BN * LDS_PITCH];
  __shared__ float row_mu[BM], row_rs[BM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const float *__restrict__ A = g.A + (long)z * g.strideA;
  const float *__restrict__ W = g.W + (long)z * g.strideW;
  float *__restrict__ C = g.C + (long)z * g.strideC;
  const float *bias = g.bias ? g.bias + (long)z * g.strideBias : nullptr;
  const float *R = g.R ? g.R + (long)z * g.strideR : nullptr;
  const i32x4 rsA = make_rsrc(A, rsrc_bytes(g.M, g.lda, g.K));
  const i32x4 rsW = make_rsrc(W, rsrc_bytes(GLU ? g.N + g.glu_offset : g.N, g.ldw, g.K));

  // ---- LayerNorm prologue: shifted one-pass moments, 4 lanes per row
  if (g.pro == PRO_LAYERNORM) {
    const int r = tid >> 2, part = tid & 3;
    const int rowb = (m0 + r) * (int)g.lda * 4;
    const float x0 = hn_buffer_load_x1(rsA, rowb, 0, 0);
    float s1 = 0.0f, s2 = 0.0f;
    for (int k = part; k < g.K; k += 4) {
      const float d = hn_buffer_load_x1(rsA, rowb + k * 4, 0, 0) - x0;
      s1 += d;
      s2 += d * d;
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
    const float inv_k = 1.0f / (float)g.K, dm = s1 * inv_k;
    if (part == 0) { row_mu[r] = x0 + dm; row_rs[r] = 1.0f / sqrtf(fmaxf(s2 * inv_k - dm * dm, 0.0f) + g.eps); }
    __syncthreads();
  }

  const int lk = tid & 31, lr = tid >> 5;     // loader: lane runs along k, 8 row groups
  float ra[8], rb[8], rg[GLU ? 8 : 1];
  auto load_tile = [&](int k0) {
    const int k = k0 + lk;
    const float km = k < g.K ? 1.0f : 0.0f;
    const int kc = min(k, g.K - 1);
    float gam = 1.0f, bet = 0.0f;
    if (g.pro != PRO_NONE) { gam = g.gamma[kc]; bet = g.beta[kc]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = lr + 8 * j;
      float a = hn_buffer_load_x1(rsA, ((m0 + r) * (int)g.lda + kc) * 4, 0, 0);
      if (g.pro == PRO_LAYERNORM) a = (a - row_mu[r]) * row_rs[r] * gam + bet;
      else if (g.pro == PRO_AFFINE) a = a * gam + bet;
      ra[j] = a * km;      // masked along k only: rows past M produce values nobody stores
      rb[j] = hn_buffer_load_x1(rsW, ((n0 + r) * (int)g.ldw + kc) * 4, 0, 0);
      if (GLU) rg[j] = hn_buffer_load_x1(rsW, ((n0 + r + g.glu_offset) * (int)g.ldw + kc) * 4, 0, 0);
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
    const long ocol = out_col(g, n);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      if (m < g.M) {
        float v = epilogue_value<GLU>(g, acc[r], accg[r], bv, bgv);
        if (R) v += R[(long)m * g.ldr + n];
        C[(long)m * g.ldc + ocol] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Large kernel: the context-side K/V projection of patch bags (M = b*N = 32 768 rows, N = 2*inner = 1024, K = D = 773
// at cfg4: 52 GF per block of the model).  Same loader / fragment scheme as gemm_kernel (dword loads with lanes along
// k: any alignment, affine prologue fused) on a 128 x 128 tile: a wave owns 64 x 64 (2 x 2 MFMA tiles), so every
// ds_read_b128 feeds 8 MFMAs instead of 4 and the tile's arithmetic intensity against L2 doubles (32 FLOP/B).
// Work-item order: the 8 column tiles that share one 128-row block of A run back to back on ONE XCD (workgroups are
// dealt round-robin to the 8 XCDs, each with its own L2), so A (100 MB at cfg4) streams from HBM once.
// ------------------------------------------------------------------------------------------------
constexpr int GM = 128, GN = 128;

__global__ __launch_bounds__(256) void gemm_big_kernel(GemmArgs g, int ntm, int ntn) {
  __shared__ __attribute__((aligned(16))) float As[GM * LDS_PITCH];
  __shared__ __attribute__((aligned(16))) float Bs[GN * LDS_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int id = blockIdx.x, xcd = id & 7, seq = id >> 3;
  const int n_tile = seq % ntn, m_tile = (seq / ntn) * 8 + xcd;
  if (m_tile >= ntm) return;
  const int m0 = m_tile * GM, n0 = n_tile * GN;
  const float *__restrict__ A = g.A;
  const float *__restrict__ W = g.W;
  const i32x4 rsA = make_rsrc(A, rsrc_bytes(g.M, g.lda, g.K));
  const i32x4 rsW = make_rsrc(W, rsrc_bytes(g.N, g.ldw, g.K));

  const int lk = tid & 31, lr = tid >> 5;     // loader: lane runs along k, 8 row groups x 16 rows
  float ra[16], rb[16];
  // per-lane byte offsets are loop invariant (row * pitch + lane's k); the k-tile base advances in an SGPR.  Columns past K
  // read the next row (rows past the end read 0 through the descriptor): A is masked with km, so whatever W holds there
  // is multiplied by 0 (weights are finite).
  int offA[16], offW[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    offA[j] = ((m0 + lr + 8 * j) * (int)g.lda + lk) * 4;
    offW[j] = ((n0 + lr + 8 * j) * (int)g.ldw + lk) * 4;
  }
  auto load_tile = [&](int k0) {
    const int k = k0 + lk;
    const float km = k < g.K ? 1.0f : 0.0f;
    const int kc = min(k, g.K - 1);
    float gam = 1.0f, bet = 0.0f;
    if (g.pro == PRO_AFFINE) { gam = g.gamma[kc]; bet = g.beta[kc]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float a = hn_buffer_load_x1(rsA, offA[j], k0 * 4, 0);
      ra[j] = (a * gam + bet) * km;
      rb[j] = hn_buffer_load_x1(rsW, offW[j], k0 * 4, 0);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int r = lr + 8 * j;
      As[r * LDS_PITCH + lk] = ra[j];
      Bs[r * LDS_PITCH + lk] = rb[j];
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, fhalf = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  const int nk = (g.K + BK - 1) / BK;
  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 a4[2], b4[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a4[i] = *(const float4 *)(As + (wm * 64 + 32 * i + frow) * LDS_PITCH + 16 * fhalf + 4 * q);
        b4[i] = *(const float4 *)(Bs + (wn * 64 + 32 * i + frow) * LDS_PITCH + 16 * fhalf + 4 * q);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + 32 * j + frow;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.0f;
    const long ocol = out_col(g, n);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        if (m < g.M) {
          float v = epilogue_value<false>(g, acc[i][j][r], 0.0f, bv, 0.0f);
          if (g.R) v += g.R[(long)m * g.ldr + n];
          g.C[(long)m * g.ldc + ocol] = v;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Aligned kernels (16-byte aligned operands, K % 4 == 0).  32 x 64 output tile -> 256 workgroups even for
// M = 4096, N = 128; operands move in full 128-byte lines (8 lanes x 16 B per row of a 32-wide k-tile) into an
// XOR-swizzled LDS image (slot = c ^ (row & 7), pitch 32 floats, no padding): conflict free for the
// ds_write_b128 of the loader and for the ds_read_b128 fragment reads of the 16x16x4 MFMA; double-buffered
// W tiles with the global loads two k-tiles ahead in registers, ONE barrier per k-tile.
//
// gemm_t32a_kernel (K <= 32*KT) additionally requests its whole 32 x K block of A, and the residual values of its
// epilogue, before anything else: A was written by the previous kernel, usually from another XCD, so its first
// touch is an Infinity-Cache / HBM round trip; issued together they cost one round trip instead of one per
// k-tile.  LayerNorm moments come from those same registers (the 8 loader lanes of a row hold the entire row).
// ------------------------------------------------------------------------------------------------
constexpr int TM = 32, TN = 64, TK = 32;

struct WStage { float4 b0, b1, g0, g1; };

template <bool GLU>
__device__ __forceinline__ void load_w_stage(const GemmArgs &g, const i32x4 &rsW, int n0, int lr, int lc, int kt, WStage &st) {
  const int kc = min(kt * TK + lc * 4, g.K - 4);       // a clamped k only re-reads columns whose A side is zero
  const int ldw = (int)g.ldw;
  st.b0 = buf4(rsW, ((n0 + lr) * ldw + kc) * 4);
  st.b1 = buf4(rsW, ((n0 + lr + 32) * ldw + kc) * 4);
  if (GLU) {
    st.g0 = buf4(rsW, ((n0 + lr + g.glu_offset) * ldw + kc) * 4);
    st.g1 = buf4(rsW, ((n0 + lr + 32 + g.glu_offset) * ldw + kc) * 4);
  }
}

template <bool GLU>
__device__ __forceinline__ void store_w_stage(float *Bs, int lr, int sw, const WStage &st) {
  *(float4 *)&Bs[lr * TK + sw] = st.b0;
  *(float4 *)&Bs[(lr + 32) * TK + sw] = st.b1;
  if (GLU) {
    *(float4 *)&Bs[(TN + lr) * TK + sw] = st.g0;
    *(float4 *)&Bs[(TN + lr + 32) * TK + sw] = st.g1;
  }
}

// 16 (32 with GLU) MFMAs of one 32-wide k-tile; the accumulator chains are interleaved
template <bool GLU>
__device__ __forceinline__ void tile_mfma(const float *At, const float *Bt, int arow, int br0, int fg, f32x4 &acc0, f32x4 &acc1,
                                          f32x4 &accg0, f32x4 &accg1) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int c = 4 * s2 + fg;                       // logical 16-byte slot: k = 16 s2 + 4 fg .. +3
    const float4 a4 = *(const float4 *)&At[arow * TK + ((c ^ (arow & 7)) * 4)];
    const int bsl = (c ^ (br0 & 7)) * 4;             // rows br0, br0 + 16 (and + TN) share (row & 7)
    const float4 b0 = *(const float4 *)&Bt[br0 * TK + bsl];
    const float4 b1 = *(const float4 *)&Bt[(br0 + 16) * TK + bsl];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1.w, acc1, 0, 0, 0);
    if (GLU) {
      const float4 g0 = *(const float4 *)&Bt[(TN + br0) * TK + bsl];
      const float4 g1 = *(const float4 *)&Bt[(TN + br0 + 16) * TK + bsl];
      accg0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, g0.x, accg0, 0, 0, 0);
      accg1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, g1.x, accg1, 0, 0, 0);
      accg0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, g0.y, accg0, 0, 0, 0);
      accg1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, g1.y, accg1, 0, 0, 0);
      accg0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, g0.z, accg0, 0, 0, 0);
      accg1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, g1.z, accg1, 0, 0, 0);
      accg0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, g0.w, accg0, 0, 0, 0);
      accg1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, g1.w, accg1, 0, 0, 0);
    }
  }
}

// epilogue of the 32x64 tile.  16x16 C/D map: col = lane & 15, row = 4 * (lane >> 4) + r.  `res` holds the residual
// values when they were prefetched (t32a), otherwise they are read here.
template <bool GLU, bool PREFETCHED>
__device__ __forceinline__ void tile_store(const GemmArgs &g, const f32x4 &av, const f32x4 &gv, int m_base, int n, const float *bias,
                                           const float *R, float *C, const float (&res)[4]) {
  if (n >= g.N) return;
  const float bv = bias ? bias[n] : 0.0f;
  const float bgv = (GLU && bias) ? bias[n + g.glu_offset] : 0.0f;
  const long ocol = out_col(g, n);
  const float a4[4] = {av.x, av.y, av.z, av.w};
  const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m_base + r;
    if (m < g.M) {
      float v = epilogue_value<GLU>(g, a4[r], g4[r], bv, bgv);
      if (PREFETCHED) v += res[r];
      else if (R) v += R[(long)m * g.ldr + n];
      C[(long)m * g.ldc + ocol] = v;
    }
  }
}

template <bool GLU>
__global__ __launch_bounds__(256) void gemm_t32_kernel(GemmArgs g) {
  constexpr int NB = GLU ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float As[2][TM * TK];
  __shared__ __attribute__((aligned(16))) float Bs[2][NB * TN * TK];
  __shared__ float row_mu[TM], row_rs[TM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fg = lane >> 4, fi = lane & 15;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN, z = blockIdx.z;
  const float *__restrict__ A = g.A + (long)z * g.strideA;
  const float *__restrict__ W = g.W + (long)z * g.strideW;
  float *__restrict__ C = g.C + (long)z * g.strideC;
  const float *bias = g.bias ? g.bias + (long)z * g.strideBias : nullptr;
  const float *R = g.R ? g.R + (long)z * g.strideR : nullptr;
  const i32x4 rsA = make_rsrc(A, rsrc_bytes(g.M, g.lda, g.K));
  const i32x4 rsW = make_rsrc(W, rsrc_bytes(GLU ? g.N + g.glu_offset : g.N, g.ldw, g.K));

  const int lr = tid >> 3, lc = tid & 7;        // loader: row 0..31, 16-byte slot 0..7 of the k-tile
  const int arowb = (m0 + lr) * (int)g.lda * 4;
  if (g.pro == PRO_LAYERNORM) {                 // shifted one-pass moments, 8 lanes per row
    const float x0 = hn_buffer_load_x1(rsA, arowb, 0, 0);
    float s1 = 0.0f, s2 = 0.0f;
    for (int k = lc * 4; k < g.K; k += 32) {
      const float4 v = buf4(rsA, arowb + k * 4);
      const float dx = v.x - x0, dy = v.y - x0, dz = v.z - x0, dw = v.w - x0;
      s1 += (dx + dy) + (dz + dw);
      s2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
    const float inv_k = 1.0f / (float)g.K, dm = s1 * inv_k;
    if (lc == 0) { row_mu[lr] = x0 + dm; row_rs[lr] = 1.0f / sqrtf(fmaxf(s2 * inv_k - dm * dm, 0.0f) + g.eps); }
    __syncthreads();
  }

  auto load_a = [&](int kt) -> float4 {
    const int k = kt * TK + lc * 4;
    const float km = k < g.K ? 1.0f : 0.0f;
    const int kc = min(k, g.K - 4);
    float4 a = buf4(rsA, arowb + kc * 4);
    if (g.pro != PRO_NONE) {
      const float4 gm = *(const float4 *)(g.gamma + kc);
      const float4 bt = *(const float4 *)(g.beta + kc);
      float mu = 0.0f, rs = 1.0f;
      if (g.pro == PRO_LAYERNORM) { mu = row_mu[lr]; rs = row_rs[lr]; }
      a.x = (a.x - mu) * rs * gm.x + bt.x; a.y = (a.y - mu) * rs * gm.y + bt.y;
      a.z = (a.z - mu) * rs * gm.z + bt.z; a.w = (a.w - mu) * rs * gm.w + bt.w;
    }
    return make_float4(a.x * km, a.y * km, a.z * km, a.w * km);
  };
  const int sw = (lc ^ (lr & 7)) * 4;

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, accg0 = {0.f, 0.f, 0.f, 0.f}, accg1 = {0.f, 0.f, 0.f, 0.f};
  const int arow = wm * 16 + fi, br0 = wn * 32 + fi;
  const int nk = (g.K + TK - 1) / TK;

  // pipeline: LDS holds tile t (computing) and t+1 (being written); registers hold t+1 / t+2 in flight
  WStage wa, wb;
  float4 aa, ab;
  aa = load_a(0);
  load_w_stage<GLU>(g, rsW, n0, lr, lc, 0, wa);
  *(float4 *)&As[0][lr * TK + sw] = aa;
  store_w_stage<GLU>(Bs[0], lr, sw, wa);
  __syncthreads();
  if (nk > 1) { aa = load_a(1); load_w_stage<GLU>(g, rsW, n0, lr, lc, 1, wa); }
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 2 < nk) { ab = load_a(kt + 2); load_w_stage<GLU>(g, rsW, n0, lr, lc, kt + 2, wb); }
    tile_mfma<GLU>(As[0], Bs[0], arow, br0, fg, acc0, acc1, accg0, accg1);
    if (kt + 1 < nk) { *(float4 *)&As[1][lr * TK + sw] = aa; store_w_stage<GLU>(Bs[1], lr, sw, wa); }
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 3 < nk) { aa = load_a(kt + 3); load_w_stage<GLU>(g, rsW, n0, lr, lc, kt + 3, wa); }
    tile_mfma<GLU>(As[1], Bs[1], arow, br0, fg, acc0, acc1, accg0, accg1);
    if (kt + 2 < nk) { *(float4 *)&As[0][lr * TK + sw] = ab; store_w_stage<GLU>(Bs[0], lr, sw, wb); }
    __syncthreads();
  }

  const float none[4] = {0.f, 0.f, 0.f, 0.f};
  tile_store<GLU, false>(g, acc0, accg0, m0 + wm * 16 + 4 * fg, n0 + wn * 32 + fi, bias, R, C, none);
  tile_store<GLU, false>(g, acc1, accg1, m0 + wm * 16 + 4 * fg, n0 + wn * 32 + 16 + fi, bias, R, C, none);
}

template <bool GLU, int KT>
__global__ __launch_bounds__(256) void gemm_t32a_kernel(GemmArgs g) {
  constexpr int NB = GLU ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float As[KT * TM * TK];
  __shared__ __attribute__((aligned(16))) float Bs[2][NB * TN * TK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fg = lane >> 4, fi = lane & 15;
  const int m0 = blockIdx.x * TM, z = blockIdx.z;
  int n0 = blockIdx.y * TN;
  if (!GLU && g.W2 != nullptr && n0 >= g.N) {     // column tiles past N belong to the second product (workgroup-uniform)
    n0 -= g.N;
    g.W = g.W2; g.C = g.C2; g.ldc = g.ldc2; g.N = g.N2; g.alpha = g.alpha2;
    g.col_group = g.col_group2; g.col_group_pitch = g.col_group_pitch2;
  }
  const float *__restrict__ A = g.A + (long)z * g.strideA;
  const float *__restrict__ W = g.W + (long)z * g.strideW;
  float *__restrict__ C = g.C + (long)z * g.strideC;
  const float *bias = g.bias ? g.bias + (long)z * g.strideBias : nullptr;
  const float *R = g.R ? g.R + (long)z * g.strideR : nullptr;
  const i32x4 rsA = make_rsrc(A, rsrc_bytes(g.M, g.lda, g.K));
  const i32x4 rsW = make_rsrc(W, rsrc_bytes(GLU ? g.N + g.glu_offset : g.N, g.ldw, g.K));
  const int lr = tid >> 3, lc = tid & 7;
  const int nk = (g.K + TK - 1) / TK;
  const int sw = (lc ^ (lr & 7)) * 4;

  // ---- everything that comes from far away is requested first: the whole A block, the residual values, W tiles 0/1
  float4 av[KT];
  const int arowb = (m0 + lr) * (int)g.lda * 4;
#pragma unroll
  for (int t = 0; t < KT; ++t) av[t] = buf4(rsA, arowb + min(t * TK + lc * 4, g.K - 4) * 4);
  float res0[4] = {0.f, 0.f, 0.f, 0.f}, res1[4] = {0.f, 0.f, 0.f, 0.f};
  if (R) {
    const i32x4 rsR = make_rsrc(R, rsrc_bytes(g.M, g.ldr, g.N));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = ((m0 + wm * 16 + 4 * fg + r) * (int)g.ldr + n0 + wn * 32 + fi) * 4;
      res0[r] = hn_buffer_load_x1(rsR, off, 0, 0);
      res1[r] = hn_buffer_load_x1(rsR, off + 64, 0, 0);
    }
  }
  WStage wa, wb;
  load_w_stage<GLU>(g, rsW, n0, lr, lc, 0, wa);
  if (nk > 1) load_w_stage<GLU>(g, rsW, n0, lr, lc, 1, wb);

  // ---- prologue on the register-resident A block, then park it in LDS (swizzled per k-tile)
  float mu = 0.0f, rs = 1.0f;
  if (g.pro == PRO_LAYERNORM) {
    const float x0 = __shfl(av[0].x, lane & ~7);          // first element of the row (loader lane lc == 0)
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const float km = (t * TK + lc * 4) < g.K ? 1.0f : 0.0f;
      const float dx = av[t].x - x0, dy = av[t].y - x0, dz = av[t].z - x0, dw = av[t].w - x0;
      s1 += km * ((dx + dy) + (dz + dw));
      s2 += km * ((dx * dx + dy * dy) + (dz * dz + dw * dw));
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
    const float inv_k = 1.0f / (float)g.K, dm = s1 * inv_k;
    mu = x0 + dm;
    rs = 1.0f / sqrtf(fmaxf(s2 * inv_k - dm * dm, 0.0f) + g.eps);
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    if (t < nk) {
      const int k = t * TK + lc * 4;
      float4 v = av[t];
      if (g.pro != PRO_NONE) {
        const int kc = min(k, g.K - 4);
        const float4 gm = *(const float4 *)(g.gamma + kc);
        const float4 bt = *(const float4 *)(g.beta + kc);
        v.x = (v.x - mu) * rs * gm.x + bt.x; v.y = (v.y - mu) * rs * gm.y + bt.y;
        v.z = (v.z - mu) * rs * gm.z + bt.z; v.w = (v.w - mu) * rs * gm.w + bt.w;
      }
      const float km = k < g.K ? 1.0f : 0.0f;
      *(float4 *)&As[t * TM * TK + lr * TK + sw] = make_float4(v.x * km, v.y * km, v.z * km, v.w * km);
    }
  }
  store_w_stage<GLU>(Bs[0], lr, sw, wa);
  __syncthreads();

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, accg0 = {0.f, 0.f, 0.f, 0.f}, accg1 = {0.f, 0.f, 0.f, 0.f};
  const int arow = wm * 16 + fi, br0 = wn * 32 + fi;
  // W pipeline: LDS holds tile t and t+1, registers hold t+1 / t+2
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 2 < nk) load_w_stage<GLU>(g, rsW, n0, lr, lc, kt + 2, wa);
    tile_mfma<GLU>(As + kt * TM * TK, Bs[0], arow, br0, fg, acc0, acc1, accg0, accg1);
    if (kt + 1 < nk) store_w_stage<GLU>(Bs[1], lr, sw, wb);
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 3 < nk) load_w_stage<GLU>(g, rsW, n0, lr, lc, kt + 3, wb);
    tile_mfma<GLU>(As + (kt + 1) * TM * TK, Bs[1], arow, br0, fg, acc0, acc1, accg0, accg1);
    if (kt + 2 < nk) store_w_stage<GLU>(Bs[0], lr, sw, wa);
    __syncthreads();
  }

  tile_store<GLU, true>(g, acc0, accg0, m0 + wm * 16 + 4 * fg, n0 + wn * 32 + fi, bias, R, C, res0);
  tile_store<GLU, true>(g, acc1, accg1, m0 + wm * 16 + 4 * fg, n0 + wn * 32 + 16 + fi, bias, R, C, res1);
}

// ------------------------------------------------------------------------------------------------
// gemm_t32a2_kernel: the K = 256 .. 512 latent GEMMs (attention out-projection, second feed-forward layer: N = 128, so only
// 2 x M/32 = 256 workgroups at cfg2 -- one wave per SIMD, every barrier and LDS round trip of 16 sequential k-tiles exposed).
// Same tile, same resident A block, but 8 waves: the second group of four walks the upper half of K on the same output tile
// with its own W pipeline, so the workgroup goes through K/64 stages instead of K/32; the two partial tiles meet in LDS.
// ------------------------------------------------------------------------------------------------
template <int KT, int SPLIT>       // SPLIT groups of four waves, each on its own 1/SPLIT of K
__global__ __launch_bounds__(256 * SPLIT) void gemm_t32a2_kernel(GemmArgs g) {
  constexpr int KH = KT / SPLIT;                   // k-tiles per group
  __shared__ __attribute__((aligned(16))) float As[KT * TM * TK];
  __shared__ __attribute__((aligned(16))) float Bs[SPLIT][2][TN * TK];
  __shared__ float rstat[SPLIT][3][TM];                // per half: sum, sum of squares, (half 0) first element of the row

  const int tid = threadIdx.x, half = tid >> 8, t = tid & 255, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fg = lane >> 4, fi = lane & 15;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
  const float *__restrict__ A = g.A;
  float *__restrict__ C = g.C;
  const float *bias = g.bias;
  const float *R = g.R;
  const i32x4 rsA = make_rsrc(A, rsrc_bytes(g.M, g.lda, g.K));
  const i32x4 rsW = make_rsrc(g.W, rsrc_bytes(g.N, g.ldw, g.K));
  const int lr = t >> 3, lc = t & 7;
  const int nk = (g.K + TK - 1) / TK;              // even, <= KT (launcher)
  const int nh = nk / SPLIT, kt0 = half * nh;      // this group's k-tiles: kt0 .. kt0 + nh - 1
  const int sw = (lc ^ (lr & 7)) * 4;

  // ---- far loads first: this half's part of the A block, the residual values (half 0), W tiles 0 / 1 of the half
  float4 av[KH];
  const int arowb = (m0 + lr) * (int)g.lda * 4;
#pragma unroll
  for (int i = 0; i < KH; ++i) av[i] = buf4(rsA, arowb + min((kt0 + i) * TK + lc * 4, g.K - 4) * 4);
  float res0[4] = {0.f, 0.f, 0.f, 0.f}, res1[4] = {0.f, 0.f, 0.f, 0.f};
  if (R && half == 0) {
    const i32x4 rsR = make_rsrc(R, rsrc_bytes(g.M, g.ldr, g.N));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = ((m0 + wm * 16 + 4 * fg + r) * (int)g.ldr + n0 + wn * 32 + fi) * 4;
      res0[r] = hn_buffer_load_x1(rsR, off, 0, 0);
      res1[r] = hn_buffer_load_x1(rsR, off + 64, 0, 0);
    }
  }
  WStage wa, wb;
  load_w_stage<false>(g, rsW, n0, lr, lc, kt0, wa);
  if (nh > 1) load_w_stage<false>(g, rsW, n0, lr, lc, kt0 + 1, wb);

  // ---- prologue: LayerNorm moments need the whole row = both halves
  float mu = 0.0f, rs = 1.0f;
  if (g.pro == PRO_LAYERNORM) {
    if (half == 0 && lc == 0) rstat[0][2][lr] = av[0].x;
    __syncthreads();
    const float x0 = rstat[0][2][lr];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < KH; ++i) {
      if (i < nh) {
        const float km = ((kt0 + i) * TK + lc * 4) < g.K ? 1.0f : 0.0f;
        const float dx = av[i].x - x0, dy = av[i].y - x0, dz = av[i].z - x0, dw = av[i].w - x0;
        s1 += km * ((dx + dy) + (dz + dw));
        s2 += km * ((dx * dx + dy * dy) + (dz * dz + dw * dw));
      }
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
    if (lc == 0) { rstat[half][0][lr] = s1; rstat[half][1][lr] = s2; }
    __syncthreads();
    s1 = 0.0f; s2 = 0.0f;
#pragma unroll
    for (int h = 0; h < SPLIT; ++h) { s1 += rstat[h][0][lr]; s2 += rstat[h][1][lr]; }
    const float inv_k = 1.0f / (float)g.K, dm = s1 * inv_k;
    mu = x0 + dm;
    rs = 1.0f / sqrtf(fmaxf(s2 * inv_k - dm * dm, 0.0f) + g.eps);
  }
#pragma unroll
  for (int i = 0; i < KH; ++i) {
    if (i < nh) {
      const int k = (kt0 + i) * TK + lc * 4;
      float4 v = av[i];
      if (g.pro != PRO_NONE) {
        const int kc = min(k, g.K - 4);
        const float4 gm = *(const float4 *)(g.gamma + kc);
        const float4 bt = *(const float4 *)(g.beta + kc);
        v.x = (v.x - mu) * rs * gm.x + bt.x; v.y = (v.y - mu) * rs * gm.y + bt.y;
        v.z = (v.z - mu) * rs * gm.z + bt.z; v.w = (v.w - mu) * rs * gm.w + bt.w;
      }
      const float km = k < g.K ? 1.0f : 0.0f;
      *(float4 *)&As[(kt0 + i) * TM * TK + lr * TK + sw] = make_float4(v.x * km, v.y * km, v.z * km, v.w * km);
    }
  }
  store_w_stage<false>(Bs[half][0], lr, sw, wa);
  __syncthreads();

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, accg0 = {0.f, 0.f, 0.f, 0.f}, accg1 = {0.f, 0.f, 0.f, 0.f};
  const int arow = wm * 16 + fi, br0 = wn * 32 + fi;
  for (int i = 0; i < nh; i += 2) {
    if (i + 2 < nh) load_w_stage<false>(g, rsW, n0, lr, lc, kt0 + i + 2, wa);
    tile_mfma<false>(As + (kt0 + i) * TM * TK, Bs[half][0], arow, br0, fg, acc0, acc1, accg0, accg1);
    if (i + 1 < nh) store_w_stage<false>(Bs[half][1], lr, sw, wb);
    __syncthreads();
    if (i + 1 >= nh) break;
    if (i + 3 < nh) load_w_stage<false>(g, rsW, n0, lr, lc, kt0 + i + 3, wb);
    tile_mfma<false>(As + (kt0 + i + 1) * TM * TK, Bs[half][1], arow, br0, fg, acc0, acc1, accg0, accg1);
    if (i + 2 < nh) store_w_stage<false>(Bs[half][0], lr, sw, wa);
    __syncthreads();
  }

  // ---- the upper half hands its partial tile over through LDS (the W buffers are free after the last barrier)
  f32x4 *hand = (f32x4 *)&Bs[0][0][0];             // (SPLIT - 1) x 256 threads x 2 quads = 8 KB per group
  if (half > 0) { hand[(half - 1) * 512 + 2 * t] = acc0; hand[(half - 1) * 512 + 2 * t + 1] = acc1; }
  __syncthreads();
  if (half > 0) return;
#pragma unroll
  for (int h = 0; h < SPLIT - 1; ++h) { acc0 += hand[h * 512 + 2 * t]; acc1 += hand[h * 512 + 2 * t + 1]; }
  tile_store<false, true>(g, acc0, accg0, m0 + wm * 16 + 4 * fg, n0 + wn * 32 + fi, bias, R, C, res0);
  tile_store<false, true>(g, acc1, accg1, m0 + wm * 16 + 4 * fg, n0 + wn * 32 + 16 + fi, bias, R, C, res1);
}

// ------------------------------------------------------------------------------------------------
// Skinny variant for M <= 32 rows (the tabular / omic modality: one context token per sample, so the
// K/V projection is b rows x 2005 features against an 8.2 MB weight -- a weight-streaming, HBM-bound
// GEMV batch, not MFMA work).  One workgroup produces 4 output columns for all rows: lanes run
// along k (coalesced dword loads of the weight rows, no alignment requirement), the four waves split
// the k range, every lane keeps 32 x 4 partial sums which are folded across the wave with a
// reduce-scatter butterfly (126 shuffles instead of 768) and across the waves through LDS.
// ------------------------------------------------------------------------------------------------
template <int CNT, int NV>
__device__ __forceinline__ void fold_half(float (&v)[NV], int lane, int mask) {
  const bool up = (lane & mask) != 0;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const float keep = up ? v[i + CNT] : v[i];
    const float send = up ? v[i] : v[i + CNT];
    v[i] = keep + __shfl_xor(send, mask);
  }
}

// SM = rows per workgroup: 32, or 8 for batches of up to 8 samples (the reference's tuned runs): a quarter of the row loads and
// accumulators, which pays for FOUR k-chunks in flight per trip (K = 2005: 2 dependent round trips per wave instead of 8 --
// 12.5 us at b = 8 before, where 14 workgroups cannot hide any of them)
template <int SM>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g) {
  constexpr int SN = 4, U = SM == 8 ? 4 : 2;      // k-chunks in flight per trip
  __shared__ float part[4][SM * SN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * SN, z = blockIdx.z, m0 = blockIdx.y * SM;      // blockIdx.y: SM-row chunk (batches above SM samples)
  const i32x4 rsA = make_rsrc(g.A + (long)z * g.strideA, rsrc_bytes(g.M, g.lda, g.K));
  const i32x4 rsW = make_rsrc(g.W + (long)z * g.strideW, rsrc_bytes(g.N, g.ldw, g.K));
  float acc[SM * SN];
#pragma unroll
  for (int i = 0; i < SM * SN; ++i) acc[i] = 0.0f;
  for (int k0 = wave * 64; k0 < g.K; k0 += 256 * U) {
    float w[U][SN], av[U][SM], gam[U], bet[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 256 * u + lane;
      const float km = k < g.K ? 1.0f : 0.0f;
      const int kc = min(k, g.K - 1);
#pragma unroll
      for (int j = 0; j < SN; ++j) w[u][j] = hn_buffer_load_x1(rsW, ((n0 + j) * (int)g.ldw + kc) * 4, 0, 0);
#pragma unroll
      for (int m = 0; m < SM; ++m) av[u][m] = hn_buffer_load_x1(rsA, ((m0 + m) * (int)g.lda + kc) * 4, 0, 0);
      gam[u] = 1.0f; bet[u] = 0.0f;
      if (g.pro == PRO_AFFINE) { gam[u] = g.gamma[kc]; bet[u] = g.beta[kc]; }
      gam[u] *= km;
      bet[u] *= km;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int m = 0; m < SM; ++m) {
        const float a = av[u][m] * gam[u] + bet[u];           // rows past M carry beta only, into outputs nobody stores
#pragma unroll
        for (int j = 0; j < SN; ++j) acc[m * SN + j] = fmaf(a, w[u][j], acc[m * SN + j]);
      }
  }
  // reduce-scatter butterfly: after the fold over lane bit b the lane keeps the half of its values whose index bit matches
  if constexpr (SM == 32) {
    fold_half<64>(acc, lane, 32);
    fold_half<32>(acc, lane, 16);
    fold_half<16>(acc, lane, 8);
    fold_half<8>(acc, lane, 4);
    fold_half<4>(acc, lane, 2);
    fold_half<2>(acc, lane, 1);
    part[wave][2 * lane] = acc[0];          // lane now owns flattened outputs 2*lane, 2*lane + 1 (index = m*SN + j)
    part[wave][2 * lane + 1] = acc[1];
  } else {
    fold_half<16>(acc, lane, 32);
    fold_half<8>(acc, lane, 16);
    fold_half<4>(acc, lane, 8);
    fold_half<2>(acc, lane, 4);
    fold_half<1>(acc, lane, 2);
    acc[0] += __shfl_xor(acc[0], 1);        // 32 outputs on 64 lanes: the pair (lane, lane ^ 1) owns output lane >> 1
    if ((lane & 1) == 0) part[wave][lane >> 1] = acc[0];
  }
  __syncthreads();
  if (threadIdx.x < SM * SN) {
    const int idx = threadIdx.x, m = m0 + idx / SN, n = n0 + idx % SN;
    if (m < g.M && n < g.N) {
      float v = g.alpha * (part[0][idx] + part[1][idx] + part[2][idx] + part[3][idx]);
      if (g.bias) v += g.bias[(long)z * g.strideBias + n];
      if (g.act == ACT_LEAKY) v = v > 0.0f ? v : 0.01f * v;
      if (g.R) v += g.R[(long)z * g.strideR + (long)m * g.ldr + n];
      g.C[(long)z * g.strideC + (long)m * g.ldc + out_col(g, n)] = v;
    }
  }
}

// gemm_skinny_kernel with per-entry operands (blockIdx.z): see GemmSkinnyMulti.  SM = 8 for batches of up to 8 samples, as above.
// SN = weight rows (output columns) per workgroup.  (32, 4): every workgroup re-reads all 32 activation rows for its four weight rows --
// at K = 2005, N = 3 x 512 (the one-token block's value projection, cfg2 b = 32) that is 98 MB through the L2s for 12 MB of weights, and
// it is that traffic, not the four round trips, the launch's 14.8 us were (eight waves on two trips: 18.9 us).  (16, 8) has the same 128
// accumulators per lane and two thirds of the requests: 8 weight + 16 activation rows per k.
template <int SM, int SN = 4>
__global__ __launch_bounds__(256) void gemm_skinny_multi_kernel(GemmSkinnyMulti g) {
  constexpr int U = SM == 8 ? 4 : 2;      // k-chunks in flight per trip
  __shared__ float part[4][SM * SN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * SN, z = blockIdx.z, m0 = blockIdx.y * SM;
  const float *gam_p = g.gamma[z], *bet_p = g.beta[z], *bias = g.bias[z];
  float *C = g.C[z];
  const i32x4 rsA = make_rsrc(g.A[z], rsrc_bytes(g.M, g.lda, g.K));
  const i32x4 rsW = make_rsrc(g.W[z], rsrc_bytes(g.N, g.ldw, g.K));
  float acc[SM * SN];
#pragma unroll
  for (int i = 0; i < SM * SN; ++i) acc[i] = 0.0f;
  for (int k0 = wave * 64; k0 < g.K; k0 += 256 * U) {
    float w[U][SN], av[U][SM], gam[U], bet[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 256 * u + lane;
      const float km = k < g.K ? 1.0f : 0.0f;
      const int kc = min(k, g.K - 1);
#pragma unroll
      for (int j = 0; j < SN; ++j) w[u][j] = hn_buffer_load_x1(rsW, ((n0 + j) * (int)g.ldw + kc) * 4, 0, 0);
#pragma unroll
      for (int m = 0; m < SM; ++m) av[u][m] = hn_buffer_load_x1(rsA, ((m0 + m) * (int)g.lda + kc) * 4, 0, 0);
      gam[u] = 1.0f; bet[u] = 0.0f;
      if (g.pro == PRO_AFFINE) { gam[u] = gam_p[kc]; bet[u] = bet_p[kc]; }
      gam[u] *= km;
      bet[u] *= km;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int m = 0; m < SM; ++m) {
        const float a = av[u][m] * gam[u] + bet[u];
#pragma unroll
        for (int j = 0; j < SN; ++j) acc[m * SN + j] = fmaf(a, w[u][j], acc[m * SN + j]);
      }
  }
  if constexpr (SM * SN == 128) {
    fold_half<64>(acc, lane, 32);
    fold_half<32>(acc, lane, 16);
    fold_half<16>(acc, lane, 8);
    fold_half<8>(acc, lane, 4);
    fold_half<4>(acc, lane, 2);
    fold_half<2>(acc, lane, 1);
    part[wave][2 * lane] = acc[0];
    part[wave][2 * lane + 1] = acc[1];
  } else {
    fold_half<16>(acc, lane, 32);
    fold_half<8>(acc, lane, 16);
    fold_half<4>(acc, lane, 8);
    fold_half<2>(acc, lane, 4);
    fold_half<1>(acc, lane, 2);
    acc[0] += __shfl_xor(acc[0], 1);
    if ((lane & 1) == 0) part[wave][lane >> 1] = acc[0];
  }
  __syncthreads();
  if (threadIdx.x < SM * SN) {
    const int idx = threadIdx.x, m = m0 + idx / SN, n = n0 + idx % SN;
    if (m < g.M && n < g.N) {
      float v = part[0][idx] + part[1][idx] + part[2][idx] + part[3][idx];
      if (bias) v += bias[n];
      if (g.act == ACT_LEAKY) v = v > 0.0f ? v : 0.01f * v;
      C[(long)m * g.ldc + n] = v;
    }
  }
}

int launch_gemm_skinny_multi(const GemmSkinnyMulti &g, hipStream_t s) {
  HN_REQUIRE(g.nz >= 1 && g.nz <= HN_SKINNY_MAXZ && g.M > 0 && g.N > 0 && g.K > 0, HN_E_SHAPE, "gemm_skinny_multi: nz=%d M=%d N=%d K=%d",
             g.nz, g.M, g.N, g.K);
  HN_REQUIRE(((long)(g.M + 64) * g.lda) * 4 < (1L << 31) && ((long)(g.N + 64) * g.ldw) * 4 < (1L << 31), HN_E_UNSUPPORTED,
             "gemm_skinny_multi: an operand spans more than 2 GiB");
  static const bool no_wide = tuning_env("HN_NO_SKINNY_WIDE") != nullptr;      // development switch: four weight rows per workgroup for every shape
  if (g.M <= 8) hipLaunchKernelGGL(gemm_skinny_multi_kernel<8>, dim3(ceil_div(g.N, 4), 1, g.nz), dim3(256), 0, s, g);
  else if (g.K >= 1024 && g.N % 8 == 0 && !no_wide)
    hipLaunchKernelGGL((gemm_skinny_multi_kernel<16, 8>), dim3(g.N / 8, ceil_div(g.M, 16), g.nz), dim3(256), 0, s, g);
  else hipLaunchKernelGGL(gemm_skinny_multi_kernel<32>, dim3(ceil_div(g.N, 4), ceil_div(g.M, 32), g.nz), dim3(256), 0, s, g);
  HN_LAUNCH_CHECK("gemm_skinny_multi");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// Tall and narrow: C (M, N <= 128) = alpha * pro(A) W^T with M in the tens of thousands -- the K/V projection of a patch bag
// through ONE 16 .. 64-wide head (the reference's tuned configs: 32 768 x 773 -> 32 / 64 / 128 columns).  The product is an HBM
// stream of A (101 MB, 1.6 GFLOP at N = 32): what matters is bytes in flight, not tiles.  On the 128 x 128 kernel above 256
// workgroups each keep 8 KB of A in flight behind a barrier per k-step: 1.3 TB/s (79 us).  Here a wave owns 16 rows and reads
// them straight into registers with 16-byte loads, one 64-column chunk ahead (no LDS, no barrier on the A side); the MFMA
// 16x16x4 contracts k = 4 g + e in its e-th issue, so a lane's float4 IS its A operand of four consecutive MFMAs and the same
// holds for a float4 of a W row on the B side.  W (any alignment) is staged per chunk in LDS with dword loads, double-buffered.
// 64 rows per workgroup, 17 - 70 KB of LDS: 2 - 4 workgroups per CU.
// ------------------------------------------------------------------------------------------------
template <int NT, int KC>      // N <= 16 * NT; k-chunk of KC columns (NT = 8: 32, the static LDS limit)
__global__ __launch_bounds__(256) void gemm_tall_narrow_kernel(GemmArgs g) {
  constexpr int WP = KC + 4, NW = 16 * NT, NJ = KC / 16, WQ = NW * KC / 256;
  __shared__ __attribute__((aligned(16))) float Ws[2][NW * WP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, gq = lane >> 4;
  const long row0 = (long)blockIdx.x * 64 + wave * 16;
  const long arow_i = row0 + i < g.M ? row0 + i : g.M - 1;
  const float *arow = g.A + arow_i * g.lda;
  const int nchunks = (g.K + KC - 1) / KC;
  const bool affine = g.pro == PRO_AFFINE;

  auto load_a = [&](int c, float4 (&a)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = c * KC + 16 * j + 4 * gq;
      // (a float4 past K stays inside the row's pitch or the next row; clamped to the last whole quad of the row)
      const int kl = k + 4 <= g.lda ? k : (int)g.lda - 4;
      float4 v = *(const float4 *)(arow + kl);
      if (kl != k) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (affine) {
        const float4 ga = k + 4 <= g.K ? *(const float4 *)(g.gamma + k) : make_float4(k < g.K ? g.gamma[k] : 0.f, k + 1 < g.K ? g.gamma[k + 1] : 0.f, k + 2 < g.K ? g.gamma[k + 2] : 0.f, 0.f);
        const float4 be = k + 4 <= g.K ? *(const float4 *)(g.beta + k) : make_float4(k < g.K ? g.beta[k] : 0.f, k + 1 < g.K ? g.beta[k + 1] : 0.f, k + 2 < g.K ? g.beta[k + 2] : 0.f, 0.f);
        v.x = v.x * ga.x + be.x; v.y = v.y * ga.y + be.y; v.z = v.z * ga.z + be.z; v.w = v.w * ga.w + be.w;
      }
      // columns past K contribute nothing whatever the buffer holds there
      v.x = k < g.K ? v.x : 0.0f; v.y = k + 1 < g.K ? v.y : 0.0f; v.z = k + 2 < g.K ? v.z : 0.0f; v.w = k + 3 < g.K ? v.w : 0.0f;
      a[j] = v;
    }
  };
  // W chunk c: NW rows x 64 columns, thread t takes elements t, t + 256, ... (a row's 64 floats are contiguous: coalesced)
  float wreg[WQ];
  auto load_w = [&](int c) {
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int idx = tid + 256 * q, n = idx / KC, kk = idx % KC, k = c * KC + kk;
      wreg[q] = (n < g.N && k < g.K) ? g.W[(long)n * g.ldw + k] : 0.0f;
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int idx = tid + 256 * q;
      Ws[buf][(idx / KC) * WP + (idx % KC)] = wreg[q];
    }
  };

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // One chunk ahead in registers.  Measured at 32 768 x 773 -> 32 (profiles/r03_zc_*): 54 us = 1.9 TB/s against 79 us on the
  // 128 x 128 kernel; three chunks in flight per wave: 62 us; 208-column chunks (832 contiguous bytes per row and wave, 256
  // VGPRs): 103 us.  What is left is not latency: the wave count is fixed at M / 16 (two per SIMD) and every load instruction
  // touches 16 rows 3 KB apart.
  float4 a_cur[NJ], a_nxt[NJ];
  load_a(0, a_cur);
  load_w(0);
  store_w(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    const bool more = c + 1 < nchunks;
    if (more) { load_a(c + 1, a_nxt); load_w(c + 1); }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 bw = *(const float4 *)&Ws[buf][(16 * t + i) * WP + 16 * j + 4 * gq];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].x, bw.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].y, bw.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].z, bw.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].w, bw.w, acc[t], 0, 0, 0);
      }
    }
    if (more) {
      store_w(buf ^ 1);                      // (the other buffer: its readers passed the barrier of the previous iteration)
#pragma unroll
      for (int j = 0; j < NJ; ++j) a_cur[j] = a_nxt[j];
    }
    __syncthreads();
  }
  // accumulator element r: row 4 gq + r, column 16 t + i
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = 16 * t + i;
    if (n >= g.N) continue;
    const int col = g.col_group > 0 ? (n / g.col_group) * g.col_group_pitch + n % g.col_group : n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = row0 + 4 * gq + r;
      if (row < g.M) g.C[row * g.ldc + col] = g.alpha * acc[t][r];
    }
  }
}

static bool aligned_eligible(const GemmArgs &g) {
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  if (g.K % 4 != 0 || g.lda % 4 != 0 || g.ldw % 4 != 0) return false;
  if (g.strideA % 4 != 0 || g.strideW % 4 != 0) return false;
  if (!al16(g.A) || !al16(g.W)) return false;
  if (g.pro != PRO_NONE && (!al16(g.gamma) || !al16(g.beta))) return false;
  return true;
}

int launch_gemm(const GemmArgs &g_in, hipStream_t s) {
  if (g_in.W2 != nullptr) {
    // two products on one A: a single launch on the resident-A tile kernel when it applies, two launches otherwise
    const GemmArgs &d = g_in;
    const bool plain = d.act == ACT_NONE && !d.bias && !d.R && d.batch == 1 && d.N % 64 == 0 && d.K <= 128;
    if (!(plain && aligned_eligible(d) && ((uintptr_t)d.W2 & 15) == 0 && !(d.M <= 32 && d.K >= 512))) {
      GemmArgs a = d, b2 = d;
      a.W2 = nullptr;
      b2.W2 = nullptr; b2.W = d.W2; b2.C = d.C2; b2.ldc = d.ldc2; b2.N = d.N2; b2.alpha = d.alpha2;
      b2.col_group = d.col_group2; b2.col_group_pitch = d.col_group_pitch2;
      const int rc = launch_gemm(a, s);
      return rc != HN_OK ? rc : launch_gemm(b2, s);
    }
    HN_REQUIRE(d.A && d.W && d.C && d.C2, HN_E_NULL, "gemm: NULL operand");
    dim3 grid(ceil_div(d.M, TM), ceil_div(d.N, TN) + ceil_div(d.N2, TN), 1);
    hipLaunchKernelGGL((gemm_t32a_kernel<false, 4>), grid, dim3(256), 0, s, d);
    HN_LAUNCH_CHECK("gemm_t32a(dual)");
    return HN_OK;
  }
  const GemmArgs &g = g_in;
  HN_REQUIRE(g.A && g.W && g.C, HN_E_NULL, "gemm: NULL operand");
  HN_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0, HN_E_SHAPE, "gemm: M=%d N=%d K=%d batch=%d", g.M, g.N,
             g.K, g.batch);
  HN_REQUIRE(g.pro == PRO_NONE || (g.gamma && g.beta), HN_E_NULL, "gemm: prologue needs gamma and beta");
  const bool glu = g.act == ACT_GLU_SELU || g.act == ACT_GLU_GELU;
  // the buffer descriptors address operands with 32-bit byte offsets
  const long a_span = ((long)(g.M + 64) * g.lda) * 4, w_span = ((long)(g.N + (glu ? g.glu_offset : 0) + 64) * g.ldw) * 4;
  HN_REQUIRE(a_span < (1L << 31) && w_span < (1L << 31) && (long)(g.M + 64) * g.ldr * 4 < (1L << 31), HN_E_UNSUPPORTED,
             "gemm: an operand spans more than 2 GiB (M=%d lda=%ld N=%d ldw=%ld)", g.M, g.lda, g.N, g.ldw);
  // M <= 32 always; up to 512 rows (32-row chunks re-stream the weight from L2) when the operands are not 16-byte aligned
  // (K = 2005: the tile kernels do not apply and the generic one runs 8 workgroups -- 265 us at b = 64 against 2 x 9)
  if ((g.M <= 32 || (g.M <= 512 && !aligned_eligible(g))) && !glu && g.pro != PRO_LAYERNORM && g.K >= 512) {
    if (g.M <= 8) hipLaunchKernelGGL(gemm_skinny_kernel<8>, dim3(ceil_div(g.N, 4), 1, g.batch), dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_skinny_kernel<32>, dim3(ceil_div(g.N, 4), ceil_div(g.M, 32), g.batch), dim3(256), 0, s, g);
    HN_LAUNCH_CHECK("gemm_skinny");
    return HN_OK;
  }
  // tall and narrow (one 16 .. 64-wide head over a patch bag): the register-streaming kernel
  static const bool no_tall = tuning_env("HN_NO_TALL_NARROW") != nullptr;      // development switch
  if (!no_tall && !glu && g.batch == 1 && (g.pro == PRO_NONE || g.pro == PRO_AFFINE) && g.M >= 2048 && g.K >= 128 && g.N >= 16 && g.N <= 128 &&
      !g.bias && !g.R && g.act == ACT_NONE && g.lda % 4 == 0 && g.lda >= 4 && ((uintptr_t)g.A & 15) == 0 &&
      (g.pro == PRO_NONE || (((uintptr_t)g.gamma | (uintptr_t)g.beta) & 15) == 0)) {
    const dim3 grid((unsigned)ceil_div(g.M, 64));
    if (g.N <= 32) hipLaunchKernelGGL((gemm_tall_narrow_kernel<2, 64>), grid, dim3(256), 0, s, g);
    else if (g.N <= 64) hipLaunchKernelGGL((gemm_tall_narrow_kernel<4, 64>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_tall_narrow_kernel<8, 32>), grid, dim3(256), 0, s, g);
    HN_LAUNCH_CHECK("gemm_tall_narrow");
    return HN_OK;
  }
  // (N >= 256, or from 32 columns when the operands are not 16-byte aligned -- one cross head of 16 .. 103 on a 773-channel patch
  // bag, the reference's tuned configs: the alternative there is the 64 x 64 dword-load kernel at a third of this one's rate)
  if (!glu && g.batch == 1 && g.pro != PRO_LAYERNORM && g.M >= 2048 && g.K >= 256 && (g.N >= 256 || (g.N >= 32 && !aligned_eligible(g)))) {
    const int ntm = ceil_div(g.M, GM), ntn = ceil_div(g.N, GN);
    const long blocks = (long)ceil_div(ntm, 8) * 8 * ntn;
    HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "gemm: grid too large");
    hipLaunchKernelGGL(gemm_big_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, ntm, ntn);
    HN_LAUNCH_CHECK("gemm_big");
    return HN_OK;
  }
  if (aligned_eligible(g) && ceil_div(g.N, TN) <= 65535) {
    dim3 grid32(ceil_div(g.M, TM), ceil_div(g.N, TN), g.batch);
    if (g.K <= 128) {
      if (glu) hipLaunchKernelGGL((gemm_t32a_kernel<true, 4>), grid32, dim3(256), 0, s, g);
      else hipLaunchKernelGGL((gemm_t32a_kernel<false, 4>), grid32, dim3(256), 0, s, g);
    } else if (g.K <= 512 && !glu && g.batch == 1 && g.K % 64 == 0 && g.K >= 256 && (long)grid32.x * grid32.y <= 1024) {
      // few workgroups (N = 128 latent GEMMs): the 8-wave split-K form; larger grids keep the 4-wave kernel (more of them per CU)
      hipLaunchKernelGGL((gemm_t32a2_kernel<16, 2>), grid32, dim3(512), 0, s, g);      // (a 4-way split measured the same)
    } else if (g.K <= 512 && !glu) {
      hipLaunchKernelGGL((gemm_t32a_kernel<false, 16>), grid32, dim3(256), 0, s, g);
    } else if (glu) {
      hipLaunchKernelGGL(gemm_t32_kernel<true>, grid32, dim3(256), 0, s, g);
    } else {
      hipLaunchKernelGGL(gemm_t32_kernel<false>, grid32, dim3(256), 0, s, g);
    }
    HN_LAUNCH_CHECK("gemm_t32");
    return HN_OK;
  }
  dim3 grid(ceil_div(g.M, BM), ceil_div(g.N, BN), g.batch);
  HN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, HN_E_UNSUPPORTED, "gemm: grid too large (N=%d batch=%d)", g.N, g.batch);
  if (glu)
    hipLaunchKernelGGL(gemm_kernel<true>, grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(gemm_kernel<false>, grid, dim3(256), 0, s, g);
  HN_LAUNCH_CHECK("gemm");
  return HN_OK;
}

}  // namespace hn
