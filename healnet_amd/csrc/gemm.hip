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

int launch_gemm(const GemmArgs &g, hipStream_t s) {
  HN_REQUIRE(g.A && g.W && g.C, HN_E_NULL, "gemm: NULL operand");
  HN_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0, HN_E_SHAPE, "gemm: M=%d N=%d K=%d batch=%d", g.M, g.N,
             g.K, g.batch);
  dim3 grid(ceil_div(g.M, BM), ceil_div(g.N, BN), g.batch);
  HN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, HN_E_UNSUPPORTED, "gemm: grid too large (N=%d batch=%d)", g.N, g.batch);
  const bool glu = g.act == ACT_GLU_SELU || g.act == ACT_GLU_GELU;
  if (glu)
    hipLaunchKernelGGL(gemm_kernel<true>, grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(gemm_kernel<false>, grid, dim3(256), 0, s, g);
  HN_LAUNCH_CHECK("gemm");
  return HN_OK;
}

}  // namespace hn
