// Backward building blocks of the fusion stack (the reference has no explicit backward: autograd of
// healnet/models/healnet.py, driven by surv_loss.backward() at healnet/main.py:464).  Only parameter gradients
// and the gradient of the latent array flow; the modality inputs need no gradient.
//
//   gemm_ex_kernel   C[i,j] (+)= alpha * sum_c A(i,c) B(j,c) with arbitrary operand strides: the three GEMM
//                    flavours of a Linear layer's backward (dX = dY W, dW = dY^T X) on the fp32 matrix cores
//   colsum_kernel    bias gradients and every other "sum over rows"
//   ln_bwd_kernel    LayerNorm backward (dx and per-block partials of dgamma / dbeta)
//   leaky_bwd / glu_bwd / head_bwd   elementwise and head pieces
#include "common.h"
#include <stdlib.h>

namespace hn {

// ------------------------------------------------------------------------------------------------
// strided GEMM: operand element (row r, contraction c) lives at base + r*rs + c*cs
// ------------------------------------------------------------------------------------------------
constexpr int XM = 64, XN = 64, XK = 32, XP = 36;

__global__ __launch_bounds__(256) void gemm_ex_kernel(GemmExArgs g) {
  __shared__ __attribute__((aligned(16))) float As[XM * XP];
  __shared__ __attribute__((aligned(16))) float Bs[XN * XP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * XM, j0 = blockIdx.y * XN, z = blockIdx.z;
  const float *__restrict__ A = g.A + (long)z * g.strideA;
  const float *__restrict__ B = g.B + (long)z * g.strideB;
  float *__restrict__ C = g.C + (long)z * g.strideC;
  // split-k launch: slice z covers contraction indices [z*K, min(k_total, (z+1)*K))
  const int Kz = g.k_total > 0 ? min(g.K, g.k_total - z * g.K) : g.K;

  // loader geometry per operand: lanes run along whichever index is contiguous in memory
  const bool a_c_contig = g.a_cs == 1, b_c_contig = g.b_cs == 1;
  float ra[8], rb[8];
  auto load_operand = [&](const float *P, long rs, long cs, bool c_contig, int r0, int rmax, int c0, float (&reg)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      int r, c;
      if (c_contig) { c = tid & 31; r = (tid >> 5) + 8 * t; }       // 32 lanes along c, 8 row groups
      else { r = tid & 63; c = (tid >> 6) + 4 * t; }                 // 64 lanes along r, 4 column groups
      const int rr = r0 + r, cc = c0 + c;
      const float ok = (rr < rmax && cc < Kz) ? 1.0f : 0.0f;
      const float v = P[(long)min(rr, rmax - 1) * rs + (long)min(cc, Kz - 1) * cs];     // clamped: always valid
      reg[t] = v * ok;
    }
  };
  auto store_operand = [&](float *S, bool c_contig, const float (&reg)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      int r, c;
      if (c_contig) { c = tid & 31; r = (tid >> 5) + 8 * t; }
      else { r = tid & 63; c = (tid >> 6) + 4 * t; }
      S[r * XP + c] = reg[t];
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, fhalf = lane >> 5;
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int nk = (Kz + XK - 1) / XK;
  load_operand(A, g.a_rs, g.a_cs, a_c_contig, i0, g.M, 0, ra);
  load_operand(B, g.b_rs, g.b_cs, b_c_contig, j0, g.N, 0, rb);
  for (int kt = 0; kt < nk; ++kt) {
    store_operand(As, a_c_contig, ra);
    store_operand(Bs, b_c_contig, rb);
    __syncthreads();
    if (kt + 1 < nk) {
      load_operand(A, g.a_rs, g.a_cs, a_c_contig, i0, g.M, (kt + 1) * XK, ra);
      load_operand(B, g.b_rs, g.b_cs, b_c_contig, j0, g.N, (kt + 1) * XK, rb);
    }
    const float4 *ap = (const float4 *)(As + (wm * 32 + frow) * XP + 16 * fhalf);
    const float4 *bp = (const float4 *)(Bs + (wn * 32 + frow) * XP + 16 * fhalf);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a4 = ap[q], b4 = bp[q];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int j = j0 + wn * 32 + frow;
  if (j < g.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      if (i < g.M) {
        float *dst = C + (long)i * g.ldc + j;
        const float v = g.alpha * acc[r];
        *dst = g.accumulate ? *dst + v : v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ part, int nsplit, long mn, int N, float *__restrict__ C,
                                                            long ldc, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += (long)gridDim.x * blockDim.x) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;     // four independent chains (loads in flight), fixed order -> deterministic
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {
      a0 += part[(long)k * mn + i]; a1 += part[(long)(k + 1) * mn + i];
      a2 += part[(long)(k + 2) * mn + i]; a3 += part[(long)(k + 3) * mn + i];
    }
    for (; k < nsplit; ++k) a0 += part[(long)k * mn + i];
    const float acc = (a0 + a1) + (a2 + a3);
    float *dst = C + (i / N) * ldc + (i % N);
    *dst = accumulate ? *dst + acc : acc;
  }
}

// ------------------------------------------------------------------------------------------------
// TN product  C[m, n] (+)= alpha * sum_k A[k, m] * B[k, n]   (A: K x M, B: K x N, both row-major): every
// weight gradient dW = dY^T X of the path and G = dKV^T z.  The contraction runs over the rows, which is exactly the
// operand layout of v_mfma_f32_32x32x2_f32 (lane l <-> row/col l % 32, k = l / 32): a wave's A / B operand for a
// k-pair is ONE coalesced dword load of two 128-byte row segments -- no LDS, no transposition, no barrier.
// A wave owns a 64 x 64 tile (2 x 2 MFMA tiles, 4 loads per 4 MFMAs), a workgroup 128 x 128; the long contraction is
// cut into grid.z slices that write partial tiles to scratch, summed in fixed order by splitk_reduce_kernel
// (deterministic).  Ragged M / N / K go through the buffer descriptor's range check (invalid lanes get an
// out-of-range offset, rows past the slice read 0): no predicate in the loop.
// ------------------------------------------------------------------------------------------------
struct GemmTnArgs {
  const float *A; long lda;
  const float *B; long ldb;
  float *C; long ldc;          // nsplit == 1: the destination; else scratch (nsplit, M, N)
  int M, N, K, kslice, nsplit;
  float alpha; int accumulate;
  float *colsum; int colsum_accumulate;      // optional: sum_k A[k, m] (bias gradient); nsplit > 1: partials (nsplit, M) in scratch
  int batch; long strideA, strideB, strideC; // independent products (per-head weight gradients); grid.z = batch * nsplit
};

template <int UN>      // k-pairs in flight: 8 for short slices (latency decides; 16 costs 272 VGPRs for nothing), 4 for long ones
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 128 + (wave >> 1) * 64, n0 = blockIdx.y * 128 + (wave & 1) * 64;
  const int z = blockIdx.z % g.nsplit, bt = blockIdx.z / g.nsplit;
  if (m0 >= g.M || n0 >= g.N) return;
  g.A += bt * g.strideA;
  g.B += bt * g.strideB;
  const int k_begin = z * g.kslice, k_end = min(g.K, k_begin + g.kslice);
  const int rows = k_end - k_begin;                     // >= 1: nsplit = ceil(K / kslice)
  const int half = lane >> 5, col = lane & 31;
  const i32x4 ars = make_rsrc(g.A + (long)k_begin * g.lda, rows > 0 ? (unsigned)(((long)(rows - 1) * g.lda + g.M) * 4) : 0u);
  const i32x4 brs = make_rsrc(g.B + (long)k_begin * g.ldb, rows > 0 ? (unsigned)(((long)(rows - 1) * g.ldb + g.N) * 4) : 0u);
  int aoff[2], boff[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int m = m0 + 32 * c + col, n = n0 + 32 * c + col;
    aoff[c] = m < g.M ? (int)(((long)half * g.lda + m) * 4) : 0x7ffffff0;
    boff[c] = n < g.N ? (int)(((long)half * g.ldb + n) * 4) : 0x7ffffff0;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // bias gradient for free: the waves of the first column tile also add up their A operands (lane (half, col) sees
  // A[2p + half, m] of every k-pair: 2 VALU adds per pair, the two halves meet in one shuffle at the end); no extra memory
  // traffic, no colsum launch.  (Ones-vector MFMAs did the same at +50 % matrix work for exactly those waves.)
  const bool do_colsum = g.colsum != nullptr && n0 == 0;
  float csum[2] = {0.0f, 0.0f};

  const int npairs = (rows + 1) / 2;
  const int sa = (int)(2 * g.lda * 4), sb = (int)(2 * g.ldb * 4);
  float a[UN][2], b[UN][2];
  // The scalar offset must stay inside the descriptor's range (the range check subtracts it from num_records), so
  // pairs past the end are not fetched at all (wave-uniform branch) instead of relying on the range check.
  auto load = [&](int pair, float (&av)[2], float (&bv)[2]) {
    if (pair < npairs) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        av[c] = hn_buffer_load_x1(ars, aoff[c], pair * sa, 0);
        bv[c] = hn_buffer_load_x1(brs, boff[c], pair * sb, 0);
      }
    } else {
      av[0] = av[1] = bv[0] = bv[1] = 0.0f;
    }
  };
#pragma unroll
  for (int u = 0; u < UN; ++u) load(u, a[u], b[u]);
  for (int p0 = 0; p0 < npairs; p0 += UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      float ca[2] = {a[u][0], a[u][1]}, cb[2] = {b[u][0], b[u][1]};
      load(p0 + UN + u, a[u], b[u]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[i], cb[j], acc[i][j], 0, 0, 0);
      if (do_colsum) { csum[0] += ca[0]; csum[1] += ca[1]; }
    }
  }
  if (do_colsum) {
    float *cs = g.colsum + (g.nsplit > 1 ? (long)z * g.M : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float v = csum[i] + __shfl_xor(csum[i], 32);
      const int m = m0 + 32 * i + col;
      if (half == 0 && m < g.M) {
        if (g.nsplit > 1) cs[m] = v;
        else cs[m] = g.colsum_accumulate ? cs[m] + v : v;
      }
    }
  }
  float *C = g.C + (g.nsplit > 1 ? ((long)bt * g.nsplit + z) * g.M * g.ldc : bt * g.strideC);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 32 * j + col;
      if (n >= g.N) continue;
      // accumulate: all sixteen old values are requested before the first store (a load behind a store to possibly the same
      // address waits for it: the one-by-one form was 64 dependent round trips per lane, 34 us for a 128 x 512 x 8 product)
      float old[16];
      const bool rmw = g.nsplit == 1 && g.accumulate;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
        old[r] = (rmw && m < g.M) ? C[(long)m * g.ldc + n] : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < g.M) C[(long)m * g.ldc + n] = g.nsplit > 1 ? acc[i][j][r] : old[r] + g.alpha * acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// K <= 16 rows: the weight gradient of a one-token context (the omic K/V projection: dW[1024, 2005] = dKV^T c over the b <= 16
// samples of a batch) is an outer-product sum -- 8 MB of output for 33 MFLOP.  On the MFMA kernel above 128 workgroups computed
// 128 x 128 tiles over four k-pairs and stored them as 64 scattered dwords per lane (21.5 us); here a thread keeps its column of
// B (K values) in registers, takes 8 rows of A as wave-uniform scalars and writes 8 coalesced row segments: one pass at the
// store rate.  The bias gradient (column sums of A) rides in the first column block.
// ------------------------------------------------------------------------------------------------
constexpr int TNS_KMAX = 16, TNS_ROWS = 8;
__global__ __launch_bounds__(256) void gemm_tn_smallk_kernel(const float *__restrict__ A, long lda, const float *__restrict__ B, long ldb,
                                                             float *__restrict__ C, long ldc, int M, int N, int K, float alpha,
                                                             int accumulate, float *__restrict__ colsum, int colsum_accumulate) {
  const int n = blockIdx.x * 256 + threadIdx.x, m0 = blockIdx.y * TNS_ROWS;
  float bv[TNS_KMAX];
#pragma unroll
  for (int k = 0; k < TNS_KMAX; ++k) bv[k] = (k < K && n < N) ? B[(long)k * ldb + n] : 0.0f;
  float acc[TNS_ROWS], old[TNS_ROWS];
#pragma unroll
  for (int r = 0; r < TNS_ROWS; ++r) {
    const int m = min(m0 + r, M - 1);
    old[r] = (accumulate && n < N) ? C[(long)m * ldc + n] : 0.0f;      // all old values requested ahead of the stores
    float a_ = 0.0f, cs = 0.0f;
#pragma unroll
    for (int k = 0; k < TNS_KMAX; ++k) {
      const float a = k < K ? A[(long)k * lda + m] : 0.0f;      // wave-uniform: scalar loads
      a_ = fmaf(a, bv[k], a_);
      cs += a;
    }
    acc[r] = a_;
    if (colsum && blockIdx.x == 0 && threadIdx.x == 0 && m0 + r < M) colsum[m] = colsum_accumulate ? colsum[m] + cs : cs;
  }
#pragma unroll
  for (int r = 0; r < TNS_ROWS; ++r)
    if (m0 + r < M && n < N) C[(long)(m0 + r) * ldc + n] = old[r] + alpha * acc[r];
}

// ------------------------------------------------------------------------------------------------
// Long-slice form of the same product (the patch-bag G = dKV^T z: 1024 x 773, K = 32 768 rows per call): the direct kernel above
// is bound by what a wave can keep in flight (4 dword loads per 4 MFMAs, no sharing between the 4 waves of a tile).  Here a
// workgroup stages 16-row slabs of both operands through LDS with 16-byte loads (each element fetched once per tile instead
// of twice), double-buffered, one barrier per slab (32 MFMAs per wave between barriers).  LDS pitch 160 floats: the two
// half-waves of a 32x32x2 operand read rows k and k + 1, 160 mod 64 = 32 puts them on disjoint banks.
// Needs 16-byte aligned operands (lda, ldb multiples of 4 floats).
// ------------------------------------------------------------------------------------------------
constexpr int TNL_BK = 16, TNL_PITCH = 160;

typedef float TnlSlab[TNL_BK * TNL_PITCH];
__device__ __forceinline__ void gemm_tn_lds_body(GemmTnArgs g, int bx, int by, int bz, TnlSlab *As, TnlSlab *Bs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bm0 = bx * 128, bn0 = by * 128;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int z = bz % g.nsplit, bt = bz / g.nsplit;
  g.A += bt * g.strideA;
  g.B += bt * g.strideB;
  const int k_begin = z * g.kslice, k_end = min(g.K, k_begin + g.kslice);
  const int rows = k_end - k_begin;
  const int half = lane >> 5, col = lane & 31;
  const i32x4 ars = make_rsrc(g.A + (long)k_begin * g.lda, rows > 0 ? (unsigned)(((long)(rows - 1) * g.lda + g.M) * 4) : 0u);
  const i32x4 brs = make_rsrc(g.B + (long)k_begin * g.ldb, rows > 0 ? (unsigned)(((long)(rows - 1) * g.ldb + g.N) * 4) : 0u);
  // loader: thread -> slab rows lr and lr + 8, 4 consecutive columns lc .. lc + 3
  const int lr = tid >> 5, lc = (tid & 31) * 4;
  const int am = bm0 + lc, bn = bn0 + lc;
  const int aoff = (int)(((long)lr * g.lda + am) * 4), boff = (int)(((long)lr * g.ldb + bn) * 4);
  const int astep = (int)(8 * g.lda * 4), bstep = (int)(8 * g.ldb * 4);
  const int aslab = (int)(TNL_BK * g.lda * 4), bslab = (int)(TNL_BK * g.ldb * 4);
  // columns past M / N inside a row would read the next row: masked in registers (whole-float4 granularity is not enough)
  const float am0 = am + 0 < g.M ? 1.f : 0.f, am1 = am + 1 < g.M ? 1.f : 0.f, am2 = am + 2 < g.M ? 1.f : 0.f, am3 = am + 3 < g.M ? 1.f : 0.f;
  const float bn0m = bn + 0 < g.N ? 1.f : 0.f, bn1m = bn + 1 < g.N ? 1.f : 0.f, bn2m = bn + 2 < g.N ? 1.f : 0.f, bn3m = bn + 3 < g.N ? 1.f : 0.f;
  const int nslabs = (rows + TNL_BK - 1) / TNL_BK;

  f32x4 ra[2], rb[2];
  auto load = [&](int slab) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // rows past the slice: the descriptor's range check returns 0 (offsets stay below 2^31: checked by the launcher)
      // unconditional: a piece past M / N reads the next row (or 0 past the slice) and is zeroed by the edge masks below; a
      // per-lane "load or 0" becomes a branch around every load with a full wait behind it
      ra[h] = hn_buffer_load_x4(ars, aoff + h * astep + slab * aslab, 0, 0);
      rb[h] = hn_buffer_load_x4(brs, boff + h * bstep + slab * bslab, 0, 0);
    }
  };
  // bias gradient sum_k A[k, m]: the loader threads of the first column tile add up what they stage (8 VALU adds per slab) --
  // ones-vector MFMAs in one wave of the tile would make that wave, and through the barrier the whole workgroup, 1.5x slower
  const bool do_colsum = g.colsum != nullptr && bn0 == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  // The masks only matter in the last row / column tile, the column sum only in the first column tile: each behind a scalar
  // branch (kept one by the empty asm), so that the interior tiles store what they loaded -- vector instructions in this loop
  // are matrix time lost (the fp32 MFMA shares the SIMD's issue with the VALU)
  const bool edge_a = bm0 + 128 > g.M, edge_b = bn0 + 128 > g.N;
  auto store = [&](int buf) {
    if (edge_a) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int h = 0; h < 2; ++h) ra[h] = (f32x4){ra[h][0] * am0, ra[h][1] * am1, ra[h][2] * am2, ra[h][3] * am3};
    }
    if (edge_b) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int h = 0; h < 2; ++h) rb[h] = (f32x4){rb[h][0] * bn0m, rb[h][1] * bn1m, rb[h][2] * bn2m, rb[h][3] * bn3m};
    }
    if (do_colsum) {
      asm volatile("" ::: "memory");
      csum += ra[0];
      csum += ra[1];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *(f32x4 *)&As[buf][(lr + 8 * h) * TNL_PITCH + lc] = ra[h];
      *(f32x4 *)&Bs[buf][(lr + 8 * h) * TNL_PITCH + lc] = rb[h];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool wave_live = bm0 + wm < g.M && bn0 + wn < g.N;
  const bool col1_live = bn0 + wn + 32 < g.N;

  if (nslabs > 0) { load(0); store(0); }
  __syncthreads();
  for (int sl = 0; sl < nslabs; ++sl) {
    const int buf = sl & 1;
    if (sl + 1 < nslabs) load(sl + 1);
    if (wave_live) {
      const float *ap = &As[buf][half * TNL_PITCH + wm + col], *bp = &Bs[buf][half * TNL_PITCH + wn + col];
      // two fragment sets: the operands of k-pair kk + 2 are requested before the four MFMAs of pair kk (pinned: left alone
      // the reads sit right in front of their MFMAs and every group of four starts with an exposed LDS round trip)
      float fa[2][2], fb[2][2];
      fa[0][0] = ap[0]; fa[0][1] = ap[32]; fb[0][0] = bp[0]; fb[0][1] = bp[32];
#pragma unroll
      for (int kk = 0; kk < TNL_BK; kk += 2) {
        const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
        if (kk + 2 < TNL_BK) {
          fa[nxt][0] = ap[(kk + 2) * TNL_PITCH]; fa[nxt][1] = ap[(kk + 2) * TNL_PITCH + 32];
          fb[nxt][0] = bp[(kk + 2) * TNL_PITCH]; fb[nxt][1] = bp[(kk + 2) * TNL_PITCH + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
        if (col1_live) {                      // wave-uniform: the second 32 columns of a ragged last column tile (773 = 6 * 128 + 5) are empty
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
        }
      }
    }
    if (sl + 1 < nslabs) store(buf ^ 1);
    __syncthreads();
  }
  if (do_colsum) {     // 8 loader rows x 128 columns of partial sums -> LDS (the slab buffers are free after the last barrier)
    *(f32x4 *)&As[0][lr * TNL_PITCH + lc] = csum;
    __syncthreads();
    if (tid < 128 && bm0 + tid < g.M) {
      float v = 0.0f;
#pragma unroll
      for (int r = 0; r < 8; ++r) v += As[0][r * TNL_PITCH + tid];
      float *cs = g.colsum + (g.nsplit > 1 ? (long)z * g.M : 0);
      const int m = bm0 + tid;
      if (g.nsplit > 1) cs[m] = v;
      else cs[m] = g.colsum_accumulate ? cs[m] + v : v;
    }
  }
  if (!wave_live) return;

  const int m0 = bm0 + wm, n0 = bn0 + wn;
  float *C = g.C + (g.nsplit > 1 ? ((long)bt * g.nsplit + z) * g.M * g.ldc : bt * g.strideC);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + 32 * j + col;
      if (n >= g.N) continue;
      float old[16];                                  // (accumulate: loads first, see gemm_tn_kernel)
      const bool rmw = g.nsplit == 1 && g.accumulate;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
        old[r] = (rmw && m < g.M) ? C[(long)m * g.ldc + n] : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < g.M) C[(long)m * g.ldc + n] = g.nsplit > 1 ? acc[i][j][r] : old[r] + g.alpha * acc[i][j][r];
      }
    }
}

__global__ __launch_bounds__(256) void gemm_tn_lds_kernel(GemmTnArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][TNL_BK * TNL_PITCH];
  __shared__ __attribute__((aligned(16))) float Bs[2][TNL_BK * TNL_PITCH];
  gemm_tn_lds_body(g, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// Several weight-gradient products of ONE contraction length in one launch (the fused latent backward, bchain.hip: dW1, dW2,
// dW_out, dW_q, dW_kv of a chain all contract over the same b * l_c rows): blockIdx.x walks the 128 x 128 tiles of all products,
// blockIdx.y the k-slices.  Partials (and the fused column-sum partials) go to scratch; splitk_reduce_multi_kernel folds them in
// a fixed order (bitwise reproducible).
__global__ __launch_bounds__(256) void gemm_tn_lds_multi_kernel(GemmTnMulti mm) {
  __shared__ __attribute__((aligned(16))) float As[2][TNL_BK * TNL_PITCH];
  __shared__ __attribute__((aligned(16))) float Bs[2][TNL_BK * TNL_PITCH];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < TN_MULTI_MAX; ++i) pi += (i < mm.n && (int)blockIdx.x >= mm.tile0[i]) ? 1 : 0;
  const TnProduct &pr = mm.p[pi];
  const int t = blockIdx.x - mm.tile0[pi], tm = (pr.M + 127) / 128;
  GemmTnArgs g;
  g.A = pr.A; g.lda = pr.lda; g.B = pr.B; g.ldb = pr.ldb;
  g.C = mm.scratch + pr.part_off; g.ldc = pr.N;
  g.M = pr.M; g.N = pr.N; g.K = mm.K; g.kslice = mm.kslice; g.nsplit = mm.nsplit;
  g.alpha = 1.0f; g.accumulate = 0;
  g.colsum = pr.colsum ? mm.scratch + pr.cs_off : nullptr; g.colsum_accumulate = 0;
  g.batch = 1; g.strideA = g.strideB = g.strideC = 0;
  gemm_tn_lds_body(g, t % tm, t / tm, blockIdx.y, As, Bs);
}

// blockIdx.y: product (entries n .. n + n_ln - 1: LayerNorm partial sums (nwg, width) -> out[width], same launch).  Entries that
// accumulate into the same destination form a chain (TnProduct.next): the head's blocks fold every member, in launch order.
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(GemmTnMulti mm) {
  const int pi = blockIdx.y;
  if (pi >= mm.n) {                                     // column sums of per-workgroup LayerNorm partials
    const LnPartial &l0 = mm.ln[pi - mm.n];
    if (l0.follower) return;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < l0.width; c += gridDim.x * blockDim.x) {
      float v = l0.out[c];
      for (int e = pi - mm.n; e >= 0; e = mm.ln[e].next) {
        const LnPartial &lp = mm.ln[e];
        // ONE workgroup walks the nwg = rows / 16 partial rows of an entry: sixteen loads in flight per trip (four were 64 dependent
        // trips at cfg2 b = 32 -- this loop, not the products' partials, was the launch's 70 us)
        float a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = 0.0f;
        int k = 0;
        for (; k + 16 <= lp.nwg; k += 16) {
          float t[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) t[u] = lp.part[(long)(k + u) * lp.stride + c];
#pragma unroll
          for (int u = 0; u < 16; ++u) a[u] += t[u];
        }
        for (; k < lp.nwg; ++k) a[0] += lp.part[(long)k * lp.stride + c];
        v += (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
      }
      l0.out[c] = v;
    }
    return;
  }
  const TnProduct &p0 = mm.p[pi];
  if (p0.follower) return;
  const long mn = (long)p0.M * p0.N, total = mn + (p0.colsum ? p0.M : 0);
  const int nsplit = mm.nsplit;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i >= mn) {
      const int m = (int)(i - mn);
      float v = p0.colsum[m];
      for (int e = pi; e >= 0; e = mm.p[e].next) {
        const float *cs_part = mm.scratch + mm.p[e].cs_off;
        float acc = 0.0f;
        for (int k = 0; k < nsplit; ++k) acc += cs_part[(long)k * p0.M + m];
        v += acc;
      }
      p0.colsum[m] = v;
      continue;
    }
    float *dst = p0.C + (i / p0.N) * p0.ldc + (i % p0.N);
    float v = *dst;
    for (int e = pi; e >= 0; e = mm.p[e].next) {
      const float *part = mm.scratch + mm.p[e].part_off;
      float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
      int k = 0;
      for (; k + 4 <= nsplit; k += 4) {
        a0 += part[(long)k * mn + i]; a1 += part[(long)(k + 1) * mn + i];
        a2 += part[(long)(k + 2) * mn + i]; a3 += part[(long)(k + 3) * mn + i];
      }
      for (; k < nsplit; ++k) a0 += part[(long)k * mn + i];
      v += (a0 + a1) + (a2 + a3);
    }
    *dst = v;
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_alpha_kernel(const float *__restrict__ part, int nsplit, long mn, int N,
                                                                  float *__restrict__ C, long ldc, float alpha, int accumulate,
                                                                  long strideC, const float *__restrict__ cs_part, int M,
                                                                  float *__restrict__ cs_out, int cs_accumulate) {
  part += (long)blockIdx.y * nsplit * mn;          // blockIdx.y: batch entry
  C += blockIdx.y * strideC;
  // the fused column-sum partials (nsplit, M) are folded by the same launch: elements mn .. mn + M - 1 of the index space
  const long total = mn + (cs_part ? M : 0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i >= mn) {
      const int m = (int)(i - mn);
      float acc = 0.0f;
      for (int k = 0; k < nsplit; ++k) acc += cs_part[(long)k * M + m];
      cs_out[m] = cs_accumulate ? cs_out[m] + acc : acc;
      continue;
    }
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;     // four independent chains (loads in flight), fixed order -> deterministic
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {
      a0 += part[(long)k * mn + i]; a1 += part[(long)(k + 1) * mn + i];
      a2 += part[(long)(k + 2) * mn + i]; a3 += part[(long)(k + 3) * mn + i];
    }
    for (; k < nsplit; ++k) a0 += part[(long)k * mn + i];
    const float acc = (a0 + a1) + (a2 + a3);
    float *dst = C + (i / N) * ldc + (i % N);
    *dst = accumulate ? *dst + alpha * acc : alpha * acc;
  }
}

constexpr long TN_SCRATCH_MIN_FLOATS = 6L << 20;      // 24 MB: floor of every split-k scratch buffer (reduce_scratch_floats)
// Many slices of a small output (launch_gemm_tn: up to 128): one thread per element walking all slices is a chain of ~30 dependent
// loads on a hundred workgroups.  Here a workgroup takes 64 elements and its four waves a quarter of the slices each (two
// independent chains per wave), folded through LDS in a fixed order: deterministic, 4x the loads in flight.
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const float *__restrict__ part, int nsplit, long mn, int N,
                                                                 float *__restrict__ C, long ldc, float alpha, int accumulate,
                                                                 const float *__restrict__ cs_part, int M, float *__restrict__ cs_out,
                                                                 int cs_accumulate) {
  __shared__ float red[4][64];
  const int li = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long total = mn + (cs_part ? M : 0);
  const long i = (long)blockIdx.x * 64 + li;
  float c0 = 0.0f, c1 = 0.0f;
  if (i < total) {
    const bool is_cs = i >= mn;
    const float *src = is_cs ? cs_part + (i - mn) : part + i;
    const long stride = is_cs ? M : mn;
    int k = w;
    for (; k + 4 < nsplit; k += 8) { c0 += src[(long)k * stride]; c1 += src[(long)(k + 4) * stride]; }
    if (k < nsplit) c0 += src[(long)k * stride];
  }
  red[w][li] = c0 + c1;
  __syncthreads();
  if (w == 0 && i < total) {
    const float acc = (red[0][li] + red[1][li]) + (red[2][li] + red[3][li]);
    if (i >= mn) {
      const int m = (int)(i - mn);
      cs_out[m] = cs_accumulate ? cs_out[m] + acc : acc;
    } else {
      float *dst = C + (i / N) * ldc + (i % N);
      *dst = accumulate ? *dst + alpha * acc : alpha * acc;
    }
  }
}

static int launch_gemm_tn(const float *A, long lda, const float *B, long ldb, float *C, long ldc, int M, int N, int K, float alpha,
                          int accumulate, float *scratch, hipStream_t s, float *colsum = nullptr, int colsum_accumulate = 0,
                          int batch = 1, long strideA = 0, long strideB = 0, long strideC = 0) {
  HN_REQUIRE(batch == 1 || colsum == nullptr, HN_E_UNSUPPORTED, "gemm_tn: the fused column sum is not batched");
  // long contraction, wide output (the patch-bag G = dKV^T z): the LDS-DMA kernel (gemm_nt.hip).  Every split-k scratch buffer holds
  // at least GEMM_EX_SPLITS * (M N + M) floats (reduce_scratch_floats)
  static const bool no_glds = tuning_env("HN_NO_GLDS_GEMM") != nullptr;      // development switch: the round-3 kernels
  if (!no_glds && scratch && batch == 1 && gemm_tn_glds_eligible(A, lda, B, ldb, M, N, K)) {
    size_t cap = (size_t)GEMM_EX_SPLITS * ((size_t)M * N + M);
    if (cap < (size_t)TN_SCRATCH_MIN_FLOATS) cap = (size_t)TN_SCRATCH_MIN_FLOATS;
    return launch_gemm_tn_glds(A, lda, B, ldb, C, ldc, M, N, K, alpha, accumulate, scratch, cap, colsum, colsum_accumulate, s);
  }
  if (K <= TNS_KMAX && batch == 1 && (long)M * N >= (1L << 14)) {      // outer-product sum over a handful of rows, wide output
    hipLaunchKernelGGL(gemm_tn_smallk_kernel, dim3(ceil_div(N, 256), ceil_div(M, TNS_ROWS)), dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, alpha,
                       accumulate, colsum, colsum_accumulate);
    HN_LAUNCH_CHECK("gemm_tn_smallk");
    return HN_OK;
  }
  const int tiles = ceil_div(M, 128) * ceil_div(N, 128) * batch;
  int nsplit = 1;
  if (scratch) {
    nsplit = 768 / tiles;                                       // 3 workgroups per CU resident (150 VGPRs): one full round,
                                                                // never a few leftovers in a second one (784 of 768 cost 40 %)
    const int max_by_k = ceil_div(K, 64);                       // at least 64 rows per slice
    if (nsplit > max_by_k) nsplit = max_by_k;
    // Small outputs over a long contraction (G = dKV^T z of ONE 16 .. 64-wide head over 32 768 patch rows: 32 x 773) have few tiles:
    // with 32 slices 224 workgroups stream the 101 MB operand at 1.2 TB/s (89 us).  Every scratch buffer holds at least
    // TN_SCRATCH_MIN_FLOATS (reduce_scratch_floats), so such a product may take up to 128 slices -- a full round of workgroups.
    int cap = GEMM_EX_SPLITS;
    const long per = ((long)M * N + (colsum ? M : 0)) * batch;
    if (batch == 1 && per * GEMM_EX_SPLITS < TN_SCRATCH_MIN_FLOATS) { const long c2 = TN_SCRATCH_MIN_FLOATS / per; cap = (int)(c2 < 128 ? c2 : 128); }
    if (cap < GEMM_EX_SPLITS) cap = GEMM_EX_SPLITS;
    if (nsplit > cap) nsplit = cap;
    if (nsplit < 1) nsplit = 1;
  }
  GemmTnArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb;
  g.M = M; g.N = N; g.K = K;
  g.kslice = ceil_div(ceil_div(K, nsplit), 2) * 2;
  g.nsplit = ceil_div(K, g.kslice);
  g.alpha = alpha; g.accumulate = accumulate;
  g.colsum = colsum; g.colsum_accumulate = colsum_accumulate;
  g.batch = batch; g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
  HN_REQUIRE((long)g.nsplit * batch <= 65535, HN_E_UNSUPPORTED, "gemm_tn: batch=%d", batch);
  if (g.nsplit > 1) { g.C = scratch; g.ldc = N; if (colsum) g.colsum = scratch + (size_t)g.nsplit * M * N; } else { g.C = C; g.ldc = ldc; }
  const bool aligned = (lda & 3) == 0 && (ldb & 3) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && (strideA & 3) == 0 &&
                       (strideB & 3) == 0;
  // (from 32 x 32 outputs: the LDS-staged kernel masks its ragged tiles; G = dKV^T z of a single 16 .. 103-wide head ran 142 us on
  // the direct kernel)
  if (g.kslice >= 64 && aligned && M >= 32 && N >= 32)
    hipLaunchKernelGGL(gemm_tn_lds_kernel, dim3(ceil_div(M, 128), ceil_div(N, 128), g.nsplit * batch), dim3(256), 0, s, g);
  else if (g.kslice <= 512) hipLaunchKernelGGL(gemm_tn_kernel<8>, dim3(ceil_div(M, 128), ceil_div(N, 128), g.nsplit * batch), dim3(256), 0, s, g);
  else hipLaunchKernelGGL(gemm_tn_kernel<4>, dim3(ceil_div(M, 128), ceil_div(N, 128), g.nsplit * batch), dim3(256), 0, s, g);
  HN_LAUNCH_CHECK("gemm_tn");
  if (g.nsplit > 1) {
    const long mn = (long)M * N;
    if (g.nsplit > GEMM_EX_SPLITS && batch == 1) {
      hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)ceil_div_ll(mn + (colsum ? M : 0), 64)), dim3(256), 0, s, scratch, g.nsplit, mn, N,
                         C, ldc, alpha, accumulate, colsum ? g.colsum : nullptr, M, colsum, colsum_accumulate);
      HN_LAUNCH_CHECK("splitk_reduce_wide");
      return HN_OK;
    }
    long blocks = ceil_div_ll(mn + (colsum ? M : 0), 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_alpha_kernel, dim3((unsigned)blocks, batch), dim3(256), 0, s, scratch, g.nsplit, mn, N, C, ldc, alpha,
                       accumulate, strideC, colsum ? g.colsum : nullptr, M, colsum, colsum_accumulate);
    HN_LAUNCH_CHECK("splitk_reduce");
  }
  return HN_OK;
}

// Plans the shared split of a multi-product launch (one full round of resident workgroups, >= 64 rows per slice) and the scratch
// layout of the partials.
static void plan_tn_multi(GemmTnMulti &m) {
  int tiles = 0;
  for (int i = 0; i < m.n; ++i) { m.tile0[i] = tiles; tiles += ceil_div(m.p[i].M, 128) * ceil_div(m.p[i].N, 128); }
  m.tile0[m.n] = tiles;
  static const int slots = tuning_env("HN_TN_MULTI_SLOTS") ? atoi(tuning_env("HN_TN_MULTI_SLOTS")) : 768;      // development knob
  int nsplit = tiles > 0 ? slots / tiles : 1;
  const int max_by_k = ceil_div(m.K, 64);
  if (nsplit > max_by_k) nsplit = max_by_k;
  if (nsplit > GEMM_EX_SPLITS) nsplit = GEMM_EX_SPLITS;
  if (nsplit < 1) nsplit = 1;
  m.kslice = ceil_div(ceil_div(m.K, nsplit), 2) * 2;
  m.nsplit = ceil_div(m.K, m.kslice);
  long off = 0;
  for (int i = 0; i < m.n; ++i) {
    m.p[i].part_off = off; off += (long)m.nsplit * m.p[i].M * m.p[i].N;
    m.p[i].cs_off = off; if (m.p[i].colsum) off += (long)m.nsplit * m.p[i].M;
    off = (off + 63) / 64 * 64;
  }
  m.scratch = nullptr;
  m.p[0].part_off += 0;
  m.tile0[m.n] = tiles;
  (void)off;
}

size_t gemm_tn_multi_scratch_floats(const GemmTnMulti &mm) {
  GemmTnMulti m = mm;
  plan_tn_multi(m);
  long off = 0;
  for (int i = 0; i < m.n; ++i) {
    off = m.p[i].cs_off + (m.p[i].colsum ? (long)m.nsplit * m.p[i].M : 0);
    off = (off + 63) / 64 * 64;
  }
  return (size_t)off;
}

int launch_gemm_tn_multi(GemmTnMulti &m, float *scratch, size_t scratch_floats, hipStream_t s) {
  HN_REQUIRE(m.n >= 0 && m.n <= TN_MULTI_MAX && m.n_ln >= 0 && m.n_ln <= TN_MULTI_LN_MAX && m.K > 0, HN_E_SHAPE, "gemm_tn_multi: n=%d n_ln=%d K=%d", m.n, m.n_ln, m.K);
  // destinations that occur more than once: one reduce pass per destination, members folded in launch order
  for (int i = 0; i < m.n; ++i) { m.p[i].next = -1; m.p[i].follower = 0; }
  for (int i = 0; i < m.n; ++i)
    for (int j = i - 1; j >= 0; --j)
      if (m.p[j].C == m.p[i].C) {
        HN_REQUIRE(m.p[j].M == m.p[i].M && m.p[j].N == m.p[i].N && m.p[j].ldc == m.p[i].ldc && m.p[j].colsum == m.p[i].colsum, HN_E_SHAPE,
                   "gemm_tn_multi: products %d and %d share a destination but not its shape", j, i);
        m.p[j].next = i; m.p[i].follower = 1;
        break;
      }
  for (int i = 0; i < m.n_ln; ++i) { m.ln[i].next = -1; m.ln[i].follower = 0; }
  for (int i = 0; i < m.n_ln; ++i)
    for (int j = i - 1; j >= 0; --j)
      if (m.ln[j].out == m.ln[i].out) {
        HN_REQUIRE(m.ln[j].width == m.ln[i].width, HN_E_SHAPE, "gemm_tn_multi: LayerNorm entries %d and %d share a destination but not its width", j, i);
        m.ln[j].next = i; m.ln[i].follower = 1;
        break;
      }
  for (int i = 0; i < m.n; ++i) {
    const TnProduct &p = m.p[i];
    HN_REQUIRE(p.A && p.B && p.C && p.M >= 16 && p.N >= 16 && (p.lda & 3) == 0 && (p.ldb & 3) == 0 && ((uintptr_t)p.A & 15) == 0 &&
                   ((uintptr_t)p.B & 15) == 0, HN_E_SHAPE, "gemm_tn_multi: product %d M=%d N=%d", i, p.M, p.N);
    HN_REQUIRE(((long)m.K * p.lda + p.M) * 4 < (1L << 31) && ((long)m.K * p.ldb + p.N) * 4 < (1L << 31), HN_E_UNSUPPORTED, "gemm_tn_multi: operand too large");
  }
  if (m.n > 0) {
    plan_tn_multi(m);
    HN_REQUIRE(scratch && scratch_floats >= gemm_tn_multi_scratch_floats(m), HN_E_WORKSPACE, "gemm_tn_multi: scratch too small");
    m.scratch = scratch;
    hipLaunchKernelGGL(gemm_tn_lds_multi_kernel, dim3(m.tile0[m.n], m.nsplit), dim3(256), 0, s, m);
    HN_LAUNCH_CHECK("gemm_tn_multi");
  }
  if (m.n + m.n_ln > 0) {
    long biggest = 128;
    for (int i = 0; i < m.n; ++i) { const long e = (long)m.p[i].M * m.p[i].N + m.p[i].M; if (e > biggest) biggest = e; }
    long blocks = ceil_div_ll(biggest, 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3((unsigned)blocks, m.n + m.n_ln), dim3(256), 0, s, m);
    HN_LAUNCH_CHECK("splitk_reduce_multi");
  }
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// NN product  C[m, n] (+)= sum_k A[m, k] * B[k, n]  (dX = dY W with W in nn.Linear layout): W is transposed into
// scratch (a few hundred KB, one tiny launch) and the product runs on the forward path's tuned NT kernels
// (gemm.hip) instead of the generic strided kernel: 2 launches of ~4 + ~12 us instead of one of 30-50 us.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src, long ld, int rows, int cols,
                                                        float *__restrict__ dst) {       // dst (cols, rows)
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(long)r * ld + c] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < cols && r < rows) dst[(long)c * rows + r] = tile[tx][ty + 8 * i];
  }
}

// ---- transposed-weight cache of one backward pass: hn_fusion_backward registers every weight its dX products will need, ONE
// batched launch per 16 of them transposes them all up front, launch_gemm_nn then finds them here instead of running a
// transpose launch in front of every product (48 per step at cfg2).  Thread-local, valid between begin and end only.
constexpr int TC_MAX = 512, TC_BATCH = 64;      // (40 bytes per entry: 2.5 KB of kernel arguments; a depth-3 model registers 37 weights --
                                                 // three launches of 5-10 us at 16 per batch, round 6: one)
struct TransposeEntry { const float *src; long ld; int rows, cols; float *dst; };
struct TransposeBatch { int n; TransposeEntry e[TC_BATCH]; };
struct TransposeCache { int n; bool ready; TransposeEntry e[TC_MAX]; };
static thread_local TransposeCache g_tc = {0, false, {}};

void transpose_cache_begin() { g_tc.n = 0; g_tc.ready = false; }
void transpose_cache_end() { g_tc.n = 0; g_tc.ready = false; }
void transpose_cache_add(const float *src, long ld, int rows, int cols) {
  if (!src || g_tc.n >= TC_MAX) return;
  for (int i = 0; i < g_tc.n; ++i)
    if (g_tc.e[i].src == src && g_tc.e[i].ld == ld && g_tc.e[i].rows == rows && g_tc.e[i].cols == cols) return;   // tied weights
  g_tc.e[g_tc.n++] = {src, ld, rows, cols, nullptr};
}
size_t transpose_cache_floats() {
  size_t n = 0;
  for (int i = 0; i < g_tc.n; ++i) n += align_up((size_t)g_tc.e[i].rows * g_tc.e[i].cols, 64);
  return n;
}

__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeBatch tb) {
  const TransposeEntry &t = tb.e[blockIdx.z];
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  if (r0 >= t.rows || c0 >= t.cols) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < t.rows && c < t.cols) ? t.src[(long)r * t.ld + c] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < t.cols && r < t.rows) t.dst[(long)c * t.rows + r] = tile[tx][ty + 8 * i];
  }
}

int transpose_cache_run(float *buf, size_t buf_floats, hipStream_t s) {
  HN_REQUIRE(buf_floats >= transpose_cache_floats(), HN_E_WORKSPACE, "transpose cache: buffer too small");
  size_t off = 0;
  for (int i = 0; i < g_tc.n; ++i) { g_tc.e[i].dst = buf + off; off += align_up((size_t)g_tc.e[i].rows * g_tc.e[i].cols, 64); }
  for (int i0 = 0; i0 < g_tc.n; i0 += TC_BATCH) {
    TransposeBatch tb;
    tb.n = min(TC_BATCH, g_tc.n - i0);
    int maxr = 1, maxc = 1;
    for (int i = 0; i < tb.n; ++i) { tb.e[i] = g_tc.e[i0 + i]; maxr = max(maxr, tb.e[i].rows); maxc = max(maxc, tb.e[i].cols); }
    for (int i = tb.n; i < TC_BATCH; ++i) tb.e[i] = {nullptr, 0, 0, 0, nullptr};
    hipLaunchKernelGGL(transpose_multi_kernel, dim3(ceil_div(maxc, 32), ceil_div(maxr, 32), tb.n), dim3(256), 0, s, tb);
    HN_LAUNCH_CHECK("transpose_multi");
  }
  g_tc.ready = true;
  return HN_OK;
}

const float *transpose_cache_lookup(const float *src, long ld, int rows, int cols) {
  if (!g_tc.ready) return nullptr;
  for (int i = 0; i < g_tc.n; ++i)
    if (g_tc.e[i].src == src && g_tc.e[i].ld == ld && g_tc.e[i].rows == rows && g_tc.e[i].cols == cols) return g_tc.e[i].dst;
  return nullptr;
}

static int launch_gemm_nn(const GemmExArgs &g, hipStream_t s, float *scratch) {
  // B(j, c) = B + j + c * b_cs: a (K x N) row-major matrix of pitch b_cs
  const float *wt = transpose_cache_lookup(g.B, g.b_cs, g.K, g.N);
  if (wt == nullptr) {
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(g.N, 32), ceil_div(g.K, 32)), dim3(256), 0, s, g.B, g.b_cs, g.K, g.N, scratch);
    HN_LAUNCH_CHECK("transpose");
    wt = scratch;
  }
  GemmArgs f = {};
  f.batch = 1; f.eps = 1e-5f;
  f.A = g.A; f.lda = g.a_rs;
  f.W = wt; f.ldw = g.K;
  f.C = g.C; f.ldc = g.ldc;
  f.M = g.M; f.N = g.N; f.K = g.K;
  f.alpha = g.alpha;
  if (g.accumulate) { f.R = g.C; f.ldr = g.ldc; }
  return launch_gemm(f, s);
}

// Other weight-gradient shaped products (head-batched, strided): small M x N, long contraction; with `scratch`
// (GEMM_EX_SPLITS * M * N floats) the contraction is cut into slices that run as extra grid.z entries.
int launch_gemm_ex(const GemmExArgs &g, hipStream_t s, float *scratch) {
  HN_REQUIRE(g.A && g.B && g.C, HN_E_NULL, "gemm_ex: NULL operand");
  HN_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.batch > 0, HN_E_SHAPE, "gemm_ex: M=%d N=%d K=%d", g.M, g.N, g.K);
  static int no_tn = -1, no_nn = -1;      // development knobs (flakiness bisect)
  if (no_tn < 0) { const char *e = getenv("HN_NO_TN"); no_tn = (e && e[0] == '1') ? 1 : 0; e = tuning_env("HN_NO_NN"); no_nn = (e && e[0] == '1') ? 1 : 0; }
  // TN form (both operands contraction-major, unit stride along their own row index): the MFMA-native kernel
  if (!no_tn && (g.batch == 1 || (!g.colsum && scratch)) && g.a_rs == 1 && g.b_rs == 1 && g.k_total == 0 &&
      (long)g.K * g.a_cs * 4 < (1L << 31) && (long)g.K * g.b_cs * 4 < (1L << 31))
    return launch_gemm_tn(g.A, g.a_cs, g.B, g.b_cs, g.C, g.ldc, g.M, g.N, g.K, g.alpha, g.accumulate, scratch, s, g.colsum, g.colsum_accumulate,
                          g.batch, g.strideA, g.strideB, g.strideC);
  if (g.colsum) {      // not the TN route: the caller's column sum still has to happen
    int rc = launch_colsum(g.A, g.a_cs, g.K, g.M, 1.0f, g.colsum, g.colsum_accumulate, s, scratch);
    if (rc != HN_OK) return rc;
  }
  // NN form (A row-major over the contraction, B contraction-major): dX = dY W
  if (!no_nn && scratch && g.batch == 1 && g.a_cs == 1 && g.b_rs == 1 && g.k_total == 0 && g.M >= 256 && (g.K & 3) == 0 && (g.a_rs & 3) == 0)
    return launch_gemm_nn(g, s, scratch);
  const int tiles = ceil_div(g.M, XM) * ceil_div(g.N, XN) * g.batch;
  if (scratch && g.batch == 1 && g.K >= 2048 && tiles <= 128) {
    GemmExArgs p = g;
    const int ksl = ceil_div(ceil_div(g.K, GEMM_EX_SPLITS), XK) * XK;       // slice length, multiple of the k-tile
    const int nsl = ceil_div(g.K, ksl);
    p.batch = nsl;
    p.K = ksl;                                   // the last slice is cut by k_total below
    p.k_total = g.K;
    p.strideA = (long)ksl * g.a_cs;
    p.strideB = (long)ksl * g.b_cs;
    p.C = scratch; p.ldc = g.N; p.strideC = (long)g.M * g.N; p.accumulate = 0;
    dim3 grid(ceil_div(g.M, XM), ceil_div(g.N, XN), nsl);
    hipLaunchKernelGGL(gemm_ex_kernel, grid, dim3(256), 0, s, p);
    HN_LAUNCH_CHECK("gemm_ex(split-k)");
    const long mn = (long)g.M * g.N;
    long blocks = ceil_div_ll(mn, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, scratch, nsl, mn, g.N, g.C, g.ldc, g.accumulate);
    HN_LAUNCH_CHECK("splitk_reduce");
    return HN_OK;
  }
  dim3 grid(ceil_div(g.M, XM), ceil_div(g.N, XN), g.batch);
  HN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, HN_E_UNSUPPORTED, "gemm_ex: grid too large");
  hipLaunchKernelGGL(gemm_ex_kernel, grid, dim3(256), 0, s, g);
  HN_LAUNCH_CHECK("gemm_ex");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// out[n] (+)= scale * sum_m X[m, n]   (fixed summation order: deterministic).  Tall inputs are reduced in two
// stages through `scratch` (COLSUM_CHUNKS x cols floats) so that the row range is spread over the chip.
// ------------------------------------------------------------------------------------------------
constexpr int COLSUM_CHUNKS = 128;

__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ X, long ld, long rows, int cols, float scale,
                                                     float *__restrict__ out, long out_pitch, int accumulate) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const long chunk = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
  float s = 0.0f;
  if (c < cols)
    for (long r = r0 + w; r < r1; r += 4) s += X[r * ld + c];
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < cols) {
    const float v = scale * (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
    float *dst = out + (long)blockIdx.y * out_pitch + c;
    *dst = accumulate ? *dst + v : v;
  }
}

// Narrow matrices (row pitch 16 or 32 floats: the per-head channel rows of the shared-context backward, 13 or 18 valid columns):
// the wide kernel above keeps 13 of 64 lanes busy, 64 bytes apart (17 us for 32768 x 16 at cfg2 b = 32, six times per image block).
// Here a wave covers 64 / ld whole rows per trip -- contiguous 256 bytes -- and the row groups are folded in LDS in a fixed order.
template <int LD>
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float *__restrict__ X, long rows, int cols, float scale,
                                                            float *__restrict__ out, long out_pitch, int accumulate) {
  constexpr int RPW = 64 / LD;                       // rows per wave and trip
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = lane % LD, rs = lane / LD;
  const long chunk = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  long r = r0 + w * RPW + rs;
  for (; r + 12 * RPW < r1; r += 16 * RPW) {
    s0 += X[r * LD + c]; s1 += X[(r + 4 * RPW) * LD + c];
    s2 += X[(r + 8 * RPW) * LD + c]; s3 += X[(r + 12 * RPW) * LD + c];
  }
  for (; r < r1; r += 4 * RPW) s0 += X[r * LD + c];
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < LD && (int)threadIdx.x < cols) {
    float v = 0.0f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww)
#pragma unroll
      for (int g = 0; g < RPW; ++g) v += part[ww][g * LD + threadIdx.x];
    float *dst = out + (long)blockIdx.y * out_pitch + threadIdx.x;
    *dst = accumulate ? *dst + scale * v : scale * v;
  }
}

int launch_colsum(const float *X, long ld, long rows, int cols, float scale, float *out, int accumulate, hipStream_t s,
                  float *scratch) {
  HN_REQUIRE(X && out && rows > 0 && cols > 0, HN_E_SHAPE, "colsum: rows=%ld cols=%d", rows, cols);
  static const bool no_narrow = getenv("HN_NO_NARROW_COLSUM") != nullptr;      // route switch (A/B)
  if (!no_narrow && (ld == 16 || ld == 32) && cols <= ld && rows >= 64) {
    const bool two = scratch != nullptr && rows >= 4096;
    const int chunks = two ? COLSUM_CHUNKS : 1;
    float *o1 = two ? scratch : out;
    if (ld == 16) hipLaunchKernelGGL(colsum_narrow_kernel<16>, dim3(1, chunks), dim3(256), 0, s, X, rows, cols, two ? 1.0f : scale, o1, two ? 16L : 0L, two ? 0 : accumulate);
    else hipLaunchKernelGGL(colsum_narrow_kernel<32>, dim3(1, chunks), dim3(256), 0, s, X, rows, cols, two ? 1.0f : scale, o1, two ? 32L : 0L, two ? 0 : accumulate);
    HN_LAUNCH_CHECK("colsum(narrow)");
    if (two) {      // the 128 chunk rows (pitch ld, columns >= cols never written / never read)
      if (ld == 16) hipLaunchKernelGGL(colsum_narrow_kernel<16>, dim3(1, 1), dim3(256), 0, s, scratch, (long)COLSUM_CHUNKS, cols, scale, out, 0L, accumulate);
      else hipLaunchKernelGGL(colsum_narrow_kernel<32>, dim3(1, 1), dim3(256), 0, s, scratch, (long)COLSUM_CHUNKS, cols, scale, out, 0L, accumulate);
      HN_LAUNCH_CHECK("colsum(narrow, stage 2)");
    }
    return HN_OK;
  }
  if (scratch && rows >= 4096) {
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(cols, 64), COLSUM_CHUNKS), dim3(256), 0, s, X, ld, rows, cols, 1.0f, scratch,
                       (long)cols, 0);
    HN_LAUNCH_CHECK("colsum(stage 1)");
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(cols, 64), 1), dim3(256), 0, s, scratch, (long)cols, (long)COLSUM_CHUNKS, cols,
                       scale, out, 0L, accumulate);
    HN_LAUNCH_CHECK("colsum(stage 2)");
    return HN_OK;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(cols, 64), 1), dim3(256), 0, s, X, ld, rows, cols, scale, out, 0L, accumulate);
  HN_LAUNCH_CHECK("colsum");
  return HN_OK;
}

size_t reduce_scratch_floats(long max_mn, int max_cols) {
  const size_t a = (size_t)GEMM_EX_SPLITS * (max_mn + max_cols), c = (size_t)COLSUM_CHUNKS * max_cols;   // split-k partials (+ colsum partials)
  // (kv_weight_grads needs 2 * KVG_CHUNKS * D floats: covered by a, since max_mn >= 2 * inner * D)
  const size_t m = a > c ? a : c;
  return m > (size_t)TN_SCRATCH_MIN_FLOATS ? m : (size_t)TN_SCRATCH_MIN_FLOATS;      // (launch_gemm_tn: up to 128 slices of small outputs)
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward.  y = (x - mu) * rs * gamma + beta per row of length d.
//   dx (+)= rs * (dyg - mean(dyg) - xn * mean(dyg * xn)),  dyg = dy * gamma,  xn = (x - mu) * rs
//   partial[block, 0:d] = sum_rows dy * xn ;  partial[block, d:2d] = sum_rows dy     (reduced by colsum)
// One wave per row (lanes along d), LN_ROWS rows per workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int LN_ROWS = 16, LN_MAXC = 16;   // d <= 64 * LN_MAXC

template <int MAXC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                     const float *__restrict__ gamma, float eps, long rows, int d,
                                                     float *__restrict__ dx, int dx_accumulate, float *__restrict__ partial) {
  extern __shared__ float red[];   // [4 waves][2 d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gsum[MAXC], bsum[MAXC], gam[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    gsum[k] = 0.0f; bsum[k] = 0.0f;
    gam[k] = lane + 64 * k < d ? gamma[lane + 64 * k] : 0.0f;
  }
  const long r0 = (long)blockIdx.x * LN_ROWS;
  const float invd = 1.0f / (float)d;
  for (int rr = wave; rr < LN_ROWS; rr += 4) {
    const long r = r0 + rr;
    if (r >= rows) break;
    const float *xr = x + r * d, *dyr = dy + r * d;
    float xv[MAXC], dv[MAXC];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + 64 * k;
      xv[k] = c < d ? xr[c] : 0.0f;
      dv[k] = c < d ? dyr[c] : 0.0f;
      s += xv[k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s * invd;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + 64 * k;
      const float t = c < d ? xv[k] - mu : 0.0f;
      q += t * t;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rs = 1.0f / sqrtf(q * invd + eps);
    float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < d) {
        const float xn = (xv[k] - mu) * rs;
        const float dyg = dv[k] * gam[k];
        gsum[k] += dv[k] * xn;
        bsum[k] += dv[k];
        m1 += dyg;
        m2 += dyg * xn;
        xv[k] = xn;
        dv[k] = dyg;
      }
    }
    for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
    m1 *= invd;
    m2 *= invd;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < d) {
        const float v = rs * (dv[k] - m1 - xv[k] * m2);
        dx[r * d + c] = dx_accumulate ? dx[r * d + c] + v : v;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < d) { red[wave * 2 * d + c] = gsum[k]; red[wave * 2 * d + d + c] = bsum[k]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * d; c += blockDim.x)
    partial[(long)blockIdx.x * 2 * d + c] = red[c] + red[2 * d + c] + red[4 * d + c] + red[6 * d + c];
}

// dgamma[c] += sum_blocks partial[., c] ; dbeta[c] += sum_blocks partial[., d + c]  -- one launch for both (fixed order)
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float *__restrict__ partial, int blocks, int d,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta) {
  __shared__ float part[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;      // 16 waves share the partial rows
  float s = 0.0f;
  if (c < 2 * d) {
#pragma unroll 4
    for (int r = w; r < blocks; r += 16) s += partial[(long)r * 2 * d + c];
  }
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < 2 * d) {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][threadIdx.x];
    if (c < d) { if (dgamma) dgamma[c] += v; }
    else if (dbeta) dbeta[c - d] += v;
  }
}

// dgamma / dbeta accumulate into their gradient buffers; `scratch` needs ln_bwd_scratch_floats(rows, d) floats
size_t ln_bwd_scratch_floats(long rows, int d) { return (size_t)ceil_div_ll(rows, LN_ROWS) * 2 * d; }

int launch_ln_bwd(const float *x, const float *dy, const float *gamma, long rows, int d, float *dx, int dx_accumulate,
                  float *dgamma, float *dbeta, float *scratch, hipStream_t s) {
  HN_REQUIRE(x && dy && gamma && dx && scratch, HN_E_NULL, "ln_bwd: NULL pointer");
  HN_REQUIRE(d > 0 && d <= 64 * LN_MAXC, HN_E_UNSUPPORTED, "ln_bwd: d=%d (<= %d supported)", d, 64 * LN_MAXC);
  const int blocks = (int)ceil_div_ll(rows, LN_ROWS);
  const size_t lds = (size_t)8 * d * sizeof(float);
#define HN_LN_BWD(C_) hipLaunchKernelGGL(ln_bwd_kernel<C_>, dim3(blocks), dim3(256), lds, s, x, dy, gamma, 1e-5f, rows, d, dx, dx_accumulate, scratch)
  if (d <= 128) HN_LN_BWD(2);
  else if (d <= 256) HN_LN_BWD(4);
  else if (d <= 512) HN_LN_BWD(8);
  else HN_LN_BWD(LN_MAXC);
#undef HN_LN_BWD
  HN_LAUNCH_CHECK("ln_bwd");
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(ceil_div(2 * d, 64)), dim3(1024), 0, s, scratch, blocks, d, dgamma, dbeta);
    HN_LAUNCH_CHECK("ln_param_reduce");
  }
  return HN_OK;
}

// Parameter gradients of a K/V projection applied to an affine-normalised context c_hat = z * gamma + beta, from
//   G = dKV^T z  (nrows, D)   and   cs = colsum(dKV)  (nrows):
//   dW[n, d] += G[n, d] * gamma[d] + cs[n] * beta[d];  dgamma[d] += sum_n W[n, d] G[n, d];  dbeta[d] += sum_n W[n, d] cs[n]
// (c_hat and its gradient are never materialised).  gamma == NULL: plain dW += G.
constexpr int KVG_CHUNKS = 32;

__global__ __launch_bounds__(256) void kv_weight_grads_kernel(const float *__restrict__ G, const float *__restrict__ cs,
                                                              const float *__restrict__ w, const float *gamma, const float *beta,
                                                              int nrows, int D, float *dw, float *__restrict__ partial) {
  __shared__ float pg[4][64], pb[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
  const int chunk = (nrows + gridDim.y - 1) / gridDim.y;
  const int n0 = blockIdx.y * chunk, n1 = min(nrows, n0 + chunk);
  float sg = 0.0f, sb = 0.0f;
  if (c < D) {
    const float gm = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
    // four rows per trip, every load (the old dW values too) requested before the first store
    for (int n = n0 + wv; n < n1; n += 16) {
      float gv[4], wv_[4], csn[4], od[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nn = min(n + 4 * u, n1 - 1);
        gv[u] = G[(long)nn * D + c]; wv_[u] = w[(long)nn * D + c]; csn[u] = cs[nn];
        od[u] = dw ? dw[(long)nn * D + c] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (n + 4 * u >= n1) break;
        if (dw) dw[(long)(n + 4 * u) * D + c] = od[u] + (gv[u] * gm + csn[u] * bt);
        sg += wv_[u] * gv[u];
        sb += wv_[u] * csn[u];
      }
    }
  }
  pg[wv][threadIdx.x & 63] = sg;
  pb[wv][threadIdx.x & 63] = sb;
  __syncthreads();
  if (wv == 0 && c < D && partial) {      // row-chunk partials of (dgamma, dbeta), summed in fixed order by the reduce kernel
    partial[((long)blockIdx.y * 2 + 0) * D + c] = pg[0][threadIdx.x] + pg[1][threadIdx.x] + pg[2][threadIdx.x] + pg[3][threadIdx.x];
    partial[((long)blockIdx.y * 2 + 1) * D + c] = pb[0][threadIdx.x] + pb[1][threadIdx.x] + pb[2][threadIdx.x] + pb[3][threadIdx.x];
  }
}

__global__ __launch_bounds__(256) void kv_affine_reduce_kernel(const float *__restrict__ partial, int nchunks, int D, float *dgamma, float *dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  float sg = 0.0f, sb = 0.0f;
  // (eight chunks = sixteen independent loads in flight per trip: the rolled loop was 32 dependent round trips, 11 us)
  int k = 0;
  for (; k + 8 <= nchunks; k += 8) {
    float g8[8], b8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { g8[u] = partial[((long)(k + u) * 2 + 0) * D + c]; b8[u] = partial[((long)(k + u) * 2 + 1) * D + c]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { sg += g8[u]; sb += b8[u]; }
  }
  for (; k < nchunks; ++k) { sg += partial[((long)k * 2 + 0) * D + c]; sb += partial[((long)k * 2 + 1) * D + c]; }
  if (dgamma) dgamma[c] += sg;
  if (dbeta) dbeta[c] += sb;
}

int launch_kv_weight_grads(const float *G, const float *cs, const float *w, const float *gamma, const float *beta, int nrows, int D,
                           float *dw, float *dgamma, float *dbeta, hipStream_t s, float *scratch) {
  HN_REQUIRE(scratch != nullptr, HN_E_NULL, "kv_weight_grads: scratch is NULL");
  const bool affine = gamma != nullptr && (dgamma != nullptr || dbeta != nullptr);
  hipLaunchKernelGGL(kv_weight_grads_kernel, dim3(ceil_div(D, 64), KVG_CHUNKS), dim3(256), 0, s, G, cs, w, gamma, beta, nrows, D, dw,
                     affine ? scratch : nullptr);
  HN_LAUNCH_CHECK("kv_weight_grads");
  if (affine) {
    hipLaunchKernelGGL(kv_affine_reduce_kernel, dim3(ceil_div(D, 256)), dim3(256), 0, s, scratch, KVG_CHUNKS, D, dgamma, dbeta);
    HN_LAUNCH_CHECK("kv_affine_reduce");
  }
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward of the one-token cross block (tabular / omic modality; healnet.py:236 with N = 1: y_b = LeakyReLU(W_out V_b + b_out) added
// to every latent row of sample b, V_b = (c_hat_b * gamma + beta) W_v^T) in FOUR launches instead of nine (round 5: at cfg4 b = 8 the
// nine were 58 us per layer of 4-10 us launches on 8 rows):
//   onetok_dyb_kernel          dyb = per-sample sum over the latent rows of dy * LeakyReLU'(.)   (was: leaky_bwd + segsum)
//   onetok_dv_kernel           dV = dyb W_out
//                              (a first version ran both in ONE workgroup per sample: 8 workgroups walking 128 rows and 128
//                              weight rows in dependent trips -- +0.13 ms per cfg4 step instead of a gain; these two keep the
//                              old kernels' parallelism)
//   onetok_bwd_weights_kernel  workgroup roles: [0, qd) dW_out[q, :] += sum_b dyb[b, q] V[b, :] and db_out[q] += sum_b dyb[b, q];
//                              the rest: dW_v += G * gamma + cs (x) beta with G = dV^T c_hat and cs = colsum(dV) formed on the fly
//                              from the b rows, and the row-chunk partials of dgamma / dbeta (was: two products, two column sums and
//                              kv_weight_grads_kernel)
//   kv_affine_reduce_kernel    as before.
// b <= ONETOK_MAX_B (the sample dimension is the contraction: its operands sit in registers).
// ------------------------------------------------------------------------------------------------
constexpr int ONETOK_MAX_B = 32;

// dyb[b, q] = sum over the L latent rows of sample b of dy * LeakyReLU'(y), y = x_out - x_in: 64 columns per workgroup, the four
// waves split the rows, four rows' loads in flight per trip, fixed-order LDS fold (segsum_kernel with the LeakyReLU factor fused in)
__global__ __launch_bounds__(256) void onetok_dyb_kernel(const float *__restrict__ dy, const float *__restrict__ x_out, const float *x_in,
                                                         int L, int qd, float *__restrict__ dyb) {
  __shared__ float part[4][64];
  const int bi = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + lane;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (q < qd) {
    const long base = (long)bi * L * qd + q;
    int r = w;
    for (; r + 12 < L; r += 16) {
      float d[4], xo[4], xi[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = base + (long)(r + 4 * u) * qd;
        d[u] = dy[i]; xo[u] = x_out[i]; xi[u] = x_in ? x_in[i] : 0.0f;
      }
      s0 += d[0] * ((xo[0] - xi[0]) > 0.0f ? 1.0f : 0.01f); s1 += d[1] * ((xo[1] - xi[1]) > 0.0f ? 1.0f : 0.01f);
      s2 += d[2] * ((xo[2] - xi[2]) > 0.0f ? 1.0f : 0.01f); s3 += d[3] * ((xo[3] - xi[3]) > 0.0f ? 1.0f : 0.01f);
    }
    for (; r < L; r += 4) {
      const long i = base + (long)r * qd;
      const float y = x_in ? x_out[i] - x_in[i] : x_out[i];
      s0 += dy[i] * (y > 0.0f ? 1.0f : 0.01f);
    }
  }
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && q < qd) dyb[(long)bi * qd + q] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// dV[b, i] = sum_q dyb[b, q] W_out[q, i]: 64 columns i per workgroup, the four waves split q, eight loads in flight per trip
__global__ __launch_bounds__(256) void onetok_dv_kernel(const float *__restrict__ dyb, const float *__restrict__ w_out, long ldwo, int qd,
                                                        int inner, float *__restrict__ dV) {
  __shared__ float part[4][64];
  const int bi = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const float *db = dyb + (long)bi * qd;
  float a0 = 0.0f, a1 = 0.0f;
  if (i < inner) {
    int q = w;
    for (; q + 28 < qd; q += 32) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = w_out[(long)(q + 4 * u) * ldwo + i];
#pragma unroll
      for (int u = 0; u < 8; u += 2) { a0 = fmaf(db[q + 4 * u], wv[u], a0); a1 = fmaf(db[q + 4 * (u + 1)], wv[u + 1], a1); }
    }
    for (; q < qd; q += 4) a0 = fmaf(db[q], w_out[(long)q * ldwo + i], a0);
  }
  part[w][lane] = a0 + a1;
  __syncthreads();
  if (w == 0 && i < inner) dV[(long)bi * inner + i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

__global__ __launch_bounds__(256) void onetok_bwd_weights_kernel(const float *__restrict__ dyb, const float *__restrict__ V,
                                                                 const float *__restrict__ dV, const float *__restrict__ ctx, int ld_ctx,
                                                                 const float *__restrict__ w_v, const float *gamma, const float *beta, int b,
                                                                 int qd, int inner, int D, float *dw_out, long ldwo, float *db_out, float *dw_v,
                                                                 float *__restrict__ partial, int nchunks, int cblocks) {
  if ((int)blockIdx.x < qd) {                         // ---- role A: one row q of dW_out, and db_out[q]
    const int q = blockIdx.x;
    float dq[ONETOK_MAX_B];
    float bsum = 0.0f;
#pragma unroll
    for (int k = 0; k < ONETOK_MAX_B; ++k) { dq[k] = k < b ? dyb[(long)k * qd + q] : 0.0f; bsum += dq[k]; }
    if (dw_out)
      for (int i = threadIdx.x; i < inner; i += blockDim.x) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < ONETOK_MAX_B; ++k)
          if (k < b) acc = fmaf(dq[k], V[(long)k * inner + i], acc);
        dw_out[(long)q * ldwo + i] += acc;
      }
    if (db_out && threadIdx.x == 0) db_out[q] += bsum;
    return;
  }
  // ---- role B: the value half of to_kv and the context LayerNorm's affine (kv_weight_grads_kernel with G and cs formed here)
  __shared__ float pg[4][64], pb[4][64];
  const int r = (int)blockIdx.x - qd, bx = r % cblocks, by = r / cblocks;
  const int c = bx * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
  const int chunk = (inner + nchunks - 1) / nchunks;
  const int n0 = by * chunk, n1 = min(inner, n0 + chunk);
  float sg = 0.0f, sb = 0.0f;
  if (c < D) {
    float zc[ONETOK_MAX_B];
#pragma unroll
    for (int k = 0; k < ONETOK_MAX_B; ++k) zc[k] = k < b ? ctx[(long)k * ld_ctx + c] : 0.0f;
    const float gm = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
    for (int n = n0 + wv; n < n1; n += 4) {
      float g = 0.0f, csn = 0.0f;
#pragma unroll
      for (int k = 0; k < ONETOK_MAX_B; ++k)
        if (k < b) { const float dv = dV[(long)k * inner + n]; g = fmaf(dv, zc[k], g); csn += dv; }
      const float wnc = w_v[(long)n * D + c];
      if (dw_v) dw_v[(long)n * D + c] += g * gm + csn * bt;
      sg += wnc * g;
      sb += wnc * csn;
    }
  }
  pg[wv][threadIdx.x & 63] = sg;
  pb[wv][threadIdx.x & 63] = sb;
  __syncthreads();
  if (wv == 0 && c < D && partial) {      // row-chunk partials of (dgamma, dbeta), summed in fixed order by kv_affine_reduce_kernel
    partial[((long)by * 2 + 0) * D + c] = pg[0][threadIdx.x] + pg[1][threadIdx.x] + pg[2][threadIdx.x] + pg[3][threadIdx.x];
    partial[((long)by * 2 + 1) * D + c] = pb[0][threadIdx.x] + pb[1][threadIdx.x] + pb[2][threadIdx.x] + pb[3][threadIdx.x];
  }
}

bool onetoken_bwd_fused_ok(int b, int qd) {
  static const bool off = getenv("HN_NO_ONETOK_FUSED") != nullptr;      // route switch (A/B): the nine-launch sequence
  return !off && b >= 1 && b <= ONETOK_MAX_B && qd >= 1 && qd <= 4096;
}

// dyb (b, qd) and dV (b, inner): scratch; `partial`: >= KVG_CHUNKS * 2 * D floats.  Gradient pointers may be NULL.
int launch_onetoken_bwd(const float *dy, const float *x_out, const float *x_in, int b, int L, int qd, const float *w_out, long ldwo,
                        int inner, const float *V, const float *ctx, int ld_ctx, int D, const float *w_v, const float *gamma,
                        const float *beta, float *dyb, float *dV, float *dw_out, float *db_out, float *dw_v, float *dgamma, float *dbeta,
                        float *partial, hipStream_t s) {
  HN_REQUIRE(dy && x_out && w_out && V && ctx && w_v && dyb && dV && partial, HN_E_NULL, "onetoken_bwd: NULL pointer");
  HN_REQUIRE(onetoken_bwd_fused_ok(b, qd), HN_E_UNSUPPORTED, "onetoken_bwd: b=%d qd=%d", b, qd);
  hipLaunchKernelGGL(onetok_dyb_kernel, dim3(ceil_div(qd, 64), b), dim3(256), 0, s, dy, x_out, x_in, L, qd, dyb);
  HN_LAUNCH_CHECK("onetok_dyb");
  hipLaunchKernelGGL(onetok_dv_kernel, dim3(ceil_div(inner, 64), b), dim3(256), 0, s, dyb, w_out, ldwo, qd, inner, dV);
  HN_LAUNCH_CHECK("onetok_dv");
  const bool affine = gamma != nullptr && (dgamma != nullptr || dbeta != nullptr);
  const int cblocks = ceil_div(D, 64);
  hipLaunchKernelGGL(onetok_bwd_weights_kernel, dim3(qd + cblocks * KVG_CHUNKS), dim3(256), 0, s, dyb, V, dV, ctx, ld_ctx, w_v, gamma, beta, b, qd,
                     inner, D, dw_out, ldwo, db_out, dw_v, affine ? partial : nullptr, KVG_CHUNKS, cblocks);
  HN_LAUNCH_CHECK("onetok_bwd_weights");
  if (affine) {
    hipLaunchKernelGGL(kv_affine_reduce_kernel, dim3(ceil_div(D, 256)), dim3(256), 0, s, partial, KVG_CHUNKS, D, dgamma, dbeta);
    HN_LAUNCH_CHECK("kv_affine_reduce");
  }
  return HN_OK;
}

// out[seg, c] = sum of `seg` consecutive rows of X (nseg segments): per-sample sums over the latent rows
__global__ __launch_bounds__(256) void segsum_kernel(const float *__restrict__ X, int seg, int cols, float *__restrict__ out) {
  // 64 columns per workgroup, the 4 waves split the segment's rows (4 independent chains each), fixed-order LDS reduce
  __shared__ float part[4][64];
  const int sidx = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  if (c < cols) {
    const float *base = X + (long)sidx * seg * cols + c;
    int r = w;
    for (; r + 12 < seg; r += 16) {
      s0 += base[(long)r * cols]; s1 += base[(long)(r + 4) * cols];
      s2 += base[(long)(r + 8) * cols]; s3 += base[(long)(r + 12) * cols];
    }
    for (; r < seg; r += 4) s0 += base[(long)r * cols];
  }
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && c < cols) out[(long)sidx * cols + c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

int launch_segsum(const float *X, int seg, int cols, int nseg, float *out, hipStream_t s) {
  hipLaunchKernelGGL(segsum_kernel, dim3(ceil_div(cols, 64), nseg), dim3(256), 0, s, X, seg, cols, out);
  HN_LAUNCH_CHECK("segsum");
  return HN_OK;
}

// y = LayerNorm(x) * gamma + beta materialised (the backward needs the normalised operand of dW = dY^T x_hat)
// dv: statistics over the first dv columns (staged models: the pad columns of x, gamma, beta are zero and stay zero in y)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float eps, long rows, int d,
                                                     float *__restrict__ y, int dv) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float *xr = x + r * d;
  float s = 0.0f;
  for (int c = lane; c < d; c += 64) s += xr[c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / (float)dv;
  float q = 0.0f;
  for (int c = lane; c < dv; c += 64) { const float t = xr[c] - mu; q += t * t; }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rs = 1.0f / sqrtf(q / (float)dv + eps);
  for (int c = lane; c < d; c += 64) y[r * d + c] = c < dv ? (xr[c] - mu) * rs * gamma[c] + beta[c] : 0.0f;
}

int launch_ln_fwd(const float *x, const float *gamma, const float *beta, long rows, int d, float *y, hipStream_t s, int dv) {
  if (dv <= 0 || dv > d) dv = d;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)ceil_div_ll(rows, 4)), dim3(256), 0, s, x, gamma, beta, 1e-5f, rows, d, y, dv);
  HN_LAUNCH_CHECK("ln_fwd");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// elementwise pieces
// ------------------------------------------------------------------------------------------------
// dpre = dy * LeakyReLU'(pre); sign(pre) == sign(y) with y = x_out - x_in (or x_out itself without residual)
__global__ __launch_bounds__(256) void leaky_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x_out,
                                                        const float *x_in, float *__restrict__ dpre, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float y = x_in ? x_out[i] - x_in[i] : x_out[i];
    dpre[i] = dy[i] * (y > 0.0f ? 1.0f : 0.01f);
  }
}

int launch_leaky_bwd(const float *dy, const float *x_out, const float *x_in, float *dpre, long n, hipStream_t s) {
  long blocks = ceil_div_ll(n, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, x_out, x_in, dpre, n);
  HN_LAUNCH_CHECK("leaky_bwd");
  return HN_OK;
}

// u = [a | g] (rows, 2*hid) pre-activations; dh (rows, hid).  Writes h = a * act(g) (rows, hid) when requested and
// du = [dh * act(g) | dh * a * act'(g)] in place of u.
__global__ __launch_bounds__(256) void glu_bwd_kernel(float *__restrict__ u, const float *__restrict__ dh, float *h_out,
                                                      long rows, int hid, int gelu) {
  const long n = rows * hid;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / hid;
    const int c = (int)(i - r * hid);
    float *ua = u + r * 2 * hid + c, *ug = ua + hid;
    const float a = *ua, gt = *ug;
    float act, dact;
    if (gelu) {
      const float cdf = 0.5f * (1.0f + erff(gt * 0.70710678118654752440f));
      act = gt * cdf;
      dact = cdf + gt * 0.3989422804014327f * expf(-0.5f * gt * gt);
    } else {
      const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
      act = scale * (gt > 0.0f ? gt : alpha * expm1f(gt));
      dact = scale * (gt > 0.0f ? 1.0f : alpha * expf(gt));
    }
    if (h_out) h_out[i] = a * act;
    if (dh) {
      const float d = dh[i];
      *ua = d * act;
      *ug = d * a * dact;
    }
  }
}

int launch_glu_bwd(float *u, const float *dh, float *h_out, long rows, int hid, int gelu, hipStream_t s) {
  long blocks = ceil_div_ll(rows * hid, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(glu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, u, dh, h_out, rows, hid, gelu);
  HN_LAUNCH_CHECK("glu_bwd");
  return HN_OK;
}

// dst (+)= src elementwise
__global__ __launch_bounds__(256) void axpy_kernel(const float *__restrict__ src, float *__restrict__ dst, long n, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = accumulate ? dst[i] + src[i] : src[i];
}

int launch_add_into(const float *src, float *dst, long n, int accumulate, hipStream_t s) {
  long blocks = ceil_div_ll(n, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n, accumulate);
  HN_LAUNCH_CHECK("add_into");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// Head backward (to_logits :181-185): logits = LN(mean_n x) W^T + bias.  One workgroup per sample writes
// dx[b, l, :] = dpooled[b, :] / L and its own row of parameter-gradient partials
// [dW (out*d) | dgamma (d) | dbeta (d) | dbias (out)]; colsum over the samples accumulates them (deterministic).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_bwd_kernel(const float *__restrict__ x, int L, int d, const float *__restrict__ nw,
                                                       const float *__restrict__ nb, const float *__restrict__ w, int out_dims,
                                                       const float *__restrict__ dlogits, float *__restrict__ dx,
                                                       float *__restrict__ partial, int dv) {
  // dv: LayerNorm over the first dv columns (staged models; the pad columns carry no gradient)
  extern __shared__ float sm[];
  float *pooled = sm, *xn = sm + d, *dyn = sm + 2 * d, *red = sm + 3 * d, *part = sm + 3 * d + 8;   // part[4][d]
  const int tid = threadIdx.x, bi = blockIdx.x;
  const float *xb = x + (long)bi * L * d;
  float *prow = partial + (long)bi * ((long)out_dims * d + 2 * d + out_dims);
  {
    const int wv = tid >> 6, ln = tid & 63;
    for (int c = ln; c < d; c += 64) {
      float s = 0.0f;
      for (int r = wv; r < L; r += 32) {                       // eight rows in flight per trip (one dependent load per row: 32 us at L = 128)
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = r + 4 * u < L ? xb[(long)(r + 4 * u) * d + c] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v8[u];
      }
      part[wv * d + c] = s;
    }
  }
  __syncthreads();
  for (int c = tid; c < d; c += blockDim.x) pooled[c] = (part[c] + part[d + c] + part[2 * d + c] + part[3 * d + c]) / (float)L;
  __syncthreads();
  float s = 0.0f;
  for (int c = tid; c < d; c += blockDim.x) s += pooled[c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)dv;
  __syncthreads();
  float q = 0.0f;
  for (int c = tid; c < dv; c += blockDim.x) { const float t = pooled[c] - mean; q += t * t; }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if ((tid & 63) == 0) red[tid >> 6] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)dv + 1e-5f);
  __syncthreads();
  float m1 = 0.0f, m2 = 0.0f;
  for (int c = tid; c < d; c += blockDim.x) {
    const float n = c < dv ? (pooled[c] - mean) * rstd : 0.0f;
    const float yh = n * nw[c] + nb[c];
    float dyh = 0.0f;
    for (int o = 0; o < out_dims; ++o) {
      const float dl = dlogits[(long)bi * out_dims + o];
      dyh += dl * w[(long)o * d + c];
      prow[(long)o * d + c] = dl * yh;
    }
    prow[(long)out_dims * d + c] = dyh * n;
    prow[(long)out_dims * d + d + c] = dyh;
    const float dyg = dyh * nw[c];
    xn[c] = n;
    dyn[c] = dyg;
    m1 += dyg;
    m2 += dyg * n;
  }
  if (tid < out_dims) prow[(long)out_dims * d + 2 * d + tid] = dlogits[(long)bi * out_dims + tid];
  for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o); m2 += __shfl_xor(m2, o); }
  if ((tid & 63) == 0) { red[tid >> 6] = m1; red[4 + (tid >> 6)] = m2; }
  __syncthreads();
  const float mm1 = (red[0] + red[1] + red[2] + red[3]) / (float)dv, mm2 = (red[4] + red[5] + red[6] + red[7]) / (float)dv;
  for (int c = tid; c < d; c += blockDim.x) pooled[c] = c < dv ? rstd * (dyn[c] - mm1 - xn[c] * mm2) / (float)L : 0.0f;   // d pooled / L
  __syncthreads();
  float *dxb = dx + (long)bi * L * d;
  for (long i = tid; i < (long)L * d; i += blockDim.x) dxb[i] = pooled[i % d];
}

size_t head_bwd_scratch_floats(int b, int d, int out_dims) { return (size_t)b * ((size_t)out_dims * d + 2 * d + out_dims); }

struct HeadColsum { const float *X; long pitch; int rows; int begin[5]; float *dst[4]; int *zero; int nzero; };
// out[seg][c - begin[seg]] += sum_r X[r, c]: the summation order of colsum_kernel with one row chunk (four strided partial sums, added
// in wave order), so the results are the bits the four separate launches produced
__global__ __launch_bounds__(256) void head_colsum_kernel(HeadColsum h) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float s = 0.0f;
  if (c < h.begin[4])
    for (long r = w; r < h.rows; r += 4) s += h.X[r * h.pitch + c];
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < h.begin[4]) {
    const float v = 1.0f * (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
    const int seg = c >= h.begin[3] ? 3 : c >= h.begin[2] ? 2 : c >= h.begin[1] ? 1 : 0;
    float *dst = h.dst[seg];
    if (dst) dst[c - h.begin[seg]] += v;
  }
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < h.nzero; i += blockDim.x) h.zero[i] = 0;
}

int launch_head_bwd(const float *x, int b, int L, int d, const float *nw, const float *nb, const float *w, int out_dims,
                    const float *dlogits, float *dx, float *dnw, float *dnb, float *dw, float *dbias, float *scratch,
                    hipStream_t s, int dv, int *zero, int nzero) {
  HN_REQUIRE(x && nw && nb && w && dlogits && dx && scratch, HN_E_NULL, "head_bwd: NULL pointer");
  HN_REQUIRE(dv >= 0 && dv <= d, HN_E_SHAPE, "head_bwd: valid=%d of d=%d", dv, d);
  if (dv == 0) dv = d;
  HN_REQUIRE(out_dims <= 256, HN_E_UNSUPPORTED, "head_bwd: out_dims=%d (<= 256)", out_dims);
  const size_t lds = (size_t)(7 * d + 8) * sizeof(float);
  HN_REQUIRE(lds <= 64 * 1024, HN_E_UNSUPPORTED, "head_bwd: l_d=%d too large", d);
  hipLaunchKernelGGL(head_bwd_kernel, dim3(b), dim3(256), lds, s, x, L, d, nw, nb, w, out_dims, dlogits, dx, scratch, dv);
  HN_LAUNCH_CHECK("head_bwd");
  // the four parameter gradients are column sums over the b rows of `scratch`, in adjacent column ranges: ONE launch (four colsum
  // launches at the ~4.6 us floor until round 6), which also clears the cluster flags of the backward chains (`zero`)
  HeadColsum hc;
  const int widths[4] = {out_dims * d, d, d, out_dims};
  float *dsts[4] = {dw, dnw, dnb, dbias};
  int off = 0;
  for (int i = 0; i < 4; ++i) { hc.begin[i] = off; hc.dst[i] = dsts[i]; off += widths[i]; }
  hc.begin[4] = off;
  hc.X = scratch; hc.pitch = (long)out_dims * d + 2 * d + out_dims; hc.rows = b; hc.zero = zero; hc.nzero = zero ? nzero : 0;
  hipLaunchKernelGGL(head_colsum_kernel, dim3(ceil_div(off, 64)), dim3(256), 0, s, hc);
  HN_LAUNCH_CHECK("head_colsum");
  return HN_OK;
}

}  // namespace hn
