// bf16-MFMA form of the patch-bag K/V projection (core_precision = "bf16", inference forward only).
//
//   C (M, N) = alpha * (A * gamma + beta) W^T      A (M, K) fp32 = the normalised context rows of an explicit binding (b * N tokens,
//                                                  K = D = 773 at cfg4 / cfg5), W (N = 2 * inner, K) fp32 = to_kv.weight
//
// is 52 GF per block at cfg4 and 42 % of the cfg5 forward on fp32 MFMA (gemm_big_kernel at 0.7 of the 157 TF/s peak).  With the
// operands rounded to bf16 ONCE and fp32 accumulation (the same contract as the bf16 image / volume core, SURVEY.md 8d, tolerance
// 2e-2 against the fp32 oracle) the same product runs on v_mfma_f32_16x16x32_bf16:
//
//   * the affine prologue leaves the loop:  (A gamma + beta) W^T = A (W gamma)^T + W beta.  gemm_bf16_stage_kernel writes
//     Wb = bf16(W * gamma) with rows zero-padded to a multiple of 64 columns (16-byte aligned whatever ldw is -- 773 floats at
//     cfg4) and the fp32 row vector cb = W beta (exact: it never passes through bf16); one launch of N workgroups per call.
//   * the context rows are rounded ONCE per forward (rows_to_bf16_kernel: Ab, same zero-padded 64-column pitch -- rows start on
//     128-byte lines) and serve every layer's projection: half the bytes per k-tile, no conversion in the loop.
//   * gemm_bf16_kernel: 128 x 128 output tile, 64-wide k-tiles, both operands as 16-byte pieces (8 lanes = one line of a row)
//     through registers into a two-stage LDS image [stage][A k-half 0, 1, W k-half 0, 1][row][32 bf16]: a 16 x 32 sub-tile is one
//     KB in the order the MFMA operand fragment reads it (lane (g, j) = row j, k = 8 g .. 8 g + 7), with the 16-byte chunks of a
//     row XOR-swizzled and the k-halves 64 bytes off a 128-byte multiple so that neither ds_read_b128 nor ds_write_b128 conflicts
//     (lane groups of MI355X_MICROARCH.md "LDS"; SQ_LDS_BANK_CONFLICT was a third of the LDS cycles without the two).
//   * one barrier per k-tile: tile kt is contracted from stage kt & 1 while tile kt + 1 moves registers -> other stage and tile
//     kt + 2 is requested from memory, each of those instructions issued in the shadow of one MFMA of the first k-half.
//   * W sub-tiles are the MFMA's A operand and context sub-tiles its B operand: the accumulator quad of a lane is then four
//     CONSECUTIVE output columns of one row -> float4 stores.
//   * work-item order as gemm_big_kernel: the column tiles that share a 128-row block of A run back to back on one XCD.
//   * gemm_bf16_kernel<true> (heads of 64): the output goes straight into the bf16 images of
//     the explicit bf16 core (attention_bf16.hip, EXPL) -- K token-major, V fragment-major per (sample, head); the column tiles of
//     the V half swap the MFMA operands so that a lane holds four consecutive tokens of a column = 8 contiguous bytes of a V tile.
#include "common.h"
#include <type_traits>
#pragma clang diagnostic ignored "-Winline-asm"      // (M0 on the clobber list of bf_glds16)

namespace hn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int HM = 128, HNT = 128, HK = 64;
// bytes between the two k-halves of an operand tile (128 rows x 32 bf16 each) + 64: the 8 lanes ds_write_b128 serves per LDS cycle hold
// the 8 pieces of ONE row -- 4 per k-half -- and the pad puts the second half's chunks on the other 16 banks of the 128-byte window
// (without it: two-way conflicts on every store, a third of the LDS cycles of the loop by SQ_LDS_BANK_CONFLICT)
constexpr int PLANE = 128 * 64 + 64;

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// Wb[n][k] = bf16(W[n][k] * gamma[k]) for k < K, 0 up to Kp; cb[n] = sum_k W[n][k] * beta[k] (gamma / beta NULL: 1 / 0)
__global__ __launch_bounds__(256) void gemm_bf16_stage_kernel(const float *__restrict__ W, long ldw, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, int K, int Kp, unsigned *__restrict__ Wb,
                                                              float *__restrict__ cb) {
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float *w = W + (long)n * ldw;
  unsigned *out = Wb + (long)n * (Kp / 2);
  float part = 0.0f;
  for (int k = 2 * tid; k < Kp; k += 512) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ke = k + e;
      float x = 0.0f;
      if (ke < K) {
        x = w[ke];
        if (beta) part = fmaf(x, beta[ke], part);
        if (gamma) x *= gamma[ke];
      }
      v[e] = x;
    }
    out[k / 2] = pk_bf16(v[0], v[1]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) cb[n] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Ab[m][k] = bf16(A[m][k]) for k < K, 0 up to Kp: the normalised context of a patch-bag modality, once per forward (every layer's
// K/V projection reads it).  One 16-byte piece (8 columns) per thread.
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float *__restrict__ A, long lda, long M, int K, int Kp, u32x4 *__restrict__ out) {
  const int ppr = Kp / 8;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= M * ppr) return;
  const long m = id / ppr;
  const int k = (int)(id - m * ppr) * 8;
  const float *a = A + m * lda + k;
  float v[8];
  if (k + 7 < K) {
    const f32x4 lo = *(const f32x4 *)a, hi = *(const f32x4 *)(a + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = k + e < K ? a[e] : 0.0f;      // nothing is read past column K - 1 (pad columns may hold anything)
  }
  u32x4 o;
  o.x = pk_bf16(v[0], v[1]); o.y = pk_bf16(v[2], v[3]); o.z = pk_bf16(v[4], v[5]); o.w = pk_bf16(v[6], v[7]);
  out[id] = o;
}

struct Bf16GemmArgs {
  const uint16_t *Ab, *Wb;      // (M, Kp) and (N, Kp) bf16 images, Kp % 64 == 0, zero beyond K
  const float *cb;              // (N) fp32: W beta
  float *C; long ldc;
  float alpha;
  int M, Kp, ntm, ntn;
  // IMG: instead of fp32 rows, the bf16 images attn_core_bf16_kernel<.., EXPL> reads (N = 2 * inner, dim_head 64):
  // K16 (M, inner) token-major; V16 per (sample, head) fragment-major (np / 32, 4, 4, 16, 8)
  uint16_t *K16, *V16;
  int inner, heads, tokens, np;      // tokens per sample and that rounded up to 32 (V tiles hold 32 tokens; the pad slots are zeroed)
  int abl;                           // bench builds only (HN_GEMM_BF16_BENCH): 1 = no stores, 2 = no loads, 4 = no MFMAs
};

// Epilogue shared by both kernels: acc[column sub-tile i][row sub-tile t] of the wave's 64 x 64 block (see gemm_bf16_kernel)
template <bool IMG>
__device__ __forceinline__ void bf16_epilogue(const Bf16GemmArgs &g, f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int fj, int fg,
                                              bool v_half) {
  if (v_half) {
    // D[m_local = 4 fg + r][n_local = fj]: four consecutive tokens of column fj -> 8 bytes of one fragment-major V tile
    // (tile (sample, head, token / 32, col / 16) = [token % 32 / 8][col % 16][token % 8]); token counts that are not multiples of 4
    // let a row quad straddle two samples: element-wise stores there
    const int blocks_per = g.np >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + 16 * i + fj;
      const float cbv = g.cb[n];
      const int c_all = n - g.inner, head = c_all >> 6, c = c_all & 63;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = m0 + wm * 64 + 16 * t + 4 * fg;
        if ((g.tokens & 3) == 0) {      // (then M % 4 == 0 too: the four rows are in or out together)
          if (m < g.M) {
            const int bi = m / g.tokens, tok = m - bi * g.tokens;
            const long tile_idx = (((long)bi * g.heads + head) * blocks_per + (tok >> 5)) * 4 + (c >> 4);
            u32x2 v;
            v.x = pk_bf16((acc[i][t][0] + cbv) * g.alpha, (acc[i][t][1] + cbv) * g.alpha);
            v.y = pk_bf16((acc[i][t][2] + cbv) * g.alpha, (acc[i][t][3] + cbv) * g.alpha);
            *(u32x2 *)(g.V16 + tile_idx * 512 + ((((tok & 31) >> 3) * 16 + (c & 15)) << 3) + (tok & 7)) = v;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int mr = m + r;
            if (mr < g.M) {
              const int bi = mr / g.tokens, tok = mr - bi * g.tokens;
              const long tile_idx = (((long)bi * g.heads + head) * blocks_per + (tok >> 5)) * 4 + (c >> 4);
              g.V16[tile_idx * 512 + ((((tok & 31) >> 3) * 16 + (c & 15)) << 3) + (tok & 7)] = (uint16_t)pk_bf16((acc[i][t][r] + cbv) * g.alpha, 0.0f);
            }
          }
        }
      }
    }
    return;
  }
  // D[n_local = 4 fg + r][m_local = fj]: four consecutive columns of row fj
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + 16 * i + 4 * fg;
    const f32x4 c4 = *(const f32x4 *)(g.cb + n);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int m = m0 + wm * 64 + 16 * t + fj;
      if (m < g.M) {
        const f32x4 v = (acc[i][t] + c4) * g.alpha;
        if (IMG) {
          u32x2 o;
          o.x = pk_bf16(v[0], v[1]);
          o.y = pk_bf16(v[2], v[3]);
          *(u32x2 *)(g.K16 + (long)m * g.inner + n) = o;
        } else {
          *(f32x4 *)(g.C + (long)m * g.ldc + n) = v;
        }
      }
    }
  }
}

template <bool IMG>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(Bf16GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];      // two stages of {A k-half 0, 1, W k-half 0, 1}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int id = blockIdx.x, xcd = id & 7, seq = id >> 3;
  const int n_tile = seq % g.ntn, m_tile = (seq / g.ntn) * 8 + xcd;
  if (m_tile >= g.ntm) return;
  const int m0 = m_tile * HM, n0 = n_tile * HNT;
  const int Kp = g.Kp;

  // loader (both operands): 16-byte piece lp of the 8 of a 64-wide row piece, rows lr + 32 j: 8 lanes = one 128-byte line
  const int lp = tid & 7, lr = tid >> 3;
  const uint16_t *arow[4], *wrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    arow[j] = g.Ab + (long)min(m0 + lr + 32 * j, g.M - 1) * Kp + 8 * lp;
    wrow[j] = g.Wb + (long)(n0 + lr + 32 * j) * Kp + 8 * lp;
  }
  u32x4 ra[4], rb[4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ra[j] = *(const u32x4 *)(arow[j] + k0);
      rb[j] = *(const u32x4 *)(wrow[j] + k0);
    }
  };
  // chunk swizzle: the 16-byte chunk c of row r lives at chunk c ^ s(r / 4 % 4), s = (0, 3, 2, 1): the 16 lanes ds_read_b128 serves per
  // LDS cycle ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... of the wave: rows j and j + 12 of one k-chunk with rows j + 4 .. j + 11 of
  // the next) then hit 16 different chunk columns (unswizzled: two-way conflicts, measured 5 % of the loop)
  const int dst = (lp >> 2) * PLANE + lr * 64 + (((lp & 3) ^ ((0 - (lr >> 2)) & 3)) * 16);
  auto store_tile = [&](unsigned char *stage) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(u32x4 *)(stage + dst + j * 32 * 64) = ra[j];
      *(u32x4 *)(stage + 2 * PLANE + dst + j * 32 * 64) = rb[j];
    }
  };

  const int wm = wave >> 1, wn = wave & 1;      // a wave owns 64 rows x 64 columns = 4 x 4 MFMA tiles
  const int fj = lane & 15, fg = lane >> 4;
  const int fsw = (fg ^ ((0 - (fj >> 2)) & 3)) * 16;
  const int a_off = (wm * 64 + fj) * 64 + fsw, w_off = 2 * PLANE + (wn * 64 + fj) * 64 + fsw;
  f32x4 acc[4][4];      // [column sub-tile i][row sub-tile t]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

  // Software pipeline, ONE barrier per k-tile: while tile kt is contracted from stage kt & 1, tile kt + 1 moves from registers into
  // the other stage and tile kt + 2 is requested from memory (a whole contraction ahead of its first use).  With two barriers and
  // one stage the phases of all resident workgroups ran in step and their costs added up (measured: matrix 33 + loads 22 + LDS 26 us
  // = the 80 us of the loop).
  const int nk = Kp / HK;
  load_tile(0);
  store_tile(lds_raw);
  if (nk > 1) load_tile(HK);
  __syncthreads();
  auto tile = [&](int kt, auto do_store, auto do_load, auto flip) {
    constexpr bool FLIP = decltype(flip)::value;      // context sub-tile as the MFMA's A operand: a lane's accumulator quad = four consecutive ROWS
    const unsigned char *cur = lds_raw + (kt & 1) * 4 * PLANE;
    unsigned char *nxt = lds_raw + ((kt + 1) & 1) * 4 * PLANE;
    const int k2 = (kt + 2) * HK;
    bf16x8 fa[2][4], fw[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fa[0][t] = *(const bf16x8 *)(cur + a_off + t * 16 * 64);
      fw[0][t] = *(const bf16x8 *)(cur + w_off + t * 16 * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    // first k-half: in the shadow of each MFMA one fragment read of the second k-half, then one LDS store (tile kt + 1) or one global
    // load (tile kt + 2)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc[q >> 2][q & 3] = FLIP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][q & 3], fw[0][q >> 2], acc[q >> 2][q & 3], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][q >> 2], fa[0][q & 3], acc[q >> 2][q & 3], 0, 0, 0);
      if (q < 4) fw[1][q] = *(const bf16x8 *)(cur + PLANE + w_off + q * 16 * 64);
      else if (q < 8) fa[1][q - 4] = *(const bf16x8 *)(cur + PLANE + a_off + (q - 4) * 16 * 64);
      if (q < 8) {
        if constexpr (decltype(do_store)::value) {
          if (q < 4) *(u32x4 *)(nxt + dst + q * 32 * 64) = ra[q & 3];
          else *(u32x4 *)(nxt + 2 * PLANE + dst + (q - 4) * 32 * 64) = rb[q & 3];
        }
      } else {
        if constexpr (decltype(do_load)::value) {
          if (q < 12) ra[q & 3] = *(const u32x4 *)(arow[q & 3] + k2);
          else rb[q & 3] = *(const u32x4 *)(wrow[q & 3] + k2);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
      acc[q >> 2][q & 3] = FLIP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][q & 3], fw[1][q >> 2], acc[q >> 2][q & 3], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][q >> 2], fa[1][q & 3], acc[q >> 2][q & 3], 0, 0, 0);
    __syncthreads();
  };
  const std::true_type yes;
  const std::false_type no;
  const bool v_half = IMG && n0 >= g.inner;      // (workgroup-uniform: a 128-column tile lies in the K or in the V half)
  if (v_half) {
    for (int kt = 0; kt + 2 < nk; ++kt) tile(kt, yes, yes, yes);
    if (nk >= 2) tile(nk - 2, yes, no, yes);
    tile(nk - 1, no, no, yes);
  } else {
    for (int kt = 0; kt + 2 < nk; ++kt) tile(kt, yes, yes, no);
    if (nk >= 2) tile(nk - 2, yes, no, no);
    tile(nk - 1, no, no, no);
  }

  bf16_epilogue<IMG>(g, acc, m0, n0, wm, wn, fj, fg, v_half);
}

// ------------------------------------------------------------------------------------------------
// The same product on the skeleton of gemm_nt.hip (round 4): both operands land in LDS by LDS-DMA (buffer_load_dwordx4 ... lds, issued
// from inline asm, every wait counted by hand), two ring slots of 32 KB, ONE barrier in the MIDDLE of a 64-wide k-tile, two 4-wave
// workgroups per CU.  A bf16 row piece of 64 columns is 128 bytes -- the byte geometry of the fp32 kernel's 32-float rows -- so an
// instruction lands 8 rows x 8 16-byte chunks (whole lines) as [row][128 B]; chunk c of row r sits at position c ^ ((r >> 1) & 7),
// applied on the SOURCE address of the DMA (the destination is linear) and again by the fragment reads.  With that swizzle the 16
// lanes ds_read_b128 serves per LDS cycle (rows {0-3, 12-15} of one chunk with rows {4-11} of the next) touch 16 different 16-byte
// slots of the 256-byte bank window.  No staging registers, no ds_write, nothing of the loader on the VALU.
// ------------------------------------------------------------------------------------------------
// Image epilogue of the LDS-DMA kernel through LDS (whole samples per tile: tokens % 128 == 0, M % 128 == 0): the direct form stores
// 8 bytes per lane -- 32-byte pieces of K rows, single 8-byte pieces of V tiles -- the pattern that held the token encoder at half its
// rate.  Here the workgroup assembles its 128 x 128 bf16 block in the (free) ring in its FINAL layout -- K: token-major
// rows of 256 bytes, 16-byte chunks XOR-swizzled by the row; V: the 32 fragment-major 1 KB tiles [token % 32 / 8][col % 16][token % 8] --
// and copies it out in 16-byte pieces: whole 256-byte rows / whole 1 KB tiles of consecutive memory per 16 / 64 lanes.
__device__ __forceinline__ void bf16_epilogue_img_lds(const Bf16GemmArgs &g, f32x4 (&acc)[4][4], unsigned char *lds, int m0, int n0, int wm,
                                                      int wn, int fj, int fg, bool v_half, int tid) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                     // every wave is done with the ring
  if (!v_half) {
    // D[n_local = 4 fg + r][m_local = fj]: four consecutive columns of row fj
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = wn * 64 + 16 * i + 4 * fg;
      const f32x4 c4 = *(const f32x4 *)(g.cb + n0 + nl);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ml = wm * 64 + 16 * t + fj;
        const f32x4 v = (acc[i][t] + c4) * g.alpha;
        u32x2 o;
        o.x = pk_bf16(v[0], v[1]);
        o.y = pk_bf16(v[2], v[3]);
        *(u32x2 *)(lds + ml * 256 + ((((nl >> 3) ^ (ml & 15)) << 4) | ((nl & 4) << 1))) = o;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = tid + 256 * k, row = idx >> 4, pos = idx & 15;
      const u32x4 v = *(const u32x4 *)(lds + row * 256 + (pos << 4));
      *(u32x4 *)(g.K16 + (long)(m0 + row) * g.inner + n0 + ((pos ^ (row & 15)) << 3)) = v;
    }
    return;
  }
  // D[m_local = 4 fg + r][n_local = fj]: four consecutive tokens of column fj
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nl = wn * 64 + 16 * i + fj;
    const float cbv = g.cb[n0 + nl];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ml = wm * 64 + 16 * t + 4 * fg;
      u32x2 v;
      v.x = pk_bf16((acc[i][t][0] + cbv) * g.alpha, (acc[i][t][1] + cbv) * g.alpha);
      v.y = pk_bf16((acc[i][t][2] + cbv) * g.alpha, (acc[i][t][3] + cbv) * g.alpha);
      *(u32x2 *)(lds + (((ml >> 5) * 8 + (nl >> 4)) << 10) + (((ml & 31) >> 3) << 8) + ((nl & 15) << 4) + ((ml & 7) << 1)) = v;
    }
  }
  __syncthreads();
  const int bi = m0 / g.tokens, tok0 = m0 - bi * g.tokens, blocks_per = g.np >> 5;
  const int head0 = (n0 - g.inner) >> 6;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = tid + 256 * k, tile = idx >> 6, piece = idx & 63;      // tile = token block (0..3) * 8 + column block (0..7)
    const int tb = tile >> 3, cb = tile & 7;
    const long tile_idx = (((long)bi * g.heads + head0 + (cb >> 2)) * blocks_per + (tok0 >> 5) + tb) * 4 + (cb & 3);
    *(u32x4 *)(g.V16 + tile_idx * 512 + piece * 8) = *(const u32x4 *)(lds + (tile << 10) + (piece << 4));
  }
}

__device__ __forceinline__ void bf_glds16(const i32x4 &rsrc, unsigned lds_byte, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_byte), "v"(voffset), "s"(rsrc), "s"(soffset)
               : "memory", "m0");
}
template <int N> __device__ __forceinline__ void bf_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool IMG>
__global__ __launch_bounds__(256, 2) void gemm_bf16_glds_kernel(Bf16GemmArgs g) {
  constexpr int S = 2, OPB = 128 * 128, STAGE_B = 2 * OPB, LW = 8;      // ring slot: A [128][128 B] then W [128][128 B]
  __shared__ __attribute__((aligned(16))) unsigned char lds[S * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int id = blockIdx.x, xcd = id & 7, seq = id >> 3;
  const int n_tile = seq % g.ntn, m_tile = (seq / g.ntn) * 8 + xcd;
  if (m_tile >= g.ntm) return;
  const int m0 = m_tile * HM, n0 = n_tile * HNT;
  const int Kp = g.Kp, rowb = Kp * 2;                                   // bytes per operand row
  const int rows_a = min(HM, g.M - m0);
  const unsigned a_clamp = (unsigned)rows_a * (unsigned)rowb, w_clamp = (unsigned)HNT * (unsigned)rowb;
  const i32x4 rsA = make_rsrc(g.Ab + (long)m0 * Kp, a_clamp);          // rows past M read as zero
  const i32x4 rsW = make_rsrc(g.Wb + (long)n0 * Kp, w_clamp);
  // loader: lane -> (row lane >> 3 of the 8-row group, position lane & 7); row group u = wave + 4 q, so (u & 1) == (wave & 1)
  const int voff = (lane >> 3) * rowb + ((((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4))) << 4));
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char *)lds);
  auto issue = [&](int kt) {
#pragma unroll
    for (int q = 0; q < LW; ++q) {
      const int u = wave + 4 * q;                                       // u < 16: A row group u; else W row group u - 16
      const unsigned dst = lds_base + (unsigned)((kt % S) * STAGE_B + u * 1024);
      const unsigned so = (unsigned)(8 * (u & 15) * rowb + kt * 128);
      if (q < 4) bf_glds16(rsA, dst, voff, (int)min(so, a_clamp));
      else bf_glds16(rsW, dst, voff, (int)min(so, w_clamp));
    }
  };
  const int wm = wave >> 1, wn = wave & 1;      // a wave owns 64 rows x 64 columns = 4 x 4 MFMA tiles
  const int fj = lane & 15, fg = lane >> 4;
  const int sw = fj >> 1;                        // ((row >> 1) & 7) of row = 16 t + fj (+ 64 wm)
  const int a_row = (wm * 64 + fj) * 128, w_row = OPB + (wn * 64 + fj) * 128;
  f32x4 acc[4][4];      // [column sub-tile i][row sub-tile t]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  struct Frags { bf16x8 a[4], w[4]; };
  auto read_frags = [&](int kt, int kh, Frags &f) {
    const unsigned char *st = lds + (kt % S) * STAGE_B;
    const int pos = ((kh * 4 + fg) ^ sw) << 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f.a[t] = *(const bf16x8 *)(st + a_row + t * 16 * 128 + pos);
      f.w[t] = *(const bf16x8 *)(st + w_row + t * 16 * 128 + pos);
    }
  };
  const bool v_half = IMG && n0 >= g.inner;      // (workgroup-uniform: a 128-column tile lies in the K or in the V half)
  auto run = [&](auto flip) {
    constexpr bool FLIP = decltype(flip)::value;      // context sub-tile as the MFMA's A operand: a lane's accumulator quad = four consecutive ROWS
    auto mfma_half = [&](const Frags &f) {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        acc[q >> 2][q & 3] = FLIP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[q & 3], f.w[q >> 2], acc[q >> 2][q & 3], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[q >> 2], f.a[q & 3], acc[q >> 2][q & 3], 0, 0, 0);
    };
    const int nk = Kp / HK;
    issue(0);
    if (nk > 1) issue(1);
    Frags f0, f1;
    if (nk > 1) bf_wait_vmcnt<LW>(); else bf_wait_vmcnt<0>();            // tile 0 landed, tile 1 may be in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(0, 0, f0);
    for (int kt = 0; kt < nk; ++kt) {
      const bool last = kt + 1 == nk;
      read_frags(kt, 1, f1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(g.abl & 4)) mfma_half(f0);
      __builtin_amdgcn_sched_barrier(0);
      if (!last) {
        bf_wait_vmcnt<0>();                                             // two slots: the tile needed next is the only one in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + S < nk && !(g.abl & 2)) issue(kt + S);
        read_frags(kt + 1, 0, f0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(g.abl & 4)) mfma_half(f1);
    }
  };
  if (v_half) run(std::true_type{}); else run(std::false_type{});
  if (g.abl & 1) { if (g.alpha != 12345.0f) return; }
  if (IMG && g.tokens % 128 == 0 && g.M % 128 == 0 && !(g.abl & 8)) {      // (workgroup-uniform)
    bf16_epilogue_img_lds(g, acc, lds, m0, n0, wm, wn, fj, fg, v_half, tid);
    return;
  }
  bf16_epilogue<IMG>(g, acc, m0, n0, wm, wn, fj, fg, v_half);
}

// token counts that are not multiples of 32: the last 32-token block of every (sample, head) V image (4 tiles = 4 KB) is zeroed before
// the projection writes its valid tokens -- the core multiplies the pad slots by probabilities that are exactly 0, which NaNs survive
__global__ __launch_bounds__(256) void v_tail_zero_kernel(u32x4 *__restrict__ V16, int blocks_per) {
  V16[((long)blockIdx.x * blocks_per + (blocks_per - 1)) * 256 + threadIdx.x] = (u32x4){0u, 0u, 0u, 0u};      // 4 tiles x 1 KB = 256 pieces
}

// Qf[(sample, head)][row][64] = bf16(Q[sample * L + row][64 head ..]) for row < L, 0 up to Lp: the query side of the explicit bf16 core
__global__ __launch_bounds__(256) void q_rows_to_bf16_kernel(const float *__restrict__ Q, long ldq, int heads, int L, int Lp, long total, u32x4 *__restrict__ out) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece (8 columns)
  if (id >= total) return;
  const int piece = (int)(id & 7);
  const long r = id >> 3;
  const int row = (int)(r % Lp);
  const long bh = r / Lp;
  const int hi = (int)(bh % heads);
  const long bi = bh / heads;
  u32x4 o = (u32x4){0u, 0u, 0u, 0u};
  if (row < L) {
    const float *q = Q + (bi * L + row) * ldq + hi * 64 + piece * 8;
    const f32x4 lo = *(const f32x4 *)q, hi4 = *(const f32x4 *)(q + 4);
    o.x = pk_bf16(lo.x, lo.y); o.y = pk_bf16(lo.z, lo.w); o.z = pk_bf16(hi4.x, hi4.y); o.w = pk_bf16(hi4.z, hi4.w);
  }
  out[id] = o;
}

}  // namespace

int gemm_bf16_pitch(int K) { return (K + HK - 1) / HK * HK; }

// floats of scratch launch_gemm_bf16 needs for an (N, K) weight: the fp32 row vector W beta + the bf16 image (rows padded to 64 columns)
size_t gemm_bf16_stage_floats(int N, int K) { return (size_t)N * gemm_bf16_pitch(K) / 2 + (size_t)N + 64; }

// shape side of the eligibility (plan time: decides whether a modality keeps a bf16 image of its context rows)
bool gemm_bf16_shape_ok(long M, int N, int K) {
  static const bool off = tuning_env("HN_NO_BF16_PROJ") != nullptr;      // development switch: the fp32 projection under core_precision = bf16
  return !off && M >= 2048 && M < (1L << 31) && K >= 256 && N >= HNT && N % HNT == 0;
}

bool gemm_bf16_eligible(const GemmArgs &g) {
  const bool plain = g.W2 == nullptr && g.batch == 1 && (g.pro == PRO_NONE || g.pro == PRO_AFFINE) && !g.bias && !g.R && g.act == ACT_NONE &&
                     g.glu_offset == 0 && (g.col_group == 0 || g.col_group == g.col_group_pitch);
  return plain && gemm_bf16_shape_ok(g.M, g.N, g.K) && g.ldc % 4 == 0 && ((uintptr_t)g.C & 15) == 0;
}

int launch_rows_to_bf16(const float *A, long lda, long M, int K, uint16_t *out, hipStream_t s) {
  HN_REQUIRE(A && out, HN_E_NULL, "rows_to_bf16: NULL operand");
  HN_REQUIRE(lda % 4 == 0 && (((uintptr_t)A | (uintptr_t)out) & 15) == 0, HN_E_SHAPE, "rows_to_bf16: unaligned operand (lda=%ld)", lda);
  const int Kp = gemm_bf16_pitch(K);
  const long blocks = (M * (Kp / 8) + 255) / 256;
  HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "rows_to_bf16: grid too large");
  hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, A, lda, M, K, Kp, (u32x4 *)out);
  HN_LAUNCH_CHECK("rows_to_bf16");
  return HN_OK;
}

// g describes the fp32 product (A is not read: Ab = its bf16 image from launch_rows_to_bf16, pitch gemm_bf16_pitch(K))
int launch_q_rows_to_bf16(const float *Q, long ldq, int b, int heads, int L, int Lp, uint16_t *Qf, hipStream_t s) {
  HN_REQUIRE(Q && Qf, HN_E_NULL, "q_rows_to_bf16: NULL operand");
  HN_REQUIRE(ldq % 4 == 0 && (((uintptr_t)Q | (uintptr_t)Qf) & 15) == 0, HN_E_SHAPE, "q_rows_to_bf16: unaligned operand (ldq=%ld)", ldq);
  const long total = (long)b * heads * Lp * 8;
  hipLaunchKernelGGL(q_rows_to_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Q, ldq, heads, L, Lp, total, (u32x4 *)Qf);
  HN_LAUNCH_CHECK("q_rows_to_bf16");
  return HN_OK;
}

// K16 / V16 != NULL: write the bf16 K / V images of the explicit bf16 core instead of the fp32 rows g.C (heads of 64, g.N = 2 * 64 * heads,
// `tokens` rows per sample; V16 holds (M / tokens) * heads images of roundup32(tokens) x 64)
int launch_gemm_bf16(const GemmArgs &g, const uint16_t *Ab, float *stage, hipStream_t s, uint16_t *K16, uint16_t *V16, int tokens) {
  HN_REQUIRE(Ab && g.W && (g.C || K16) && stage, HN_E_NULL, "gemm_bf16: NULL operand");
  const bool img = K16 != nullptr;
  HN_REQUIRE(!img || (V16 && tokens > 0 && g.M % tokens == 0 && g.N % 256 == 0 && (((uintptr_t)K16 | (uintptr_t)V16) & 15) == 0),
             HN_E_SHAPE, "gemm_bf16: image output needs whole samples (M=%d, tokens=%d) and N %% 256 == 0 (N=%d)", g.M, tokens, g.N);
  HN_REQUIRE(gemm_bf16_eligible(g), HN_E_UNSUPPORTED, "gemm_bf16: M=%d N=%d K=%d not eligible", g.M, g.N, g.K);
  HN_REQUIRE((((uintptr_t)stage | (uintptr_t)Ab) & 15) == 0, HN_E_WORKSPACE, "gemm_bf16: staging buffers must be 16-byte aligned");
  HN_REQUIRE(g.pro == PRO_NONE || (g.gamma && g.beta), HN_E_NULL, "gemm_bf16: prologue needs gamma and beta");
  const int Kp = gemm_bf16_pitch(g.K);
  float *cb = stage;                                        // N floats (N % 128 == 0: the image behind it stays 16-byte aligned)
  uint16_t *Wb = (uint16_t *)(stage + g.N);
  const bool affine = g.pro == PRO_AFFINE;
  hipLaunchKernelGGL(gemm_bf16_stage_kernel, dim3((unsigned)g.N), dim3(256), 0, s, g.W, g.ldw, affine ? g.gamma : nullptr,
                     affine ? g.beta : nullptr, g.K, Kp, (unsigned *)Wb, cb);
  HN_LAUNCH_CHECK("gemm_bf16_stage");
  Bf16GemmArgs a;
  a.Ab = Ab; a.Wb = Wb; a.cb = cb; a.C = g.C; a.ldc = g.ldc; a.alpha = g.alpha; a.M = g.M; a.Kp = Kp;
  a.ntm = (g.M + HM - 1) / HM; a.ntn = g.N / HNT;
  a.abl = 0;
#ifdef HN_GEMM_BF16_BENCH
  if (const char *e = tuning_env("HN_BF16_ABL")) a.abl = atoi(e);
#endif
  a.K16 = K16; a.V16 = V16; a.inner = g.N / 2; a.heads = g.N / 128; a.tokens = tokens; a.np = (tokens + 31) / 32 * 32;
  if (img && tokens % 32 != 0) {
    hipLaunchKernelGGL(v_tail_zero_kernel, dim3((unsigned)((g.M / tokens) * a.heads)), dim3(256), 0, s, (u32x4 *)V16, a.np / 32);
    HN_LAUNCH_CHECK("v_tail_zero");
  }
  const long blocks = (long)((a.ntm + 7) / 8) * 8 * a.ntn;
  HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "gemm_bf16: grid too large");
  constexpr int lds_bytes = 2 * 4 * PLANE;      // 66 048: above the 64 KB a kernel gets without asking
  static bool configured[64] = {};
  int dev = 0;
  HN_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    HN_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    HN_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  static const bool no_glds = tuning_env("HN_NO_GLDS_GEMM") != nullptr;      // development switch: the register-staged kernel
  if (!no_glds && (long)g.M * Kp * 2 < (1L << 31)) {                     // (the DMA descriptors address a tile's rows with 32-bit offsets anyway;
    if (img) hipLaunchKernelGGL(gemm_bf16_glds_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);      //  the bound keeps row * pitch in 31 bits)
    else hipLaunchKernelGGL(gemm_bf16_glds_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  } else if (img) hipLaunchKernelGGL(gemm_bf16_kernel<true>, dim3((unsigned)blocks), dim3(256), lds_bytes, s, a);
  else hipLaunchKernelGGL(gemm_bf16_kernel<false>, dim3((unsigned)blocks), dim3(256), lds_bytes, s, a);
  HN_LAUNCH_CHECK("gemm_bf16");
  return HN_OK;
}

}  // namespace hn
