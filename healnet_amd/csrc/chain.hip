// latent_chain_kernel -- the row-local part of the latent side as ONE persistent launch per chain
// (SURVEY.md 7 step 6 / 8(b) hn_latent_block_fwd; replaces healnet/models/healnet.py:236-245 between two attention cores):
//
//   x1 = x + LeakyReLU(O W_out^T + b_out)        attention out-projection (:426, :385) + residual (:236 / :244)
//        | x + y[sample]                          ... or the row-broadcast output of a one-token cross block
//   x2 = x1 + FF(LN(x1))                          PreNorm + gated feed-forward (:313-321, :339-351) + residual (:237 / :245)
//   Q  = alpha * LN'(x2) W_q^T,  KV = LN'(x2) W_kv^T      projections of the NEXT attention block (:403-405)
//
// Every step is local to a latent row, so a workgroup owns 16 rows (b * l_c / 16 workgroups: 256 at cfg2 b = 32, one per CU)
// and walks the whole chain with its rows in LDS: the x tile, its LayerNorm-ed image, the 16 x 512 feed-forward hidden tile.
// What streams is the WEIGHTS, and they never touch LDS: eight waves split each 128-column chunk of a GEMM (16 columns per
// wave), and every wave fetches its own B fragments -- 16 weight rows x 64 bytes per buffer load, already in the MFMA's
// operand layout -- straight into a four-deep register ring that runs ahead across chunk and stage boundaries (one flat block
// order for the whole chain).  No barrier inside a GEMM; waves only meet where a stage hands its tile to the next.
//   (First version: weights staged through a three-buffer LDS ring with one barrier per 16 KB block.  At 16 rows per
//   workgroup every weight byte feeds only 16 FMAs, so that ring moved 16 KB in and 16 KB out of LDS per 512 MFMA cycles:
//   the ds_write path alone (~75 B/clk/CU) ate 40 % of a step, a step took 2.1x its MFMA time, 8 waves instead of 4 changed
//   nothing -- measured with the NOLOAD / NOLDSW / NOMFMA variants of tools/bench_chain.py.)
//
// Bound (measured, cfg2 b = 32, tools/bench_chain.py variants): the L2 -> L1 fill rate.  With the loads removed a 128 x 32
// weight block costs 518 cycles per workgroup (8 MFMAs per wave x two waves per SIMD = 512: the matrix work is AT its bound);
// with them 1100 cycles = 15 B/clk/CU, 9 TB/s over the chip, at a 92 % L2 hit rate (TCC_HIT / TCC_REQ) -- the same 6-9 TB/s
// the 2-D tiled latent GEMMs reach.  Every CU has to pull ALL weights of the chain (1.1-2 MB) through its L1, so a chain
// costs about weights / 38 GB/s + 13 us whatever the tile height up to 256 workgroups; neither an L2 warm-up pass, nor
// rotating the chunk order between the workgroups of an XCD, nor eight instead of four waves changed that.  Larger batches
// (more rows per CU) move the same kernel towards the MFMA bound.
// Shapes: l_d = 128, hidden 512, K a multiple of 128, N a multiple of 128, rows % 16 == 0.
#include "common.h"

namespace hn {

namespace {

constexpr int CR = 16;                  // rows per workgroup
constexpr int CD = 128;                 // latent width
constexpr int CHID = 512;               // feed-forward hidden width (4 * CD)
constexpr int WN = 128, WK = 32;        // weight block: 128 output columns x 32 k
constexpr int WBLK = WN * WK;           // floats per block (16 KB)
constexpr int ATILE = CR * WK;          // floats per A k-tile (16 rows x 32 k, 16-byte slots XOR-swizzled by row & 7)
constexpr int XP = 132;                 // pitch of the x tile
constexpr int LDS_FLOATS = 16 * ATILE + 4 * ATILE + CR * XP;       // 48.25 KB
enum { CS_OUT = 0, CS_FF1 = 1, CS_FF2 = 2, CS_Q = 3, CS_KV = 4, CS_END = 5 };

// Pointers that arrive inside the argument struct are generic: hipcc emits flat_load / flat_store for them, and with flat
// operations in flight its wait-count insertion falls back to "wait for everything" in front of every use of a prefetched
// register.  Everything outside the weight stream therefore goes through explicit global-address-space accesses.
typedef float __attribute__((address_space(1))) gf32;
typedef f32x4 __attribute__((address_space(1))) gf32x4;
__device__ __forceinline__ float4 gld4(const gf32 *p) {
  const f32x4 v = *(const gf32x4 *)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float gld1(const gf32 *p) { return *p; }
__device__ __forceinline__ void gst4(gf32 *p, const float4 &v) {
  f32x4 t = {v.x, v.y, v.z, v.w};
  *(gf32x4 *)p = t;
}
__device__ __forceinline__ void gst1(gf32 *p, float v) { *p = v; }
// ... and every LDS access through address space 3 with integer offsets (a generic pointer that the compiler cannot trace back
// to the LDS symbol becomes a flat access, which counts against BOTH wait counters)
typedef float __attribute__((address_space(3))) lf32;
typedef f32x4 __attribute__((address_space(3))) lf32x4;
__device__ __forceinline__ float4 lld4(const lf32 *base, int off) {
  const f32x4 v = *(const lf32x4 *)(base + off);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lst4(lf32 *base, int off, const float4 &v) {
  f32x4 t = {v.x, v.y, v.z, v.w};
  *(lf32x4 *)(base + off) = t;
}

__device__ __forceinline__ float selu_f(float x) {
  const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
  return scale * (x > 0.0f ? x : alpha * expm1f(x));
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

}  // namespace

__global__ __launch_bounds__(512) void latent_chain_kernel(const ChainArgs args) {
  // Every field of the by-value argument struct is unpacked ONCE into a local (pointers as global-address-space pointers): the
  // lambdas below capture locals only.  Capturing the struct itself keeps a copy of it in scratch, and the compiler then
  // turns "select among pointers" into loads from a selected scratch address followed by flat accesses.
  const gf32 *const a_x_in = (const gf32 *)args.x_in;
  const gf32 *const a_O = (const gf32 *)args.O;
  const gf32 *const a_w_out = (const gf32 *)args.w_out;
  const gf32 *const a_b_out = (const gf32 *)args.b_out;
  const gf32 *const a_y = (const gf32 *)args.y;
  const gf32 *const a_f_nw = (const gf32 *)args.f_nw;
  const gf32 *const a_f_nb = (const gf32 *)args.f_nb;
  const gf32 *const a_w1 = (const gf32 *)args.w1;
  const gf32 *const a_b1 = (const gf32 *)args.b1;
  const gf32 *const a_w2 = (const gf32 *)args.w2;
  const gf32 *const a_b2 = (const gf32 *)args.b2;
  const gf32 *const a_p_nw = (const gf32 *)args.p_nw;
  const gf32 *const a_p_nb = (const gf32 *)args.p_nb;
  const gf32 *const a_wq = (const gf32 *)args.wq;
  const gf32 *const a_wkv = (const gf32 *)args.wkv;
  gf32 *const a_x_out = (gf32 *)args.x_out;
  gf32 *const a_Q = (gf32 *)args.Q;
  gf32 *const a_KV = (gf32 *)args.KV;
  const int a_rows = args.rows;
  const int a_L = args.L;
  const int a_head = args.head;
  const int a_ldo = args.ldo;
  const int a_inner_o = args.inner_o;
  const int a_has_ff = args.has_ff;
  const int a_gate = args.gate;
  const int a_nq = args.nq;
  const int a_nkv = args.nkv;
  const int a_ldq = args.ldq;
  const int a_ldkv = args.ldkv;
  const float a_alpha_q = args.alpha_q;
  __shared__ __attribute__((aligned(16))) float lds_raw[LDS_FLOATS];
  lf32 *lds = (lf32 *)lds_raw;              // float offsets into the one LDS allocation:
  constexpr int Abig = 0;                   // [16][ATILE]  attention output tile (A of the out-projection), then the FF hidden tile
  constexpr int Ahat = Abig + 16 * ATILE;   // [4][ATILE]   LayerNorm-ed x (A of FF1 / of the projections)
  constexpr int xs = Ahat + 4 * ATILE;      // [CR][XP]     the x tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fg = lane >> 4, fi = lane & 15;
  const int m0 = blockIdx.x * CR;

  // ---- the block stream: stage -> n-chunk -> k-chunk, identical for the loader and the consumer.  The loader addresses
  // block number `lb` of the whole chain with branch-free scalar arithmetic (a branchy iterator breaks the step into many
  // basic blocks and the compiler then waits for ALL outstanding loads in front of every LDS store).
  const int nk_out = a_head == 1 ? a_inner_o / WK : 0;           // out-projection: ONE 128-column chunk of nk_out k-chunks
  const int nq_ch = a_nq / WN, nkv_ch = a_nkv / WN;
  const int e0 = nk_out;                                         // first block of FF1
  const int e1 = e0 + (a_has_ff ? 8 * (CD / WK) : 0);            //                FF2
  const int e2 = e1 + (a_has_ff ? CHID / WK : 0);                //                Q
  const int e3 = e2 + nq_ch * (CD / WK);                         //                KV
  const int nblocks = e3 + nkv_ch * (CD / WK);
  // (the descriptor words pass through readfirstlane HERE: as plain loads of the by-value argument struct the compiler turns the
  // per-block "select among five descriptors" into a load from a selected address of a scratch copy of the struct)
  auto pinned = [](const gf32 *base, unsigned bytes) {
    i32x4 r = make_rsrc((const void *)(unsigned long long)base, bytes);
    r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y); r.z = __builtin_amdgcn_readfirstlane(r.z);
    return r;
  };
  const i32x4 rs_out = pinned(a_w_out, a_head == 1 ? (unsigned)((long)CD * a_inner_o * 4) : 0u);
  const i32x4 rs_w1 = pinned(a_w1, a_has_ff ? (unsigned)(2 * CHID * CD * 4) : 0u);
  const i32x4 rs_w2 = pinned(a_w2, a_has_ff ? (unsigned)(CD * CHID * 4) : 0u);
  const i32x4 rs_q = pinned(a_wq, (unsigned)((long)a_nq * CD * 4));
  const i32x4 rs_kv = pinned(a_wkv, (unsigned)((long)a_nkv * CD * 4));
  const int inner_o = __builtin_amdgcn_readfirstlane(a_inner_o);
  int lb = 0;                               // next block to request

  auto issue = [&](float4 (&r)[2]) {        // global loads of block lb (past the end: the last block again, never consumed)
    const int bi = min(lb, nblocks - 1);
    ++lb;
    // stage of the block as 0 / -1 masks, everything selected with AND / OR: a chain of ?: keyed by the stage is recognised as
    // a switch and lowered to lookup tables IN SCRATCH (the five descriptors, the stage offsets), read back through flat loads
    const int m_out = -(int)(bi < e0), m_ff1 = -(int)(bi >= e0 && bi < e1), m_ff2 = -(int)(bi >= e1 && bi < e2);
    const int m_q = -(int)(bi >= e2 && bi < e3), m_kv = -(int)(bi >= e3);
    const int local = bi - ((e0 & m_ff1) | (e1 & m_ff2) | (e2 & m_q) | (e3 & m_kv));
    const int one_chunk = m_out | m_ff2;                         // stages with a single 128-column chunk: k = local
    const int j = (local >> 2) & ~one_chunk;                     // chunk within the stage (4 k-chunks per chunk at K = 128)
    const int k = (local & one_chunk) | (local & 3 & ~one_chunk);
    const int ldw = (inner_o & m_out) | (CHID & m_ff2) | (CD & ~(m_out | m_ff2));
    const int rb = ((((j & 1) * CHID + (j >> 1) * WN) & m_ff1) | ((j * WN) & ~m_ff1));   // FF1: value chunk, then its gate chunk
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((rs_out.x & m_out) | (rs_w1.x & m_ff1) | (rs_w2.x & m_ff2) | (rs_q.x & m_q) | (rs_kv.x & m_kv));
    rs.y = __builtin_amdgcn_readfirstlane((rs_out.y & m_out) | (rs_w1.y & m_ff1) | (rs_w2.y & m_ff2) | (rs_q.y & m_q) | (rs_kv.y & m_kv));
    rs.z = __builtin_amdgcn_readfirstlane((rs_out.z & m_out) | (rs_w1.z & m_ff1) | (rs_w2.z & m_ff2) | (rs_q.z & m_q) | (rs_kv.z & m_kv));
    rs.w = 0x00020000;
    // this wave's B fragments of the block: weight row 16 wave + fi of the chunk, k = 16 s2 + 4 fg .. + 3 (component s of the
    // 16 bytes feeds MFMA k-step s: the A fragments in LDS use the same permutation of the contraction index)
    const int off = ((wave * 16 + fi) * ldw + 4 * fg) * 4;
    const int soff = __builtin_amdgcn_readfirstlane((rb * ldw + k * WK) * 4);      // block part (scalar)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const f32x4 v = hn_buffer_load_x4(rs, off + s2 * 64, soff, 0);
      r[s2] = make_float4(v.x, v.y, v.z, v.w);
    }
  };

  // ---- far loads first: four weight blocks per wave, the x tile, the attention-output tile
  float4 B0[2], B1[2], B2[2], B3[2];
  issue(B0);
  issue(B1);
  issue(B2);
  issue(B3);
  {
    const int row = tid >> 5, l32 = tid & 31;           // 32 lanes per row, one 16-byte chunk each
    float4 v0 = gld4(a_x_in + (long)(m0 + row) * CD + 4 * l32);
    if (a_head == 2) {                      // one-token cross block: the same output row for every latent row of a sample
      const float4 y0 = gld4(a_y + (long)((m0 + row) / a_L) * CD + 4 * l32);
      v0.x += y0.x; v0.y += y0.y; v0.z += y0.z; v0.w += y0.w;
    }
    lst4(lds, xs + row * XP + 4 * l32, v0);
    if (a_head == 1) {
      const gf32 *orow = a_O + (long)(m0 + row) * a_ldo;
      for (int q = l32; q < a_inner_o / 4; q += 32) {       // 16-byte chunk q of the row: k-tile q >> 3, slot q & 7
        const float4 o = gld4(orow + 4 * q);
        lst4(lds, Abig + (q >> 3) * ATILE + row * WK + (((q & 7) ^ (row & 7)) * 4), o);
      }
    }
  }
  __syncthreads();

  // ---- consumer state: A fragments of the even / odd block of a pair (slots s2 = 0 / 1)
  float4 fa0[2], fa1[2];
  auto read_a = [&](float4 (&f)[2], int A, int kt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) f[s2] = lld4(lds, A + kt * ATILE + fi * WK + (((4 * s2 + fg) ^ (fi & 7)) * 4));
  };

  // One block: the A fragments of the next block, the 8 MFMAs of this one, then the register slot is re-used for the request
  // of block t + 4.  Straight-line code, no barrier.
  auto step = [&](float4 (&Bq)[2], const float4 (&fa)[2], float4 (&fan)[2], int A, int kt_next, f32x4 &c0, f32x4 &c1) {
    read_a(fan, A, kt_next);
#ifdef CHAIN_EXP_NOMFMA
    c0.x += fa[0].x + Bq[0].x + fa[1].y + Bq[1].x;
#else
    // two accumulators (k-slots 0 and 1) so that consecutive MFMAs never depend on each other; they are summed in the epilogue
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].x, Bq[0].x, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].x, Bq[1].x, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].y, Bq[0].y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].y, Bq[1].y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].z, Bq[0].z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].z, Bq[1].z, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].w, Bq[0].w, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].w, Bq[1].w, c1, 0, 0, 0);
#endif
#ifndef CHAIN_EXP_NOLOAD
    issue(Bq);
#endif
  };
  // the k loop of one 128-column chunk, four blocks per iteration (nk is a multiple of 4): register slots and A fragment
  // sets alternate, nothing is copied.  The A fragments requested by the last step (k-tile 0 again) serve the next chunk of
  // the same stage.
  auto run_chunk = [&](int A, int nk, f32x4 &c0, f32x4 &c1) {
#ifdef CHAIN_EXP_NOSTEP
    nk = 0;
#endif
    for (int kc = 0; kc < nk; kc += 4) {
      step(B0, fa0, fa1, A, kc + 1, c0, c1);
      step(B1, fa1, fa0, A, kc + 2, c0, c1);
      step(B2, fa0, fa1, A, kc + 3, c0, c1);
      step(B3, fa1, fa0, A, kc + 4 == nk ? 0 : kc + 4, c0, c1);
    }
  };
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // LayerNorm of the x tile -> Ahat (A layout); gamma == NULL: the rows as they are
  auto layer_norm = [&](const gf32 *gamma, const gf32 *beta) {
    const int row = tid >> 5, l32 = tid & 31;
    float4 v = lld4(lds, xs + row * XP + 4 * l32);
    if (gamma) {
      float sm = (v.x + v.y) + (v.z + v.w);
      sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 4); sm += __shfl_xor(sm, 8); sm += __shfl_xor(sm, 16);
      const float mu = sm * (1.0f / CD);
      v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
      float q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8); q += __shfl_xor(q, 16);
      const float rs = 1.0f / sqrtf(q * (1.0f / CD) + 1e-5f);
      const float4 g0 = gld4(gamma + 4 * l32), b0 = gld4(beta + 4 * l32);
      v.x = v.x * rs * g0.x + b0.x; v.y = v.y * rs * g0.y + b0.y; v.z = v.z * rs * g0.z + b0.z; v.w = v.w * rs * g0.w + b0.w;
    }
    lst4(lds, Ahat + (l32 >> 3) * ATILE + row * WK + (((l32 & 7) ^ (row & 7)) * 4), v);   // k = 4 l32: k-tile l32 >> 3, slot l32 & 7
  };
  // accumulator element r: row 4 fg + r, column 16 wave + fi of the 128-column chunk
  const int ncol = wave * 16 + fi;

  // ================= stage OUT: x += LeakyReLU(O W_out^T + b_out) =================
  if (nk_out) {
    f32x4 c0 = zero, c1 = zero;
    read_a(fa0, Abig, 0);
    run_chunk(Abig, nk_out, c0, c1);
    const float bv = gld1(a_b_out + ncol);
    const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = v[r] + bv;
      p = p > 0.0f ? p : 0.01f * p;
      lds[xs + (4 * fg + r) * XP + ncol] += p;
    }
    __syncthreads();
  }

  // ================= stages FF1 / FF2: x += (a * gate(g)) W2^T + b2,  [a | g] = LN(x) W1^T + b1 =================
  if (a_has_ff) {
    layer_norm(a_f_nw, a_f_nb);
    __syncthreads();
    read_a(fa0, Ahat, 0);
    for (int hc = 0; hc < 4; ++hc) {
      f32x4 a0 = zero, a1 = zero, g0 = zero, g1 = zero;
      run_chunk(Ahat, CD / WK, a0, a1);
      run_chunk(Ahat, CD / WK, g0, g1);
      const int h0 = hc * WN + ncol;
      const float ba = gld1(a_b1 + h0), bg = gld1(a_b1 + CHID + h0);
      const float va[4] = {a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w};
      const float vg[4] = {g0.x + g1.x, g0.y + g1.y, g0.z + g1.z, g0.w + g1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * fg + r;
        const float gt = vg[r] + bg;
        const float hv = (va[r] + ba) * (a_gate == HN_GATE_SELU ? selu_f(gt) : gelu_erf(gt));
        lds[Abig + (h0 >> 5) * ATILE + row * WK + ((((h0 & 31) >> 2) ^ (row & 7)) * 4) + (h0 & 3)] = hv;
      }
    }
    __syncthreads();
    {
      f32x4 c0 = zero, c1 = zero;
      read_a(fa0, Abig, 0);
      run_chunk(Abig, CHID / WK, c0, c1);
      const float bv = gld1(a_b2 + ncol);
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[xs + (4 * fg + r) * XP + ncol] += v[r] + bv;
    }
    __syncthreads();
  }

  // ---- x is final: hand it to the next attention block (its input / residual, and the trace slot of hn_attn_probs)
  if (a_x_out) {
    const int row = tid >> 5, l32 = tid & 31;
    gst4(a_x_out + (long)(m0 + row) * CD + 4 * l32, lld4(lds, xs + row * XP + 4 * l32));
  }

  // ================= stages Q / KV: the next attention block's projections of LN'(x) =================
  if (nq_ch + nkv_ch > 0) {
    layer_norm(a_p_nw, a_p_nb);
    __syncthreads();
    read_a(fa0, Ahat, 0);
    for (int pj = 0; pj < nq_ch + nkv_ch; ++pj) {
      const bool isq = pj < nq_ch;
      const int j = isq ? pj : pj - nq_ch;
      f32x4 c0 = zero, c1 = zero;
      run_chunk(Ahat, CD / WK, c0, c1);
      gf32 *C = isq ? a_Q : a_KV;
      const long ldc = isq ? a_ldq : a_ldkv;
      const float al = isq ? a_alpha_q : 1.0f;
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#ifndef CHAIN_EXP_NOSTORE
        gst1(C + (long)(m0 + 4 * fg + r) * ldc + j * WN + ncol, al * v[r]);
#else
        if (al * v[r] == 1.2345f) gst1(C, 0.f);
#endif
      }
    }
  }
}

bool latent_chain_supported(int rows, int d, int hidden) { return d == CD && hidden == CHID && rows > 0 && rows % CR == 0; }

int launch_latent_chain(const ChainArgs &a, hipStream_t s) {
  HN_REQUIRE(a.rows > 0 && a.rows % CR == 0 && a.L > 0, HN_E_SHAPE, "latent_chain: rows=%d L=%d", a.rows, a.L);
  HN_REQUIRE(a.x_in, HN_E_NULL, "latent_chain: x_in is NULL");
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  if (a.head == 1) {
    HN_REQUIRE(a.O && a.w_out && a.b_out, HN_E_NULL, "latent_chain: out-projection operand is NULL");
    HN_REQUIRE(a.inner_o > 0 && a.inner_o % (4 * WK) == 0 && a.inner_o <= 16 * WK && a.ldo % 4 == 0 && al16(a.O) && al16(a.w_out), HN_E_SHAPE,
               "latent_chain: inner=%d ldo=%d", a.inner_o, a.ldo);
  } else if (a.head == 2) {
    HN_REQUIRE(a.y && al16(a.y), HN_E_NULL, "latent_chain: y is NULL / unaligned");
  }
  if (a.has_ff) {
    HN_REQUIRE(a.w1 && a.b1 && a.w2 && a.b2, HN_E_NULL, "latent_chain: feed-forward operand is NULL");
    HN_REQUIRE(al16(a.w1) && al16(a.w2) && (!a.f_nw || (al16(a.f_nw) && al16(a.f_nb))), HN_E_SHAPE, "latent_chain: unaligned operand");
  }
  HN_REQUIRE(a.nq >= 0 && a.nkv >= 0 && a.nq % WN == 0 && a.nkv % WN == 0, HN_E_SHAPE, "latent_chain: nq=%d nkv=%d", a.nq, a.nkv);
  HN_REQUIRE(a.nq == 0 || (a.wq && a.Q && al16(a.wq)), HN_E_NULL, "latent_chain: Q projection operand is NULL");
  HN_REQUIRE(a.nkv == 0 || (a.wkv && a.KV && al16(a.wkv)), HN_E_NULL, "latent_chain: KV projection operand is NULL");
  HN_REQUIRE(al16(a.x_in) && (!a.x_out || al16(a.x_out)) && (!a.p_nw || (al16(a.p_nw) && al16(a.p_nb))), HN_E_SHAPE,
             "latent_chain: unaligned operand");
  hipLaunchKernelGGL(latent_chain_kernel, dim3(a.rows / CR), dim3(512), 0, s, a);
  HN_LAUNCH_CHECK("latent_chain");
  return HN_OK;
}

}  // namespace hn
