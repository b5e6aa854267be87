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
// What streams is the WEIGHTS: every GEMM of the chain is cut into 128 x 32 blocks (16 KB) consumed in one fixed order, and a
// single software pipeline runs across all stages -- global loads four blocks ahead (two register stages), a ring of three
// LDS buffers, B fragments of block t+1 read while the MFMAs of block t issue, one barrier per block.  The separate launches
// this replaces were each a latency chain of load -> LayerNorm -> LDS -> 16..64 MFMAs -> store per workgroup (2-4x their MFMA
// time); here the MFMA pipe of every SIMD sees 16 MFMAs per 16 KB of weights back to back.
//
// Bound: fp32 MFMA (v_mfma_f32_16x16x4_f32).  Per 16-row workgroup and block: 16 MFMAs per wave = 512 cycles against 16 KB
// from L2 (32 B/clk/CU, half the L2->CU path).  Shapes: l_d = 128, hidden 512, K and N multiples of 32 / 128, rows % 16 == 0.
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
constexpr int LDS_FLOATS = 3 * WBLK + 16 * ATILE + 4 * ATILE + CR * XP;
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

__global__ __launch_bounds__(256) void latent_chain_kernel(const ChainArgs args) {
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
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  lf32 *lds = (lf32 *)lds_raw;              // float offsets into the one LDS allocation:
  constexpr int Wb = 0;                     // [3][WBLK]    weight ring
  constexpr int Abig = Wb + 3 * WBLK;       // [16][ATILE]  attention output tile (A of the out-projection), then the FF hidden tile
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
  const int lr = tid >> 3, lc = tid & 7;    // loader: rows lr + {0, 32, 64, 96}, 16-byte slot lc of the 32-float k-chunk
  const int lsw = (lc ^ (lr & 7)) * 4;      // (lr + 32 i) & 7 == lr & 7
  int lb = 0;                               // next block to request

  auto issue = [&](float4 (&r)[4]) {        // global loads of block lb (past the end: the last block again, never consumed)
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
    const int off = (lr * ldw + lc * 4) * 4;                     // per-lane part
    const int soff = __builtin_amdgcn_readfirstlane((rb * ldw + k * WK) * 4);      // block part (scalar)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = hn_buffer_load_x4(rs, off + i * 32 * ldw * 4, soff, 0);
      r[i] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  auto store_block = [&](const float4 (&r)[4], int buf) {
    const int B = Wb + buf * WBLK;
#pragma unroll
    for (int i = 0; i < 4; ++i) lst4(lds, B + (lr + 32 * i) * WK + lsw, r[i]);
  };

  // ---- far loads first: two weight blocks, the x tile, the attention-output tile
  float4 R0[4], R1[4];
  issue(R0);
  issue(R1);
  {
    const int row = tid >> 4, l16 = tid & 15;
    const gf32 *xr = a_x_in + (long)(m0 + row) * CD + 8 * l16;
    float4 v0 = gld4(xr), v1 = gld4(xr + 4);
    if (a_head == 2) {                      // one-token cross block: the same output row for every latent row of a sample
      const gf32 *yr = a_y + (long)((m0 + row) / a_L) * CD + 8 * l16;
      const float4 y0 = gld4(yr), y1 = gld4(yr + 4);
      v0.x += y0.x; v0.y += y0.y; v0.z += y0.z; v0.w += y0.w;
      v1.x += y1.x; v1.y += y1.y; v1.z += y1.z; v1.w += y1.w;
    }
    lst4(lds, xs + row * XP + 8 * l16, v0);
    lst4(lds, xs + row * XP + 8 * l16 + 4, v1);
    if (a_head == 1) {
      const gf32 *orow = a_O + (long)(m0 + row) * a_ldo;
      for (int q = l16; q < a_inner_o / 4; q += 16) {       // 16-byte chunk q of the row: k-tile q >> 3, slot q & 7
        const float4 o = gld4(orow + 4 * q);
        lst4(lds, Abig + (q >> 3) * ATILE + row * WK + (((q & 7) ^ (row & 7)) * 4), o);
      }
    }
  }
  store_block(R0, 0);
  store_block(R1, 1);
  issue(R0);                                // blocks 2 and 3 in flight
  issue(R1);
  __syncthreads();

  // ---- consumer state
  int cur = 0, nxt = 1, nn = 2;             // LDS buffers of blocks t, t+1, t+2
  float4 fa0[2], fb0[4], fa1[2], fb1[4];    // fragments of the even / odd block of a pair: A slots s2 = 0 / 1; B slots x column tiles
  const int brow = wave * 32 + fi;          // B rows brow (tile 0) and brow + 16 (tile 1); (row & 7) == (fi & 7)
  auto read_b = [&](float4 (&f)[4], int buf) {
    const int B = Wb + buf * WBLK;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int sl = ((4 * s2 + fg) ^ (fi & 7)) * 4;
      f[2 * s2] = lld4(lds, B + brow * WK + sl);
      f[2 * s2 + 1] = lld4(lds, B + (brow + 16) * WK + sl);
    }
  };
  auto read_a = [&](float4 (&f)[2], int A, int kt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) f[s2] = lld4(lds, A + kt * ATILE + fi * WK + (((4 * s2 + fg) ^ (fi & 7)) * 4));
  };
  read_b(fb0, 0);

  // One block.  `R` holds block t+2 (requested two steps ago): park it in LDS, re-use the registers for the request of
  // block t+4, fetch the fragments of block t+1 while the 16 MFMAs of block t issue, one barrier.  Straight-line code.
  auto step = [&](float4 (&R)[4], const float4 (&fa)[2], const float4 (&fb)[4], float4 (&fan)[2], float4 (&fbn)[4], int A,
                  int kt_next, f32x4 &c0, f32x4 &c1) {
    store_block(R, nn);
    issue(R);
    read_b(fbn, nxt);
    read_a(fan, A, kt_next);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float4 av = fa[s2], b0 = fb[2 * s2], b1 = fb[2 * s2 + 1];
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1.x, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0.y, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0.z, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1.z, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0.w, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1.w, c1, 0, 0, 0);
    }
    __syncthreads();
    const int t3 = cur; cur = nxt; nxt = nn; nn = t3;
  };
  // the k loop of one 128-column chunk, two blocks per iteration (nk is even): register sets and fragment sets alternate,
  // nothing is copied.  The A fragments requested by the last step (k-tile 0 again) serve the next chunk of the same stage.
  auto run_chunk = [&](int A, int nk, f32x4 &c0, f32x4 &c1) {
    for (int kc = 0; kc < nk; kc += 2) {
      step(R0, fa0, fb0, fa1, fb1, A, kc + 1, c0, c1);
      step(R1, fa1, fb1, fa0, fb0, A, kc + 2 == nk ? 0 : kc + 2, c0, c1);
    }
  };
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // LayerNorm of the x tile -> Ahat (A layout); gamma == NULL: the rows as they are
  auto layer_norm = [&](const gf32 *gamma, const gf32 *beta) {
    const int row = tid >> 4, l16 = tid & 15;
    float4 v0 = lld4(lds, xs + row * XP + 8 * l16), v1 = lld4(lds, xs + row * XP + 8 * l16 + 4);
    if (gamma) {
      float s = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
      const float mu = s * (1.0f / CD);
      v0.x -= mu; v0.y -= mu; v0.z -= mu; v0.w -= mu; v1.x -= mu; v1.y -= mu; v1.z -= mu; v1.w -= mu;
      float q = ((v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w)) + ((v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w));
      q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
      const float rs = 1.0f / sqrtf(q * (1.0f / CD) + 1e-5f);
      const float4 g0 = gld4(gamma + 8 * l16), g1 = gld4(gamma + 8 * l16 + 4);
      const float4 b0 = gld4(beta + 8 * l16), b1 = gld4(beta + 8 * l16 + 4);
      v0.x = v0.x * rs * g0.x + b0.x; v0.y = v0.y * rs * g0.y + b0.y; v0.z = v0.z * rs * g0.z + b0.z; v0.w = v0.w * rs * g0.w + b0.w;
      v1.x = v1.x * rs * g1.x + b1.x; v1.y = v1.y * rs * g1.y + b1.y; v1.z = v1.z * rs * g1.z + b1.z; v1.w = v1.w * rs * g1.w + b1.w;
    }
    const int kt = l16 >> 2, s0 = (2 * l16) & 7;
    lst4(lds, Ahat + kt * ATILE + row * WK + ((s0 ^ (row & 7)) * 4), v0);
    lst4(lds, Ahat + kt * ATILE + row * WK + (((s0 + 1) ^ (row & 7)) * 4), v1);
  };
  // accumulator element r of column tile T: row 4 fg + r, column wave * 32 + 16 T + fi of the chunk
  const int ncol = wave * 32 + fi;

  // ================= stage OUT: x += LeakyReLU(O W_out^T + b_out) =================
  if (nk_out) {
    f32x4 c0 = zero, c1 = zero;
    read_a(fa0, Abig, 0);
    run_chunk(Abig, nk_out, c0, c1);
    const float bv0 = gld1(a_b_out + ncol), bv1 = gld1(a_b_out + ncol + 16);
    const float v0[4] = {c0.x, c0.y, c0.z, c0.w}, v1[4] = {c1.x, c1.y, c1.z, c1.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * fg + r;
      float p = v0[r] + bv0, q = v1[r] + bv1;
      p = p > 0.0f ? p : 0.01f * p;
      q = q > 0.0f ? q : 0.01f * q;
      lds[xs + row * XP + ncol] += p;
      lds[xs + row * XP + ncol + 16] += q;
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
      const int h0 = hc * WN + ncol, h1 = h0 + 16;
      const float ba0 = gld1(a_b1 + h0), ba1 = gld1(a_b1 + h1), bg0 = gld1(a_b1 + CHID + h0), bg1 = gld1(a_b1 + CHID + h1);
      const float va0[4] = {a0.x, a0.y, a0.z, a0.w}, va1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float vg0[4] = {g0.x, g0.y, g0.z, g0.w}, vg1[4] = {g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * fg + r;
        const float gt0 = vg0[r] + bg0, gt1 = vg1[r] + bg1;
        const float h_0 = (va0[r] + ba0) * (a_gate == HN_GATE_SELU ? selu_f(gt0) : gelu_erf(gt0));
        const float h_1 = (va1[r] + ba1) * (a_gate == HN_GATE_SELU ? selu_f(gt1) : gelu_erf(gt1));
        lds[Abig + (h0 >> 5) * ATILE + row * WK + ((((h0 & 31) >> 2) ^ (row & 7)) * 4) + (h0 & 3)] = h_0;
        lds[Abig + (h1 >> 5) * ATILE + row * WK + ((((h1 & 31) >> 2) ^ (row & 7)) * 4) + (h1 & 3)] = h_1;
      }
    }
    __syncthreads();
    {
      f32x4 c0 = zero, c1 = zero;
      read_a(fa0, Abig, 0);
      run_chunk(Abig, CHID / WK, c0, c1);
      const float bv0 = gld1(a_b2 + ncol), bv1 = gld1(a_b2 + ncol + 16);
      const float v0[4] = {c0.x, c0.y, c0.z, c0.w}, v1[4] = {c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * fg + r;
        lds[xs + row * XP + ncol] += v0[r] + bv0;
        lds[xs + row * XP + ncol + 16] += v1[r] + bv1;
      }
    }
    __syncthreads();
  }

  // ---- x is final: hand it to the next attention block (its input / residual, and the trace slot of hn_attn_probs)
  if (a_x_out) {
    const int row = tid >> 4, l16 = tid & 15;
    gf32 *xo = a_x_out + (long)(m0 + row) * CD + 8 * l16;
    gst4(xo, lld4(lds, xs + row * XP + 8 * l16));
    gst4(xo + 4, lld4(lds, xs + row * XP + 8 * l16 + 4));
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
      const float v0[4] = {c0.x, c0.y, c0.z, c0.w}, v1[4] = {c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gf32 *crow = C + (long)(m0 + 4 * fg + r) * ldc + j * WN + ncol;
        gst1(crow, al * v0[r]);
        gst1(crow + 16, al * v1[r]);
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
    HN_REQUIRE(a.inner_o > 0 && a.inner_o % (2 * WK) == 0 && a.inner_o <= 16 * WK && a.ldo % 4 == 0 && al16(a.O) && al16(a.w_out), HN_E_SHAPE,
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
  // one-time opt-in per device to > 64 KB of dynamic LDS (a function attribute; setting it twice is harmless, so the flag
  // needs no lock)
  static bool configured[64] = {};
  const int lds_bytes = LDS_FLOATS * (int)sizeof(float);
  int dev = 0;
  HN_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    HN_HIP_CHECK(hipFuncSetAttribute((const void *)latent_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  hipLaunchKernelGGL(latent_chain_kernel, dim3(a.rows / CR), dim3(256), lds_bytes, s, a);
  HN_LAUNCH_CHECK("latent_chain");
  return HN_OK;
}

}  // namespace hn
