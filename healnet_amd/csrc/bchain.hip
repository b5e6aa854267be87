// latent_bchain_kernel -- the BACKWARD of the row-local latent side as one launch per chain (the reverse of chain.hip; autograd of
// healnet/models/healnet.py:236-245 between two attention cores, i.e. of :313-321 PreNorm, :339-351 FeedForward, :385 / :426 the
// attention out-projection + LeakyReLU, :403-405 the q / kv projections).
//
// In backward order a chain is
//   P    (optional) projection backward of the attention block whose core backward has just run:
//            dxh = dQ W_q + dKV W_kv ;  G = LayerNorm'(dxh; x_p, gamma_p) + dY          (dY: the residual path around that block)
//   FF   (optional) the feed-forward block in front of it, recomputed from its taped input x_m:
//            xhat = LN(x_m) ; [a | g] = xhat W1^T + b1 ; dz = G W2 ; h = a act(g) ; dU = [dz act(g) | dz a act'(g)]
//            dxh = dU W1 ;  G = LayerNorm'(dxh; x_m, gamma_f) + G
//   OUT  (optional) the out-projection of the attention block in front of THAT:
//            dpre = G * LeakyReLU'(x_m - x_a) ; dO = dpre W_out
// Every step is local to a latent row: a workgroup owns 16 rows and keeps G, the x tiles, the 16 x 1024 dU tile in LDS; the
// weights (W1, and the transposes W2^T, W1^T, W_out^T, W_q^T, W_kv^T staged once per backward by transpose_multi_kernel) stream
// through the per-wave rings of chain.hip (same block table, same pinned step).  What leaves for HBM is exactly what the
// WEIGHT-gradient products need -- h, dU, xhat, G_ff, dpre (contraction over the rows: gemm_tn_lds_multi_kernel, one launch per
// chain) -- plus dO for the attention core backward, the new G, and per-workgroup partial sums of the LayerNorm parameter
// gradients (folded by splitk_reduce_multi_kernel in a fixed order: bitwise reproducible).
// Replaces, per feed-forward + attention pair, ~25 launches of 5-26 us (profiles/r03_b_train_cfg2_b32_kernel_stats.csv:
// gemm_t32a / gemm_t32 / gemm_t32a2 / ln_fwd / ln_bwd / ln_param_reduce / glu_bwd / leaky_bwd and, through the batched
// weight-gradient launch, 4-5 gemm_tn_lds + splitk_reduce pairs).
//
// Shapes: l_d = 128, hidden 512, nq / nkv / inner_o multiples of 128 (nq + nkv <= 1536), rows % 16 == 0.
#include "common.h"
#include <stddef.h>
#include <stdlib.h>

namespace hn {

namespace {

#include "chain_common.h"

constexpr int BMAXBLK = 160;                      // P <= 48, FF 48 + 32, OUT <= 16 blocks
constexpr int B_Wr = 0;                           // [8 waves][WSLOT]   per-wave transpose slot of the weight stream
constexpr int B_Abig = B_Wr + 8 * WSLOT;          // [32][ATILE]        P: the dQ | dKV tile (two segments); FF: the dU tile
constexpr int B_Ahat = B_Abig + 32 * ATILE;       // [4][ATILE]         LN(x_m) (A of the a / g recompute)
constexpr int B_Adz = B_Ahat + 4 * ATILE;         // [4][ATILE]         G in A layout (A of dz = G W2), then dpre (A of dO)
constexpr int B_xs = B_Adz + 4 * ATILE;           // [CR][XP]           x tile of the current LayerNorm, then column-sum scratch
constexpr int B_xn = B_xs + CR * XP;              // [CR][XP]           its normalised image
constexpr int B_gs = B_xn + CR * XP;              // [CR][XP]           the running gradient G
constexpr int B_ts = B_gs + CR * XP;              // [CR][XP]           GEMM result tile dxh / per-wave store staging
constexpr int B_b1 = B_ts + CR * XP;              // b1 (1024)
constexpr int B_gf = B_b1 + 2 * CHID;             // gamma_f, beta_f (128 each)
constexpr int B_gp = B_gf + 256;                  // gamma_p (128)
constexpr int B_rs = B_gp + 128;                  // rstd of the 16 rows (+ pad)
constexpr int B_tbl = B_rs + 32;
constexpr int B_LDS_FLOATS = B_tbl + 2 * BMAXBLK;          // 34 560 floats = 135 KB: one workgroup per CU

}  // namespace

// EXT: as in chain.hip (false = the default model's shapes exactly; true adds valid widths, ragged rows, idle cluster workgroups, dropout)
template <bool EXT>
__global__ __launch_bounds__(512) void latent_bchain_kernel(const BChainArgs args) {
  // every field unpacked once into locals (see chain.hip: capturing the struct keeps a scratch copy and turns pointer selects
  // into flat accesses)
  const gf32 *const a_dy = (const gf32 *)args.dy;
  gf32 *const a_dx_out = (gf32 *)args.dx_out;
  const int a_has_p = args.has_p, a_nq = args.nq, a_nkv = args.nkv, a_lddq = args.lddq, a_lddkv = args.lddkv;
  const gf32 *const a_dQ = (const gf32 *)args.dQ;
  const gf32 *const a_dKV = (const gf32 *)args.dKV;
  const gf32 *const a_p_x = (const gf32 *)args.p_x;
  const gf32 *const a_p_nw = (const gf32 *)args.p_nw;
  const int a_has_ff = args.has_ff, a_gate = args.gate;
  const gf32 *const a_f_x = (const gf32 *)args.f_x;
  const gf32 *const a_f_nw = (const gf32 *)args.f_nw;
  const gf32 *const a_f_nb = (const gf32 *)args.f_nb;
  const gf32 *const a_b1 = (const gf32 *)args.b1;
  gf32 *const a_H = (gf32 *)args.H;
  gf32 *const a_dU = (gf32 *)args.dU;
  gf32 *const a_Xhat = (gf32 *)args.Xhat;
  gf32 *const a_dYff = (gf32 *)args.dYff;
  const int a_has_out = args.has_out, a_inner_o = args.inner_o, a_lddo = args.lddo;
  const gf32 *const a_o_x = (const gf32 *)args.o_x;
  gf32 *const a_dPre = (gf32 *)args.dPre;
  gf32 *const a_dO = (gf32 *)args.dO;
  gf32 *const a_lnpart = (gf32 *)args.lnpart;
  // staged (padded) models: valid widths (common.h)
  const int a_rows = args.rows;
  const int a_dv = EXT && args.dv > 0 ? args.dv : CD;
  const float inv_dv = 1.0f / (float)a_dv;
  const int a_q_cols = EXT && args.q_cols > 0 ? args.q_cols : args.nq;
  const int a_kv_cols = EXT && args.kv_cols > 0 ? args.kv_cols : args.nkv;
  const uint32_t d_thr = EXT ? args.ff_drop.thr : 0u;   // (the rest of the generator state is read from the argument segment where it is used)

  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  lf32 *lds = (lf32 *)lds_raw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fg = lane >> 4, fi = lane & 15;
  // Cluster mode (<= 128 row tiles; see chain.hip): C = 4 or 2 workgroups share a row tile.  P: each member contracts ITS quarter
  // of the dQ | dKV columns (partial dx_hat, exchange 1); FF: its hidden chunks -- a, g, dz, the matching h / dU columns, and the
  // matching k-tiles of dx_hat = dU W1 (partial, exchange 2); OUT: its share of the dO chunks.  Partials are summed in member
  // order by every member (identical bits everywhere); member 0 writes what is per row tile.
  const int a_C = args.cluster > 1 ? args.cluster : 1;
  const int ntiles = gridDim.x / a_C;
  int member_, tile_;
  cluster_decode((int)blockIdx.x, a_C, ntiles, args.split_order, member_, tile_);      // members of a tile: one XCD, consecutive in its dispatch order (chain_common.h)
  const int member = __builtin_amdgcn_readfirstlane(member_), tile = __builtin_amdgcn_readfirstlane(tile_);
  if (EXT && tile >= args.tiles) return;            // (cluster grids are rounded up to 8 tiles per member row: idle workgroups)
  const int my_chunks = 4 / a_C;
  const int m0 = tile * CR;
  const int row = tid >> 5, l32 = tid & 31;                  // row layout: 32 lanes per row, one 16-byte chunk each
  const long grow = (long)(m0 + row) * CD + 4 * l32;         // this thread's chunk of a (rows, 128) tensor

  // ---- the block stream of the chain
  const int ptot = a_has_p ? a_nq + a_nkv : 0;               // contraction length of P, in two segments of <= 1024 columns
  // (cluster: the member's contiguous share of the columns, one segment)
  // -- when the shares are whole 128-column groups; a short contraction (staged one-head models: 128 columns) is run in full by
  // every member instead, without an exchange
  const bool p_split = a_C > 1 && ptot % (a_C * WN) == 0;
  const int pc0 = p_split ? member * (ptot / a_C) : 0;
  const int pmine = p_split ? ptot / a_C : ptot;
  const int pseg1 = min(pmine, 1024), pseg2 = pmine - pseg1;
  const int nk1 = pseg1 / WK, nk2 = pseg2 / WK;
  const int my_out = a_has_out ? (a_inner_o / WN - member + a_C - 1) / a_C : 0;      // dO chunks j = member, member + C, ...
  const int eP = nk1 + nk2;
  const int eF1 = eP + (a_has_ff ? 12 * my_chunks : 0);      // per own hidden chunk: a, g (W1), dz (W2^T), 4 k-blocks each
  const int eF2 = eF1 + (a_has_ff ? 8 * my_chunks : 0);      // dxh = dU W1: W1^T (128, 1024), the k-tiles of the own chunks
  const int nblocks = max(1, eF2 + my_out * (CD / WK));      // dO = dpre W_out: W_out^T (inner_o, 128)
  if (tid < BMAXBLK) {
    const int bi = min(tid, nblocks - 1);
    const float *W;
    int ldw;
    if (bi < eP) {
      const int c0 = pc0 + bi * WK;
      if (c0 < a_nq) { W = args.wqT + c0; ldw = a_nq; } else { W = args.wkvT + (c0 - a_nq); ldw = a_nkv; }
    } else if (bi < eF1) {
      const int local = bi - eP, hci = local / 12, hc = member + hci * a_C, t = local - hci * 12, which = t >> 2, k = t & 3;
      ldw = CD;
      if (which == 0) W = args.w1 + (long)(hc * WN) * CD + k * WK;
      else if (which == 1) W = args.w1 + (long)(CHID + hc * WN) * CD + k * WK;
      else W = args.w2T + (long)(hc * WN) * CD + k * WK;
    } else if (bi < eF2) {
      // k-tiles of the own chunks: the da columns of chunk hc (tiles 4 hc .. 4 hc + 3), then its dg columns (16 + 4 hc ..)
      const int local = bi - eF1, hci = local >> 3, hc = member + hci * a_C, r = local & 7;
      const int kt = a_C == 1 ? local : (r < 4 ? 4 * hc + r : 16 + 4 * hc + (r - 4));      // one workgroup per tile: all 32 in order
      W = args.w1T + kt * WK; ldw = 2 * CHID;
    } else {
      const int local = bi - eF2, j = member + (local >> 2) * a_C, k = local & 3;
      W = args.woT + (long)(j * WN) * CD + k * WK; ldw = CD;
    }
    const unsigned long long addr = (unsigned long long)W;
    const unsigned long long desc = (addr & 0x0000ffffffffffffull) | ((unsigned long long)(ldw * 4) << 48);
    *(__attribute__((address_space(3))) unsigned long long *)(lds + B_tbl + 2 * tid) = desc;
  }
  const int r8 = lane >> 3, pos = lane & 7;
  const int wslot = B_Wr + wave * WSLOT + r8 * WK + ((pos ^ (r8 & 7)) * 4);
  const int vrow = wave * 16 + r8, pos16 = pos * 16;
  unsigned long long ent = 0;
  int tp = B_tbl;
  asm volatile("" : "+v"(tp));
  auto fetch_entry = [&]() {
    ent = *(const __attribute__((address_space(3))) unsigned long long *)(lds + tp);
    tp += 2;
  };
  auto load2 = [&](float4 (&r)[2]) {
    i32x4 rs;
    rs.x = (int)__builtin_amdgcn_readfirstlane((unsigned)ent);
    rs.y = (int)__builtin_amdgcn_readfirstlane((unsigned)(ent >> 32));
    rs.z = WN;
    rs.w = 0x00020000;
    const f32x4 v0 = hn_sbuffer_load_x4(rs, vrow, pos16, 0, 0), v1 = hn_sbuffer_load_x4(rs, vrow + 8, pos16, 0, 0);
    r[0] = make_float4(v0.x, v0.y, v0.z, v0.w);
    r[1] = make_float4(v1.x, v1.y, v1.z, v1.w);
  };
  auto issue = [&](float4 (&r)[2]) {
    load2(r);
    fetch_entry();
  };
  auto park = [&](const float4 (&r)[2]) {
    lst4(lds, wslot, r[0]);
    lst4(lds, wslot + 8 * WK, r[1]);
  };
  // 16-byte chunk q (columns 4q .. 4q + 3) of tile row r in the A layout: k-tile q >> 3, slot q & 7, XOR-swizzled by the row
  auto a_at = [&](int base, int r, int q) { return base + (q >> 3) * ATILE + r * WK + (((q & 7) ^ (r & 7)) * 4); };

  __syncthreads();                           // the block table
  fetch_entry();
  float4 Bp[2], B0[2], B1[2], B2[2], B3[2];
  issue(Bp);
  issue(B0);
  issue(B1);
  issue(B2);
  issue(B3);

  // ---- tiles: G <- dY, the first x tile, (P) the first segment of dQ | dKV; everything later is requested now as well so
  // that no HBM round trip sits between the stages (second P segment, x_m, x_a: 6 float4 per thread)
  float4 seg2[4], xm_pre = make_float4(0.f, 0.f, 0.f, 0.f), xa_pre = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    lst4(lds, B_gs + row * XP + 4 * l32, gld4(a_dy + grow));
    if (a_has_p) {
      lst4(lds, B_xs + row * XP + 4 * l32, gld4(a_p_x + grow));
      {
        float4 seg1[8];                      // all requests first (<= 8 passes of 128 columns), then the LDS stores
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int c = pc0 + min(128 * i, pseg1 - 128) + 4 * l32; // passes beyond the segment re-read its last one (not stored)
          const gf32 *src = c < a_nq ? a_dQ + (long)(m0 + row) * a_lddq + c : a_dKV + (long)(m0 + row) * a_lddkv + (c - a_nq);
          seg1[i] = make_float4(0.f, 0.f, 0.f, 0.f);               // (staged models: the contraction beyond the operand's columns is zero)
          if (!EXT || (c < a_nq ? c < a_q_cols : c - a_nq < a_kv_cols)) seg1[i] = gld4(src);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (128 * i < pseg1) lst4(lds, a_at(B_Abig, row, 32 * i + l32), seg1[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = pc0 + pseg1 + 128 * i + 4 * l32;
        seg2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pseg2 > 0 && c < ptot && (!EXT || c - a_nq < a_kv_cols)) seg2[i] = gld4(a_dKV + (long)(m0 + row) * a_lddkv + (c - a_nq));
      }
      if (a_p_nw) lst4(lds, B_gp + 4 * l32, gld4(a_p_nw + 4 * l32));    // (16 rows write the same 128 floats)
      if (a_has_ff) xm_pre = gld4(a_f_x + grow);
    } else if (a_has_ff) {
      lst4(lds, B_xs + row * XP + 4 * l32, gld4(a_f_x + grow));
    }
    if (a_has_ff) {
      if (tid < 256) lst4(lds, B_b1 + 4 * tid, gld4(a_b1 + 4 * tid));
      else if (tid < 288) { if (a_f_nw) lst4(lds, B_gf + 4 * (tid - 256), gld4(a_f_nw + 4 * (tid - 256))); }
      else if (tid < 320) { if (a_f_nb) lst4(lds, B_gf + 128 + 4 * (tid - 288), gld4(a_f_nb + 4 * (tid - 288))); }
    }
    if (a_has_out) xa_pre = gld4(a_o_x + grow);
  }
  park(Bp);
  __syncthreads();

  float4 fa0[2], fa1[2], fb0[2], fb1[2];
  auto read_a = [&](float4 (&f)[2], int A, int kt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) f[s2] = lld4(lds, A + kt * ATILE + fi * WK + (((4 * s2 + fg) ^ (fi & 7)) * 4));
  };
  auto read_b = [&](float4 (&f)[2]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) f[s2] = lld4(lds, B_Wr + wave * WSLOT + fi * WK + (((4 * s2 + fg) ^ (fi & 7)) * 4));
  };
  read_b(fb0);

  // one block (chain.hip's pinned step): 8 MFMAs of block t, park block t + 1, request block t + 5, fragments of t + 1
  auto step = [&](float4 (&Bq)[2], const float4 (&fa)[2], const float4 (&fb)[2], float4 (&fan)[2], float4 (&fbn)[2], int A, int kt_next,
                  f32x4 &c0, f32x4 &c1) {
#define CH_SB __builtin_amdgcn_sched_barrier(0)
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].x, fb[0].x, c0, 0, 0, 0); CH_SB;
    lst4(lds, wslot, Bq[0]); CH_SB;
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].x, fb[1].x, c1, 0, 0, 0); CH_SB;
    lst4(lds, wslot + 8 * WK, Bq[1]); CH_SB;
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].y, fb[0].y, c0, 0, 0, 0); CH_SB;
    read_b(fbn); CH_SB;
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].y, fb[1].y, c1, 0, 0, 0); CH_SB;
    read_a(fan, A, kt_next); CH_SB;
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].z, fb[0].z, c0, 0, 0, 0); CH_SB;
    i32x4 rs;
    rs.x = (int)__builtin_amdgcn_readfirstlane((unsigned)ent);
    rs.y = (int)__builtin_amdgcn_readfirstlane((unsigned)(ent >> 32));
    rs.z = WN;
    rs.w = 0x00020000;
    {
      const f32x4 v0 = hn_sbuffer_load_x4(rs, vrow, pos16, 0, 0);
      Bq[0] = make_float4(v0.x, v0.y, v0.z, v0.w);
    }
    CH_SB;
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].z, fb[1].z, c1, 0, 0, 0); CH_SB;
    {
      const f32x4 v1 = hn_sbuffer_load_x4(rs, vrow + 8, pos16, 0, 0);
      Bq[1] = make_float4(v1.x, v1.y, v1.z, v1.w);
    }
    fetch_entry(); CH_SB;
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].w, fb[0].w, c0, 0, 0, 0); CH_SB;
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].w, fb[1].w, c1, 0, 0, 0); CH_SB;
#undef CH_SB
  };
  // four blocks; the A fragments the last step requests come from tile `ktn` of `An` (the operand of whatever runs next)
  auto run4 = [&](int A, int kc, int An, int ktn, f32x4 &c0, f32x4 &c1) {
    step(B0, fa0, fb0, fa1, fb1, A, kc + 1, c0, c1);
    step(B1, fa1, fb1, fa0, fb0, A, kc + 2, c0, c1);
    step(B2, fa0, fb0, fa1, fb1, A, kc + 3, c0, c1);
    step(B3, fa1, fb1, fa0, fb0, An, ktn, c0, c1);
  };
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int ncol = wave * 16 + fi;           // accumulator element r: row 4 fg + r, column ncol of the 128-column chunk
  const int stg = B_ts + wave * 256;         // this wave's 16 x 16 store-staging tile (B_ts is free outside the LayerNorm steps)
  // 16 rows x 16 columns of a wave's chunk through LDS to ONE 16-byte store per lane
  auto store_tile = [&](gf32 *C, long ldc, int col0, const float (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[stg + (4 * fg + r) * 16 + fi] = v[r];
    const int srow = lane >> 2, c4 = lane & 3;
    gst4_nt(C + (long)(m0 + srow) * ldc + col0 + wave * 16 + 4 * c4, lld4(lds, stg + srow * 16 + 4 * c4));
  };

  // cluster exchange `which` (0: P, 1: FF) of this launch: the member's partial 16 x 128 tile (accumulator layout v) goes to its
  // slot, the members' partials are summed in member order into B_ts.  Relaxed agent-scope atomics at the shared XCD's L2, the
  // stores acknowledged before the flag (chain.hip explains both).
  auto exchange_into_ts = [&](const float (&v)[4], int which) {
    float *slot = args.xchg + (((long)which * ntiles + tile) * a_C) * (CR * CD);
    int *flags = args.xflags + ((long)which * ntiles + tile) * a_C;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      __hip_atomic_store(slot + (long)member * (CR * CD) + (4 * fg + r) * CD + ncol, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && !(late_kernarg<int>(offsetof(BChainArgs, inject_loss)) && member == a_C - 1))      // (fault injection: the last member's flag never goes up)
      __hip_atomic_store(flags + member, args.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int timed_out = 0;                         // bounded wait, reported (cluster_wait, chain_common.h): no hang ...
    if (tid < a_C)
      timed_out = cluster_wait(flags + tid, args.seq, late_kernarg<unsigned>(offsetof(BChainArgs, wait_ticks)), args.xflags + 2 * ntiles * a_C,
                               late_kernarg<unsigned *>(offsetof(BChainArgs, status)), late_kernarg<unsigned>(offsetof(BChainArgs, token)));
    // ... and no silently incomplete sum either: a tile that gave up on a member becomes NaN (chain.hip explains)
    const bool lost = __syncthreads_or(timed_out) != 0;
    float4 acc = lost ? make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < a_C; ++c) {
      const float *pp = slot + (long)c * (CR * CD) + row * CD + 4 * l32;
      acc.x += __hip_atomic_load(pp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc.y += __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc.z += __hip_atomic_load(pp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc.w += __hip_atomic_load(pp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lst4(lds, B_ts + row * XP + 4 * l32, acc);
  };

  // LayerNorm statistics of the tile in B_xs: normalised rows -> B_xn, rstd -> B_rs; optionally the affine image in the A layout
  // (and to HBM for the weight-gradient product)
  auto ln_stats = [&](bool normalise, bool emit_hat, int gamma, int beta, bool has_beta) {
    float4 v = lld4(lds, B_xs + row * XP + 4 * l32);
    if (normalise) {
      const float mu = half_wave_sum((v.x + v.y) + (v.z + v.w)) * inv_dv;      // (pad columns of x are zero)
      v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
      if (EXT && a_dv < CD) {                // ... and stay out of the variance (and of the normalised image)
        const int c = 4 * l32;
        v.x = c < a_dv ? v.x : 0.0f; v.y = c + 1 < a_dv ? v.y : 0.0f; v.z = c + 2 < a_dv ? v.z : 0.0f; v.w = c + 3 < a_dv ? v.w : 0.0f;
      }
      const float q = half_wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
      const float rs = 1.0f / sqrtf(q * inv_dv + 1e-5f);
      v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
      if (l32 == 0) lds[B_rs + row] = rs;
      lst4(lds, B_xn + row * XP + 4 * l32, v);
      if (emit_hat) {
        const float4 g0 = lld4(lds, gamma + 4 * l32);
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_beta) b0 = lld4(lds, beta + 4 * l32);
        v.x = v.x * g0.x + b0.x; v.y = v.y * g0.y + b0.y; v.z = v.z * g0.z + b0.z; v.w = v.w * g0.w + b0.w;
      }
    }
    if (emit_hat) {
      lst4(lds, a_at(B_Ahat, row, l32), v);
      if (member == 0) gst4_nt(a_Xhat + grow, v);
    }
  };
  // G <- LayerNorm'(dxh in B_ts; B_xn, B_rs, gamma) + G ; per-workgroup column sums of dxh * xn (dgamma) and dxh (dbeta) -> lnpart
  auto ln_bwd = [&](bool normalise, int gamma, int slot) {
    const float4 dv = lld4(lds, B_ts + row * XP + 4 * l32);
    float4 G = lld4(lds, B_gs + row * XP + 4 * l32);
    if (normalise) {
      const float4 xn = lld4(lds, B_xn + row * XP + 4 * l32), g0 = lld4(lds, gamma + 4 * l32);
      const float rs = lds[B_rs + row];
      const float4 gg = make_float4(dv.x * g0.x, dv.y * g0.y, dv.z * g0.z, dv.w * g0.w);
      const float m1 = half_wave_sum((gg.x + gg.y) + (gg.z + gg.w)) * inv_dv;           // (gamma, xn are zero in the pad columns)
      const float m2 = half_wave_sum((gg.x * xn.x + gg.y * xn.y) + (gg.z * xn.z + gg.w * xn.w)) * inv_dv;
      float4 dxv = make_float4(rs * (gg.x - m1 - xn.x * m2), rs * (gg.y - m1 - xn.y * m2), rs * (gg.z - m1 - xn.z * m2),
                               rs * (gg.w - m1 - xn.w * m2));
      if (EXT && a_dv < CD) {                // no gradient into the pad columns
        const int c = 4 * l32;
        dxv.x = c < a_dv ? dxv.x : 0.0f; dxv.y = c + 1 < a_dv ? dxv.y : 0.0f; dxv.z = c + 2 < a_dv ? dxv.z : 0.0f; dxv.w = c + 3 < a_dv ? dxv.w : 0.0f;
      }
      G.x += dxv.x; G.y += dxv.y; G.z += dxv.z; G.w += dxv.w;
      lst4(lds, B_gs + row * XP + 4 * l32, G);
      // column sums over the 16 rows: contributions to two scratch tiles (B_ts holds dxh already; B_xs <- dxh * xn)
      lst4(lds, B_xs + row * XP + 4 * l32, make_float4(dv.x * xn.x, dv.y * xn.y, dv.z * xn.z, dv.w * xn.w));
      __syncthreads();
      if (tid < 256 && member == 0) {
        const int c = tid & 127, src = tid < 128 ? B_xs : B_ts;
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < CR; ++r) acc += !EXT || m0 + r < a_rows ? lds[src + r * XP + c] : 0.0f;      // (tail rows of a ragged last tile: nothing)
        gst1(a_lnpart + ((long)tile * 4 + slot + (tid >> 7)) * CD + c, acc);
      }
    } else {
      G.x += dv.x; G.y += dv.y; G.z += dv.z; G.w += dv.w;
      lst4(lds, B_gs + row * XP + 4 * l32, G);
    }
    __syncthreads();
  };

  // ================= stage P: dxh = dQ W_q + dKV W_kv over one or two segments of the contraction =================
  if (a_has_p) {
    ln_stats(a_p_nw != nullptr, false, 0, 0, false);          // x_p -> B_xn / B_rs (no barrier needed before the GEMM: other regions)
    f32x4 c0 = zero, c1 = zero;
    read_a(fa0, B_Abig, 0);
    for (int kc = 0; kc < nk1; kc += 4) run4(B_Abig, kc, B_Abig, kc + 4 == nk1 ? 0 : kc + 4, c0, c1);
    if (nk2) {
      __syncthreads();                       // every wave is done with the first segment
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (128 * i < pseg2) lst4(lds, a_at(B_Abig, row, 32 * i + l32), seg2[i]);
      __syncthreads();
      read_a(fa0, B_Abig, 0);
      for (int kc = 0; kc < nk2; kc += 4) run4(B_Abig, kc, B_Abig, kc + 4 == nk2 ? 0 : kc + 4, c0, c1);
    }
    const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
    if (p_split) {
      exchange_into_ts(v, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[B_ts + (4 * fg + r) * XP + ncol] = v[r];
    }
    __syncthreads();
    ln_bwd(a_p_nw != nullptr, B_gp, 0);
    if (a_has_ff) lst4(lds, B_xs + row * XP + 4 * l32, xm_pre);
    __syncthreads();
  }

  // ================= stage FF =================
  float4 lmask = make_float4(1.f, 1.f, 1.f, 1.f);            // LeakyReLU'(x_m - x_a) of this thread's chunk (stage OUT)
  if (a_has_ff) {
    if (a_has_out) {
      const float4 xm = lld4(lds, B_xs + row * XP + 4 * l32);
      lmask.x = xm.x - xa_pre.x > 0.0f ? 1.0f : 0.01f; lmask.y = xm.y - xa_pre.y > 0.0f ? 1.0f : 0.01f;
      lmask.z = xm.z - xa_pre.z > 0.0f ? 1.0f : 0.01f; lmask.w = xm.w - xa_pre.w > 0.0f ? 1.0f : 0.01f;
    }
    ln_stats(a_f_nw != nullptr, true, B_gf, B_gf + 128, a_f_nb != nullptr);
    {
      float4 G = lld4(lds, B_gs + row * XP + 4 * l32);
      if (d_thr != 0) {                      // the forward's dropout on the block output: the block proper sees G * keep / (1 - p)
        uint32_t w[4];
        philox4x32(args.ff_drop.seed_lo, args.ff_drop.seed_hi, (uint32_t)l32, (uint32_t)(m0 + row), args.ff_drop.sid, drop_counter(args.ff_drop), w);
        const float d_scale = args.ff_drop.scale;
        G.x = w[0] >= d_thr ? G.x * d_scale : 0.0f; G.y = w[1] >= d_thr ? G.y * d_scale : 0.0f;
        G.z = w[2] >= d_thr ? G.z * d_scale : 0.0f; G.w = w[3] >= d_thr ? G.w * d_scale : 0.0f;
      }
      lst4(lds, a_at(B_Adz, row, l32), G);
      if (member == 0) gst4_nt(a_dYff + grow, G);             // the gradient that entered the block (dW2 = G^T h, db2)
    }
    __syncthreads();
    read_a(fa0, B_Ahat, 0);
    int du_at[4];                             // dU-tile element (row 4 fg + r, column ncol of chunk 0) in the A layout
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 4 * fg + r;
      du_at[r] = B_Abig + (ncol >> 5) * ATILE + rr * WK + ((((ncol & 31) >> 2) ^ (rr & 7)) * 4) + (ncol & 3);
    }
    for (int hci = 0; hci < my_chunks; ++hci) {
      const int hc = member + hci * a_C;
      f32x4 a0 = zero, a1 = zero, g0 = zero, g1 = zero, z0 = zero, z1 = zero;
      run4(B_Ahat, 0, B_Ahat, 0, a0, a1);
      run4(B_Ahat, 0, B_Adz, 0, g0, g1);
      run4(B_Adz, 0, B_Ahat, 0, z0, z1);     // (after the last chunk the fragments are re-read below, behind the barrier)
      const int h0 = hc * WN + ncol;
      const float ba = lds[B_b1 + h0], bg = lds[B_b1 + CHID + h0];
      const float va[4] = {a0.x + a1.x + ba, a0.y + a1.y + ba, a0.z + a1.z + ba, a0.w + a1.w + ba};
      const float vg[4] = {g0.x + g1.x + bg, g0.y + g1.y + bg, g0.z + g1.z + bg, g0.w + g1.w + bg};
      const float vz[4] = {z0.x + z1.x, z0.y + z1.y, z0.z + z1.z, z0.w + z1.w};
      float act[4], dact[4];
      if (a_gate == HN_GATE_SELU) {
        asm volatile("" ::: "memory");
        const float alpha = 1.6732632423543772848170429916717f, scale = 1.0507009873554804934193349852946f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ex = __expf(fminf(vg[r], 0.0f));
          act[r] = scale * (vg[r] > 0.0f ? vg[r] : alpha * (ex - 1.0f));
          dact[r] = scale * (vg[r] > 0.0f ? 1.0f : alpha * ex);
        }
      } else {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float cdf = 0.5f * (1.0f + erff(vg[r] * 0.70710678118654752440f));
          act[r] = vg[r] * cdf;
          dact[r] = cdf + vg[r] * 0.3989422804014327f * __expf(-0.5f * vg[r] * vg[r]);
        }
      }
      float hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hv[r] = va[r] * act[r];
        lds[du_at[r] + hc * 4 * ATILE] = vz[r] * act[r];                        // da: k = h0
        lds[du_at[r] + (16 + hc * 4) * ATILE] = vz[r] * va[r] * dact[r];        // dg: k = 512 + h0
      }
      store_tile(a_H, CHID, hc * WN, hv);
    }
    __syncthreads();
    // the dU tile to HBM for dW1 = dU^T xhat / db1 (16 rows x 4 KB, 16-byte pieces: 8 per thread)
    if (a_C == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = l32 + 32 * i;
        gst4_nt(a_dU + (long)(m0 + row) * (2 * CHID) + 4 * q, lld4(lds, a_at(B_Abig, row, q)));
      }
      f32x4 c0 = zero, c1 = zero;
      read_a(fa0, B_Abig, 0);
#pragma unroll
      for (int kc = 0; kc < 32; kc += 4) run4(B_Abig, kc, B_Abig, kc + 4 == 32 ? 0 : kc + 4, c0, c1);
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[B_ts + (4 * fg + r) * XP + ncol] = v[r];
    } else {
      // cluster: the own chunks' dU columns to HBM, the own k-tiles of dx_hat = dU W1 (partial), exchange 2
      for (int hci = 0; hci < my_chunks; ++hci) {
        const int hc = member + hci * a_C;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int q = half * 128 + hc * 32 + l32;        // 16-byte chunk of the row: da columns, then dg columns of the chunk
          gst4_nt(a_dU + (long)(m0 + row) * (2 * CHID) + 4 * q, lld4(lds, a_at(B_Abig, row, q)));
        }
      }
      f32x4 c0 = zero, c1 = zero;
      read_a(fa0, B_Abig, 4 * member);
      for (int hci = 0; hci < my_chunks; ++hci) {
        const int hc = member + hci * a_C;
        run4(B_Abig, 4 * hc, B_Abig, 16 + 4 * hc, c0, c1);
        run4(B_Abig, 16 + 4 * hc, B_Abig, hci + 1 < my_chunks ? 4 * (hc + a_C) : 0, c0, c1);
      }
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
      exchange_into_ts(v, 1);
    }
    __syncthreads();
    ln_bwd(a_f_nw != nullptr, B_gf, 2);
  }

  // ---- G is the gradient w.r.t. the input of the feed-forward block (= the output of the attention block in front of it)
  {
    const float4 G = lld4(lds, B_gs + row * XP + 4 * l32);
    if (member == 0) gst4_nt(a_dx_out + grow, G);
    if (a_has_out) {
      const float4 dp = make_float4(G.x * lmask.x, G.y * lmask.y, G.z * lmask.z, G.w * lmask.w);
      if (member == 0) gst4_nt(a_dPre + grow, dp);
      lst4(lds, a_at(B_Adz, row, l32), dp);
    }
  }

  // ================= stage OUT: dO = dpre W_out =================
  if (a_has_out) {
    __syncthreads();
    read_a(fa0, B_Adz, 0);
    const int a_o_cols = EXT && args.o_cols > 0 ? args.o_cols : a_inner_o;
    for (int j = member; j < a_inner_o / WN; j += a_C) {
      f32x4 c0 = zero, c1 = zero;
      run4(B_Adz, 0, B_Adz, 0, c0, c1);
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
      if (!EXT || j * WN + wave * 16 < a_o_cols) store_tile(a_dO, a_lddo, j * WN, v);      // (staged models keep the leading columns only)
    }
  }
}

// (any row count: the callers allocate every (rows, .) operand for the count rounded up to 16, api_internal.h rows16)
bool latent_bchain_supported(int rows, int d, int hidden) { return d == CD && hidden == CHID && rows > 0; }

int launch_latent_bchain(const BChainArgs &a, hipStream_t s) {
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  HN_REQUIRE(a.rows > 0, HN_E_SHAPE, "latent_bchain: rows=%d", a.rows);
  HN_REQUIRE(a.dv >= 0 && a.dv <= CD && a.q_cols >= 0 && a.q_cols <= a.nq && a.q_cols % 4 == 0 && a.kv_cols >= 0 && a.kv_cols <= a.nkv &&
                 a.kv_cols % 4 == 0 && a.o_cols >= 0 && a.o_cols <= a.inner_o && a.o_cols % 16 == 0,
             HN_E_SHAPE, "latent_bchain: valid widths dv=%d q_cols=%d kv_cols=%d o_cols=%d", a.dv, a.q_cols, a.kv_cols, a.o_cols);
  HN_REQUIRE(a.ff_drop.thr == 0 || a.has_ff, HN_E_SHAPE, "latent_bchain: dropout without a feed-forward stage");
  HN_REQUIRE(a.dy && a.dx_out && al16(a.dy) && al16(a.dx_out), HN_E_NULL, "latent_bchain: dy / dx_out NULL or unaligned");
  HN_REQUIRE(a.has_p || a.has_ff, HN_E_SHAPE, "latent_bchain: empty chain");
  HN_REQUIRE(!a.has_out || a.has_ff, HN_E_UNSUPPORTED, "latent_bchain: the out-projection stage follows a feed-forward stage");
  if (a.has_p) {
    HN_REQUIRE(a.dQ && a.wqT && a.p_x && a.nq > 0 && a.nq % WN == 0 && a.nkv >= 0 && a.nkv % WN == 0 && a.nq + a.nkv <= 1536 &&
                   a.lddq % 4 == 0 && al16(a.dQ) && al16(a.wqT) && al16(a.p_x) && al16(a.p_nw),
               HN_E_SHAPE, "latent_bchain: P stage nq=%d nkv=%d", a.nq, a.nkv);
    HN_REQUIRE(a.nkv == 0 || (a.dKV && a.wkvT && a.lddkv % 4 == 0 && al16(a.dKV) && al16(a.wkvT)), HN_E_SHAPE, "latent_bchain: dKV operand");
    // the second segment only holds dKV columns
    HN_REQUIRE(a.nq + a.nkv <= 1024 || a.nq <= 1024, HN_E_SHAPE, "latent_bchain: nq=%d", a.nq);
  }
  if (a.has_ff) {
    HN_REQUIRE(a.f_x && a.w1 && a.b1 && a.w2T && a.w1T && a.H && a.dU && a.Xhat && a.dYff, HN_E_NULL, "latent_bchain: feed-forward operand is NULL");
    HN_REQUIRE(al16(a.f_x) && al16(a.w1) && al16(a.b1) && al16(a.w2T) && al16(a.w1T) && al16(a.H) && al16(a.dU) && al16(a.Xhat) && al16(a.dYff) &&
                   al16(a.f_nw) && al16(a.f_nb), HN_E_SHAPE, "latent_bchain: unaligned feed-forward operand");
    HN_REQUIRE((a.f_nw == nullptr) == (a.f_nb == nullptr) || a.f_nw != nullptr, HN_E_SHAPE, "latent_bchain: LayerNorm bias without weight");
  }
  if (a.has_out) {
    HN_REQUIRE(a.o_x && a.woT && a.dPre && a.dO && a.inner_o > 0 && a.inner_o % WN == 0 && a.inner_o <= 512 && a.lddo % 4 == 0 && al16(a.o_x) &&
                   al16(a.woT) && al16(a.dPre) && al16(a.dO), HN_E_SHAPE, "latent_bchain: OUT stage inner=%d", a.inner_o);
  }
  HN_REQUIRE(a.lnpart && al16(a.lnpart), HN_E_NULL, "latent_bchain: lnpart is NULL");
  static bool configured[64] = {};
  const int lds_bytes = B_LDS_FLOATS * (int)sizeof(float);
  int dev = 0;
  HN_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    HN_HIP_CHECK(hipFuncSetAttribute((const void *)latent_bchain_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    HN_HIP_CHECK(hipFuncSetAttribute((const void *)latent_bchain_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  const bool no_cluster = !cluster_enabled(dev);
  BChainArgs ac = a;
  const int tiles = (a.rows + CR - 1) / CR;
  const int gtiles = (tiles + 7) / 8 * 8;    // member rows of a cluster grid: a multiple of 8 tiles (members of a tile share an XCD)
  ac.cluster = 1;
  ac.tiles = tiles;
  if (!no_cluster && a.xchg && a.xflags && a.seq > 0 && a.has_ff && gtiles <= 128 && al16(a.xchg)) {
    const int C = gtiles <= 64 ? 4 : 2;
    // every member of every tile must be resident at once (the exchanges spin): ONE 135 KB workgroup fits a CU
    static int cu_count[64] = {};
    if (dev >= 0 && dev < 64 && cu_count[dev] == 0) {
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 1;
      cu_count[dev] = n;
    }
    const int cus = (dev >= 0 && dev < 64) ? cu_count[dev] : 1;
    if (gtiles * C <= cus) ac.cluster = C;
  }
  if (ac.cluster > 1) {
    ClusterTicket t;
    cluster_before_launch(dev, s, &t);
    ac.status = t.status; ac.token = t.token; ac.wait_ticks = t.wait_ticks; ac.inject_loss = t.inject_loss;
    static const bool split_order = tuning_env("HN_FORCE_CLUSTER_SPLIT_ORDER") != nullptr;      // route switch (A/B): the former grid order
    ac.split_order = split_order ? 1 : 0;
  }
  const bool ext = a.rows % CR != 0 || (a.dv > 0 && a.dv < CD) || a.ff_drop.thr != 0 || (a.has_p && a.q_cols > 0 && a.q_cols < a.nq) ||
                   (a.has_p && a.kv_cols > 0 && a.kv_cols < a.nkv) || (a.has_out && a.o_cols > 0 && a.o_cols < a.inner_o) ||
                   (ac.cluster > 1 && gtiles != tiles);
  const dim3 grid(ac.cluster > 1 ? gtiles * ac.cluster : tiles);
  if (ext) hipLaunchKernelGGL(latent_bchain_kernel<true>, grid, dim3(512), lds_bytes, s, ac);
  else hipLaunchKernelGGL(latent_bchain_kernel<false>, grid, dim3(512), lds_bytes, s, ac);
  HN_LAUNCH_CHECK("latent_bchain");
  if (ac.cluster > 1) cluster_after_launch(dev, s);
  return HN_OK;
}

}  // namespace hn
