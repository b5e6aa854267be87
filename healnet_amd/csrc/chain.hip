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
// What streams is the WEIGHTS.  Eight waves split each 128-column chunk of a GEMM (16 columns = 16 weight rows per wave), and
// every wave moves ITS 16 rows x 32 k of each block: two 16-byte buffer loads per lane in FULL 128-byte lines (8 lanes per row)
// into a four-deep register ring that runs across chunk and stage boundaries (one flat block order for the whole chain), then
// two ds_write_b128 into the wave's private 2 KB LDS slot (XOR-swizzled), from where the MFMA B fragments are read back with
// ds_read_b128.  A wave only ever reads what it wrote itself, so there is NO barrier inside a GEMM; waves meet only where a
// stage hands its tile to the next.
//   History (tools/bench_chain.py and tools/ubench/l2_fill.hip; cfg2 b = 32, 12 chains per forward):
//   v1  weights staged global -> VGPR -> one SHARED 3-buffer LDS ring, barrier per block, 4 or 8 waves: 52.6 us per chain.  A
//       block took 2.1x its MFMA time; removing the ds_writes alone saved 36 % of it, removing the global loads almost nothing.
//   v2  B fragments straight from global memory into a register ring in the MFMA operand layout (16 rows x 64 B per load, the
//       attention core's pattern), no LDS, no barrier: 49 us.  With the loads removed a block costs 0.216 us (= 8 MFMAs per wave
//       on two waves per SIMD: the matrix work is AT its bound), with them 0.46 us: 38 GB/s per CU.
//   v3  per-wave LDS rings filled by global_load_lds_dwordx4 (full lines, counted vmcnt from inline asm): 49 us again.
//   The micro-benchmark then separated pattern from path -- all 256 CUs streaming the same L2-resident 2 MB: 117-124 GB/s per
//   CU with full-line loads into registers, 81 GB/s through global_load_lds, and exactly the chain's 38 GB/s with the 16 x 64 B
//   fragment pattern (K = 128 or 512 alike; neither an L2 warm-up pass nor rotating the chunk order had changed anything).
//   v4  this version: full-line register loads + a private LDS transpose per wave: 47 us -- and the loads were still not the
//       limit: with every address replaced by a constant the forward dropped by 0.19 ms.  What each wave paid per block was the
//       SCALAR work of finding it (stage -> weight pointer / leading dimension / chunk / k offset: ~45 SALU instructions and a
//       handful of v_readfirstlane on the critical path of the issue).  The block order of a chain is known at kernel start, so
//       512 threads now write it ONCE as a table in LDS (64-bit byte address of the block's first row, ldw/128 - 1 in the two
//       low bits) and a wave fetches the entry of the NEXT block with one broadcast ds_read_b64 a step ahead: 40 us per chain,
//       forward 3.224 -> 3.036 ms at cfg2 b = 32.  (An L2 warm-up pass was re-tried on top of this: no change.)
//   v5  stage cycle stamps (tools/chain_profile.py, -DCHAIN_PROFILE) then showed 740-925 cycles per block against the 512 of its
//       MFMAs, and the ISA why: in the unrolled feed-forward stages the scheduler had sunk every global load to just in front of
//       its ds_write (s_waitcnt vmcnt(0) per block), and everywhere it grouped the ds_writes, the address arithmetic and the
//       loads in front of the first MFMA of a step.  The step is now pinned slot by slot (one memory instruction behind each
//       MFMA): 700-870 cycles per block, forward 2.99 ms.  Streaming (nt) stores for x / Q / KV -- the end-of-kernel write-back
//       of 24 MB dirty lines cost 4-8 us per chain -- 2.975 ms; a split expm1 in the SELU epilogue and DPP row sums in LayerNorm
//       (two ds_bpermute chains of 5) another ~1 us per chain.
//       Ablations on v5 (cycles per block, OUT / FF2 / KV): as is 774 / 722 / 728; no global loads 700 / 654 / 664; loads kept
//       but consumed straight from registers (no LDS transpose: what weights pre-tiled in MFMA order would buy) 830 / 710 / 790
//       -- NO gain, so the LDS path is not the cost; neither loads nor B-side LDS 582 / 546 / 595; a ring twice as deep 830 /
//       838 / 762 (worse: not latency).  What is left is what two waves per SIMD cannot overlap of ~30 non-MFMA instructions
//       per block; the micro-benchmark (mfma_stream.hip) sits at the same 690 cycles.
//   v6  ... and the micro-benchmark then named the instructions: the fp32 MFMA shares the SIMD's issue with the VALU, and the
//       64-bit per-lane addresses of global_load cost five vector instructions per block (v_mul_lo_u32, v_or, v_ashr, two
//       v_lshl_add_u64).  With buffer loads -- base and row pitch of the block in the SGPR descriptor (the table entry IS words
//       0-1 of it), row index and piece offset in loop-invariant VGPRs -- the same stream costs 0.241 us per block instead of
//       0.307 (MFMAs alone: 0.225): 590-730 cycles per block in the chain, 30.4 us per launch, forward 2.92 ms, and the chain now
//       beats the per-block launches at every batch size (profiles/r02_g_chain_sweep.txt).
//   head 3  behind a shared-context (image) block the chain also merges the core's split partials and applies the folded value
//       projection (vfold_kernel stages it once per forward): merge_vproj_kernel's launch (17 us) disappears, the chain's
//       prologue grows by one round trip: forward 2.773 -> 2.742 ms.
//   ... and one more vector-instruction item: `gate == SELU ? selu(g) : gelu(g)` in the FF1 epilogue was a select, so BOTH gates
//       (erff among them) were evaluated for every hidden element; behind a scalar branch, with the lane's four hidden-tile
//       addresses precomputed: 2.747 -> 2.724 ms.
//
// Shapes: l_d = 128, hidden 512, K a multiple of 128, N a multiple of 128, rows % 16 == 0.
#include "common.h"
#include <stdlib.h>
#include <mutex>
#include <string.h>
#include <stddef.h>

namespace hn {

namespace {

#include "chain_common.h"
constexpr int PRM = 128 + 2 * CHID + 128 + 4 * 128;      // b_out | b1 | b2 | ff gamma, beta | projection gamma, beta
constexpr int MAXBLK = 128;             // blocks of a chain: <= 16 (out) + 32 + 16 (ff) + 16 (q) + 32 (kv) = 112
constexpr int LDS_FLOATS = 8 * WSLOT + 16 * ATILE + 4 * ATILE + CR * XP + PRM + 2 * MAXBLK;       // 72.25 KB
enum { CS_OUT = 0, CS_FF1 = 1, CS_FF2 = 2, CS_Q = 3, CS_KV = 4, CS_END = 5 };

#ifdef CHAIN_PROFILE
// development only (tools/chain_profile.py builds a private library with -DCHAIN_PROFILE): cycle stamps of one workgroup at the
// stage boundaries of the last 16 launches
__device__ unsigned long long g_chain_prof[16 * 16];
__device__ int g_chain_seq;
#define CHAIN_PROF(i)                                                                                          \
  do {                                                                                                         \
    if (blockIdx.x == 100 % gridDim.x && threadIdx.x == 0)                                                     \
      g_chain_prof[(g_chain_seq & 15) * 16 + (i)] = (i) >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); \
  } while (0)
#else
#define CHAIN_PROF(i)
#endif

}  // namespace

// EXT = false: the shapes of the default model exactly (rows % 16 == 0, l_d = 128, no dropout, whole operands); EXT = true adds
// the valid widths of staged models, ragged row counts, idle workgroups of rounded-up cluster grids and the feed-forward dropout
// -- a second instantiation so that none of it costs the first a register (13 spilled SGPRs when it was one kernel).
template <bool EXT>
__global__ __launch_bounds__(512) void latent_chain_kernel(const ChainArgs args) {
  // Every field of the by-value argument struct is unpacked ONCE into a local (pointers as global-address-space pointers): the
  // lambdas below capture locals only.  Capturing the struct itself keeps a copy of it in scratch, and the compiler then
  // turns "select among pointers" into loads from a selected scratch address followed by flat accesses.
  const gf32 *const a_x_in = (const gf32 *)args.x_in;
  const gf32 *const a_O = (const gf32 *)args.O;
  const gf32 *const a_b_out = (const gf32 *)args.b_out;
  const gf32 *const a_y = (const gf32 *)args.y;
  const gf32 *const a_Opart = (const gf32 *)args.Opart;
  const gf32 *const a_Mpart = (const gf32 *)args.Mpart;
  const gf32 *const a_Lpart = (const gf32 *)args.Lpart;
  const gf32 *const a_wvf = (const gf32 *)args.wvf;
  gf32 *const a_stats = (gf32 *)args.stats;
  const int a_nsplit = args.nsplit, a_Lp = args.Lp, a_heads = args.heads, a_dh = args.dh;
  const gf32 *const a_f_nw = (const gf32 *)args.f_nw;
  const gf32 *const a_f_nb = (const gf32 *)args.f_nb;
  const gf32 *const a_b1 = (const gf32 *)args.b1;
  const gf32 *const a_b2 = (const gf32 *)args.b2;
  const gf32 *const a_p_nw = (const gf32 *)args.p_nw;
  const gf32 *const a_p_nb = (const gf32 *)args.p_nb;
  gf32 *const a_x_out = (gf32 *)args.x_out;
  gf32 *const a_x_mid = (gf32 *)args.x_mid;
  gf32 *const a_Q = (gf32 *)args.Q;
  gf32 *const a_KV = (gf32 *)args.KV;
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
  // staged (padded) models: valid widths (common.h)
  const int a_rows = args.rows;
  const int a_dv = EXT && args.dv > 0 ? args.dv : CD;
  const float inv_dv = 1.0f / (float)a_dv;
  const uint32_t d_thr = EXT ? args.ff_drop.thr : 0u;   // (the rest of the generator state is read from the argument segment where it is used)
  CHAIN_PROF(14);
  CHAIN_PROF(0);
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  lf32 *lds = (lf32 *)lds_raw;              // float offsets into the one LDS allocation:
  constexpr int Wr = 0;                     // [8 waves][WSLOT]  per-wave transpose slot of the weight stream
  constexpr int Abig = Wr + 8 * WSLOT;      // [16][ATILE]  attention output tile (A of the out-projection), then the FF hidden tile,
                                            //              then (projections) the per-wave output staging tiles
  constexpr int Ahat = Abig + 16 * ATILE;   // [4][ATILE]   LayerNorm-ed x (A of FF1 / of the projections)
  constexpr int xs = Ahat + 4 * ATILE;      // [CR][XP]     the x tile
  constexpr int prm = xs + CR * XP;         // small parameters (no global load sits between the steps: the weight ring owns the memory queue)
  constexpr int tbl = prm + PRM;            // [MAXBLK] byte address of every block of the chain (64 bit; bits 0-1: row length / 128 - 1)
  constexpr int p_bout = prm, p_b1 = p_bout + 128, p_b2 = p_b1 + 2 * CHID, p_fnw = p_b2 + 128, p_fnb = p_fnw + 128,
                p_pnw = p_fnb + 128, p_pnb = p_pnw + 128;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fg = lane >> 4, fi = lane & 15;
  // Small batches (rows / 16 <= 64 workgroups would leave most of the chip idle while each pulls the whole weight stream through
  // one CU): a CLUSTER of C = 2 or 4 workgroups shares a row tile.  Every member runs the head stage (it needs the whole x tile for
  // the LayerNorm anyway), then only ITS hidden chunks of the feed-forward block (FF1 columns and the matching k-tiles of FF2: a
  // partial x tile), the members exchange the partials ONCE through global memory (fixed summation order: all members hold the
  // same bits afterwards), and each computes its share of the projection chunks.  Members of a tile sit `ntiles` workgroups
  // 8 workgroups apart inside a group of 8 C (cluster_decode, chain_common.h): they share an XCD (its L2) and are dispatched
  // back to back.
  const int a_C = args.cluster > 1 ? args.cluster : 1;
  const int ntiles = gridDim.x / a_C;
  int member_, tile_;
  cluster_decode((int)blockIdx.x, a_C, ntiles, args.split_order, member_, tile_);
  const int member = __builtin_amdgcn_readfirstlane(member_), tile = __builtin_amdgcn_readfirstlane(tile_);
  if (EXT && tile >= args.tiles) return;            // cluster grids are rounded up to 8 tiles per member row (XCD alignment): idle workgroups
  const int my_chunks = 4 / a_C;             // hidden chunks (128 columns) of this member: member, member + C, ...
  const int m0 = tile * CR;

  // ---- the block stream: stage -> n-chunk -> k-chunk, identical for the loader and the consumer.  The loader addresses
  // block number `lb` of the whole chain with branch-free scalar arithmetic.
  const int nk_out = (a_head == 1 || a_head == 3) ? a_inner_o / WK : 0;      // out-projection: ONE 128-column chunk of nk_out k-chunks
  const int nq_ch = a_nq / WN, nkv_ch = a_nkv / WN;
  const int my_proj = (nq_ch + nkv_ch - member + a_C - 1) / a_C;            // projection chunks pj = member, member + C, ...
  const int e0 = nk_out;                                         // first block of FF1
  const int e1 = e0 + (a_has_ff ? 2 * my_chunks * (CD / WK) : 0);           //                FF2
  const int e2 = e1 + (a_has_ff ? my_chunks * (CD / WK) : 0);    //                Q / KV
  const int nblocks = max(1, e2 + my_proj * (CD / WK));
  // ---- the address of every block, once: thread bi works out (stage, chunk, k-chunk) of block bi and leaves the byte address of
  // its first row in an LDS table.  Doing this arithmetic in the loader -- ~45 scalar instructions and four readfirstlanes per
  // block and wave -- was what kept the weight stream from overlapping with the MFMAs (0.42 us per block against 0.24 without
  // loads; 0.25 with a constant-stride dummy address): a table entry costs one broadcast ds_read_b64, fetched a step ahead.
  if (tid < MAXBLK) {
    const int bi = min(tid, nblocks - 1);
    const int s_out = bi < e0, s_ff1 = bi >= e0 && bi < e1, s_ff2 = bi >= e1 && bi < e2;
    const int local = bi - (s_out ? 0 : s_ff1 ? e0 : s_ff2 ? e1 : e2);
    // FF1: per own hidden chunk hc its value rows, then its gate rows (4 k-blocks each); FF2: the k-tiles of the own chunks;
    // projections: chunks pj = member + ci * C of the [Q | KV] column range
    const int hc1 = member + (local >> 3) * a_C, hc2 = member + (local >> 2) * a_C, pj = member + (local >> 2) * a_C;
    const int s_q = !s_out && !s_ff1 && !s_ff2 && pj < nq_ch;
    const int k = s_out ? local : s_ff2 ? 4 * hc2 + (local & 3) : local & 3;
    const int ldw = s_out ? a_inner_o : s_ff2 ? CHID : CD;
    const int rb = s_out || s_ff2 ? 0 : s_ff1 ? ((local >> 2) & 1) * CHID + hc1 * WN : (s_q ? pj : pj - nq_ch) * WN;
    const float *W = s_out ? args.w_out : s_ff1 ? args.w1 : s_ff2 ? args.w2 : s_q ? args.wq : args.wkv;
    // words 0 and 1 of the block's buffer descriptor: base = its first row at k, stride = the row pitch in bytes
    const unsigned long long addr = (unsigned long long)(W + (long)rb * ldw + k * WK);
    const unsigned long long desc = (addr & 0x0000ffffffffffffull) | ((unsigned long long)(ldw * 4) << 48);
    *(__attribute__((address_space(3))) unsigned long long *)(lds + tbl + 2 * tid) = desc;
  }
  const int r8 = lane >> 3, pos = lane & 7;                      // loader lane: row r8 (and r8 + 8) of the wave's 16, 16-byte piece pos
  const int wslot = Wr + wave * WSLOT + r8 * WK + ((pos ^ (r8 & 7)) * 4);   // ... parked at slot pos ^ (row & 7) of its LDS row
  const int vrow = wave * 16 + r8, pos16 = pos * 16;             // the lane's row of the 128-row block (index), byte offset of its piece
  unsigned long long ent = 0;               // table entry of the next block to request (fetched a step ahead)
  // the table pointer lives in a VGPR (opaque to the compiler: as a scalar it is copied into one with a v_mov per fetch) and
  // advances by one entry per fetch: immediate offsets inside the unrolled groups of four steps, one v_add per group
  int tp = tbl;
  asm volatile("" : "+v"(tp));
  auto fetch_entry = [&]() {                // at most nblocks + 4 < MAXBLK fetches
    ent = *(const __attribute__((address_space(3))) unsigned long long *)(lds + tp);
    tp += 2;
  };

  // this wave's 2 KB of block lb (past the end: the last block again, never consumed): rows 16 wave + r8 and + 8 of the chunk
  // as FULL 128-byte lines (8 lanes per row) -- 16 rows x 64 B per load, the MFMA fragment layout, runs at a third of the rate
  // (tools/ubench/l2_fill.hip: 38 against 117 GB/s per CU when all CUs stream the same L2-resident weights)
  // The loads are BUFFER loads with the block's base and row pitch in the SGPR descriptor and loop-invariant VGPR operands (row
  // index, piece offset): no vector instruction computes an address.  The fp32 MFMA shares the SIMD's issue with the VALU --
  // every VALU instruction in the step is matrix time lost -- and the 64-bit per-lane addresses of global_load (v_mul_lo,
  // two v_lshl_add_u64, ...) were most of what kept a block at 1.45x its MFMA time (tools/ubench/mfma_stream.hip: 8 MFMAs
  // + 2 loads per wave and block 0.307 us with global loads, 0.241 us with buffer loads, 0.225 us without loads).
  auto load2 = [&](float4 (&r)[2]) {
    i32x4 rs;
    rs.x = (int)__builtin_amdgcn_readfirstlane((unsigned)ent);
    rs.y = (int)__builtin_amdgcn_readfirstlane((unsigned)(ent >> 32));
    rs.z = WN;                              // 128 records (rows)
    rs.w = 0x00020000;
    const f32x4 v0 = hn_sbuffer_load_x4(rs, vrow, pos16, 0, 0), v1 = hn_sbuffer_load_x4(rs, vrow + 8, pos16, 0, 0);
    r[0] = make_float4(v0.x, v0.y, v0.z, v0.w);
    r[1] = make_float4(v1.x, v1.y, v1.z, v1.w);
  };
  auto issue = [&](float4 (&r)[2]) {
    load2(r);
    fetch_entry();
  };
  // ... and into the wave's LDS slot, from where the MFMA fragments are read back (a wave only reads what it wrote: no barrier)
  auto park = [&](const float4 (&r)[2]) {
    lst4(lds, wslot, r[0]);
    lst4(lds, wslot + 8 * WK, r[1]);
  };

  // ---- head 3: the attention-output tile is built HERE from the split partials of the shared-context core (what
  // merge_vproj_kernel did in a launch of its own: 17 us between the core and this chain).  Wave w = head w; lane (g, i):
  // row i of the tile, 16-byte piece g of the 16-wide merged row (dp = 16: the image binding).  Merge over the splits (weights 2^(M_s - M), sum
  // l = sum w_s l_s), normalise, then the folded value projection on the matrix cores: the lane's four floats ARE the A
  // It runs BEFORE the weight ring starts: its 72 landing registers and the ring's 40 are then never live together (the kernel
  // stays within 128 VGPRs, two workgroups per CU at b >= 64), at the price of one serial round trip.  The lane's four floats: the A
  // operands of the four k-steps (k = 4 g + s: the contraction order is free), B = wvf rows (column dp-1 of the merged row is
  // exactly 1 -- the ones column -- and carries the beta term).  The 16 x dh result goes into the A tile of the out-projection.
  if (a_head == 3 && wave < a_heads) {
    const int i = lane & 15, gq = lane >> 4;
    const int bi = m0 / a_L, q = m0 - bi * a_L + i;                       // L % 16 == 0: a tile never straddles two samples
    const long prow = ((long)(bi * a_heads + wave) * a_nsplit) * a_Lp + q;      // + s * Lp
    // groups of CHAIN_MERGE_GROUP splits (all requests of a group in flight together), folded into a running (M, l, o) in split
    // order: one group for the usual 8-12 splits, up to four when a small batch needs more splits to fill the wave slots
    float M = -3.0e38f, l = 0.0f;
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f);
    // the lane's rows of the folded value projection (dh / 16 <= 4 column tiles), requested WITH the partials: loading each in front
    // of its four MFMAs was a dependent round trip per tile -- 13.7 k cycles in this prologue against ~2 k without head 3
    // (tools/chain_profile.py, round 5)
    float4 bw[4];
    {
      const gf32 *wbase = a_wvf + ((long)(wave * a_dh + i) * 16 + 4 * gq);      // tile ct: + ct * 256 floats
      const int nct = a_dh >> 4;
      bw[0] = gld4(wbase);
      bw[1] = gld4(wbase + (1 < nct ? 256 : 0));
      bw[2] = gld4(wbase + (2 < nct ? 512 : 0));
      bw[3] = gld4(wbase + (3 < nct ? 768 : 0));
    }
    for (int s0 = 0; s0 < a_nsplit; s0 += CHAIN_MERGE_GROUP) {
      float mv[CHAIN_MERGE_GROUP], lv[CHAIN_MERGE_GROUP];
      float4 ov[CHAIN_MERGE_GROUP];
#pragma unroll
      for (int s = 0; s < CHAIN_MERGE_GROUP; ++s) {                       // clamped: weight 0 past nsplit
        const long pr = prow + (long)min(s0 + s, a_nsplit - 1) * a_Lp;
        mv[s] = gld1(a_Mpart + pr);
        lv[s] = gld1(a_Lpart + pr);
        ov[s] = gld4(a_Opart + pr * 16 + 4 * gq);
      }
      // (the scheduler otherwise sinks some of the partial-O requests to just in front of their use to save registers: three
      // load / s_waitcnt vmcnt(0) / fma round trips one behind the other in this prologue -- the ISA, end of round 5)
      __builtin_amdgcn_sched_barrier(0);
      float Mn = M;
#pragma unroll
      for (int s = 0; s < CHAIN_MERGE_GROUP; ++s) Mn = fmaxf(Mn, s0 + s < a_nsplit ? mv[s] : -3.0e38f);
      const float sc = __builtin_amdgcn_exp2f(M - Mn);                    // first group: 2^(-3e38 - Mn) = 0 on an all-zero state
      l *= sc;
      o0.x *= sc; o0.y *= sc; o0.z *= sc; o0.w *= sc;
#pragma unroll
      for (int s = 0; s < CHAIN_MERGE_GROUP; ++s) {
        const float w = s0 + s < a_nsplit ? __builtin_amdgcn_exp2f(mv[s] - Mn) : 0.0f;
        l = fmaf(w, lv[s], l);
        o0.x = fmaf(w, ov[s].x, o0.x); o0.y = fmaf(w, ov[s].y, o0.y); o0.z = fmaf(w, ov[s].z, o0.z); o0.w = fmaf(w, ov[s].w, o0.w);
      }
      M = Mn;
    }
    const float inv = 1.0f / l;
    o0.x *= inv; o0.y *= inv; o0.z *= inv; o0.w *= inv;
    if (a_stats && gq == 0 && member == 0) {
      gf32 *st = a_stats + ((long)(bi * a_heads + wave) * a_L + q) * 2;
      gst1(st, M);
      gst1(st + 1, l);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {                                      // 16 output columns of the head at a time
      if (ct >= (a_dh >> 4)) break;
      const float4 b0 = bw[ct];
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(o0.x, b0.x, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(o0.y, b0.y, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(o0.z, b0.z, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(o0.w, b0.w, c, 0, 0, 0);
      const int col = wave * a_dh + 16 * ct + i;                          // accumulator register r: row 4 gq + r, column col
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * gq + r;
        lds[Abig + (col >> 5) * ATILE + row * WK + ((((col & 31) >> 2) ^ (row & 7)) * 4) + (col & 3)] = c[r];
      }
    }
  }
  // ---- far loads first: five weight blocks per wave (block 0, then the 4-deep register ring: blocks 1..4), then the small
  // parameters, the x tile, the attention-output tile
  __syncthreads();                           // the block table
  CHAIN_PROF(1);
  fetch_entry();
  float4 Bp[2], B0[2], B1[2], B2[2], B3[2];
  issue(Bp);
  issue(B0);
  issue(B1);
  issue(B2);
  issue(B3);
  if (tid < PRM / 4) {                      // 448 threads, one 16-byte piece each
    const int q = tid;
    const gf32 *src = nullptr;
    if (q < 32) src = a_b_out ? a_b_out + 4 * q : nullptr;
    else if (q < 288) src = a_b1 ? a_b1 + 4 * (q - 32) : nullptr;
    else if (q < 320) src = a_b2 ? a_b2 + 4 * (q - 288) : nullptr;
    else if (q < 352) src = a_f_nw ? a_f_nw + 4 * (q - 320) : nullptr;
    else if (q < 384) src = a_f_nb ? a_f_nb + 4 * (q - 352) : nullptr;
    else if (q < 416) src = a_p_nw ? a_p_nw + 4 * (q - 384) : nullptr;
    else src = a_p_nb ? a_p_nb + 4 * (q - 416) : nullptr;
    if (src) lst4(lds, prm + 4 * q, gld4(src));
  }
  {
    const int row = tid >> 5, l32 = tid & 31;           // 32 lanes per row, one 16-byte chunk each
    float4 v0 = gld4(a_x_in + (long)(m0 + row) * CD + 4 * l32);
    if (a_head == 2) {                      // one-token cross block: the same output row for every latent row of a sample
      const float4 y0 = gld4(a_y + (long)((EXT ? min(m0 + row, a_rows - 1) : m0 + row) / a_L) * CD + 4 * l32);
      v0.x += y0.x; v0.y += y0.y; v0.z += y0.z; v0.w += y0.w;
    }
    lst4(lds, xs + row * XP + 4 * l32, v0);
    if (a_head == 1) {
      const gf32 *orow = a_O + (long)(m0 + row) * a_ldo;
      const int a_o_cols = EXT && args.o_cols > 0 ? args.o_cols : a_inner_o;
      // 16-byte chunk q of the row: k-tile q >> 3, slot q & 7.  A lane owns at most FOUR chunks (inner_o <= 16 WK = 512 floats,
      // launch_latent_chain), and all of them are requested before the first is parked: as a loop with a run-time trip count this
      // was one dependent round trip per chunk (global_load, s_waitcnt vmcnt(0), ds_write, four times over) -- the prologue of a
      // chain behind an out-projection took 7.6 k cycles against 3.5-4 k for one without the tile (tools/chain_profile.py, round 5)
      const int nq4 = a_inner_o >> 2;
      float4 o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = l32 + 32 * u;
        o[u] = make_float4(0.f, 0.f, 0.f, 0.f);             // (staged models: the contraction beyond O's own columns is zero)
        if (q < nq4 && 4 * q < a_o_cols) o[u] = gld4(orow + 4 * q);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = l32 + 32 * u;
        if (q < nq4) lst4(lds, Abig + (q >> 3) * ATILE + row * WK + (((q & 7) ^ (row & 7)) * 4), o[u]);
      }
    }
  }
  park(Bp);                                 // block 0
  __syncthreads();
  CHAIN_PROF(2);

  // ---- consumer state: fragments of the even / odd block of a pair (slots s2 = 0 / 1): A from the shared tiles, B from the
  // wave's slot (row fi of its 16, same XOR swizzle on both sides)
  float4 fa0[2], fa1[2], fb0[2], fb1[2];
  auto read_a = [&](float4 (&f)[2], int A, int kt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) f[s2] = lld4(lds, A + kt * ATILE + fi * WK + (((4 * s2 + fg) ^ (fi & 7)) * 4));
  };
  auto read_b = [&](float4 (&f)[2]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) f[s2] = lld4(lds, Wr + wave * WSLOT + fi * WK + (((4 * s2 + fg) ^ (fi & 7)) * 4));
  };
  read_b(fb0);                              // block 0

  // One block.  `Bq` holds block t+1 (requested four steps ago): park it in the wave's LDS slot (the fragments of block t
  // left it a step ago), re-use the registers for the request of block t+5, read the fragments of block t+1 back, and issue
  // the 8 MFMAs of block t.  Straight-line code, no barrier; the compiler counts vmcnt (6 younger loads stay in flight).
  auto step = [&](float4 (&Bq)[2], const float4 (&fa)[2], const float4 (&fb)[2], float4 (&fan)[2], float4 (&fbn)[2], int A, int kt_next,
                  f32x4 &c0, f32x4 &c1) {
    // The order below is pinned slot by slot (sched_barrier(0): nothing crosses).  A wave issues in order and there are only two
    // waves per SIMD, so every memory instruction sits BEHIND an MFMA of the same wave that keeps the matrix pipe busy (32 cycles)
    // while the memory instruction waits for its queue; left to itself the scheduler puts the two ds_writes, the address
    // arithmetic and the two global loads in front of the first MFMA of a step (~200 idle cycles per block), and in the unrolled
    // feed-forward stages it even sank every global load to just in front of its ds_write (s_waitcnt vmcnt(0) per block).
    // Two accumulators (k-slots 0 and 1) so that consecutive MFMAs never depend on each other; they are summed in the epilogue.
#define CH_SB __builtin_amdgcn_sched_barrier(0)
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].x, fb[0].x, c0, 0, 0, 0); CH_SB;
    lst4(lds, wslot, Bq[0]); CH_SB;                                        // park block t+1, first half
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].x, fb[1].x, c1, 0, 0, 0); CH_SB;
    lst4(lds, wslot + 8 * WK, Bq[1]); CH_SB;                               //                 second half
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].y, fb[0].y, c0, 0, 0, 0); CH_SB;
    read_b(fbn); CH_SB;                                                    // its fragments back
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1].y, fb[1].y, c1, 0, 0, 0); CH_SB;
    read_a(fan, A, kt_next); CH_SB;
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0].z, fb[0].z, c0, 0, 0, 0); CH_SB;
    i32x4 rs;
    rs.x = (int)__builtin_amdgcn_readfirstlane((unsigned)ent);
    rs.y = (int)__builtin_amdgcn_readfirstlane((unsigned)(ent >> 32));
    rs.z = WN;
    rs.w = 0x00020000;
    {
      const f32x4 v0 = hn_sbuffer_load_x4(rs, vrow, pos16, 0, 0);            // request block t+5 into the registers just parked
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
  // the k loop of one 128-column chunk, four blocks per iteration (nk is a multiple of 4): the register slots and the fragment
  // sets alternate, nothing is copied.  The A fragments requested by the last step (k-tile 0 again) serve the next chunk of
  // the same stage.
  auto run4 = [&](int A, int kc, int nk, f32x4 &c0, f32x4 &c1) {
    step(B0, fa0, fb0, fa1, fb1, A, kc + 1, c0, c1);
    step(B1, fa1, fb1, fa0, fb0, A, kc + 2, c0, c1);
    step(B2, fa0, fb0, fa1, fb1, A, kc + 3, c0, c1);
    step(B3, fa1, fb1, fa0, fb0, A, kc + 4 == nk ? 0 : kc + 4, c0, c1);
  };
  auto run4n = [&](int A, int kc, int ktn, f32x4 &c0, f32x4 &c1) {      // ... with the k-tile the last step prefetches given
    step(B0, fa0, fb0, fa1, fb1, A, kc + 1, c0, c1);
    step(B1, fa1, fb1, fa0, fb0, A, kc + 2, c0, c1);
    step(B2, fa0, fb0, fa1, fb1, A, kc + 3, c0, c1);
    step(B3, fa1, fb1, fa0, fb0, A, ktn, c0, c1);
  };
  // K = 128 and K = 512 with the k-tile a compile-time constant: the A-fragment addresses are immediates (with a run-time k-tile
  // the compiler adds the scalar tile offset to the lane's VGPR offset with a v_add per fragment read -- VALU time the MFMAs lose)
  auto run_chunk_k128 = [&](int A, f32x4 &c0, f32x4 &c1) { run4(A, 0, 4, c0, c1); };
  auto run_chunk_k512 = [&](int A, f32x4 &c0, f32x4 &c1) {
#pragma unroll
    for (int kc = 0; kc < 16; kc += 4) run4(A, kc, 16, c0, c1);
  };
  auto run_chunk = [&](int A, int nk, f32x4 &c0, f32x4 &c1) {       // run-time K (the out-projection: inner_o / 32 k-tiles)
    for (int kc = 0; kc < nk; kc += 4) run4(A, kc, nk, c0, c1);
  };
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // LayerNorm of the x tile -> Ahat (A layout); no affine: the rows as they are
  auto layer_norm = [&](bool affine, int gamma, int beta, gf32 *keep = nullptr) {
    const int row = tid >> 5, l32 = tid & 31;
    float4 v = lld4(lds, xs + row * XP + 4 * l32);
    if (affine) {
      const float mu = half_wave_sum((v.x + v.y) + (v.z + v.w)) * inv_dv;      // (pad columns of x are zero)
      v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
      if (EXT && a_dv < CD) {                // ... and stay out of the variance
        const int c = 4 * l32;
        v.x = c < a_dv ? v.x : 0.0f; v.y = c + 1 < a_dv ? v.y : 0.0f; v.z = c + 2 < a_dv ? v.z : 0.0f; v.w = c + 3 < a_dv ? v.w : 0.0f;
      }
      const float q = half_wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
      const float rs = 1.0f / sqrtf(q * inv_dv + 1e-5f);
      const float4 g0 = lld4(lds, gamma + 4 * l32), b0 = lld4(lds, beta + 4 * l32);
      v.x = v.x * rs * g0.x + b0.x; v.y = v.y * rs * g0.y + b0.y; v.z = v.z * rs * g0.z + b0.z; v.w = v.w * rs * g0.w + b0.w;
    }
    lst4(lds, Ahat + (l32 >> 3) * ATILE + row * WK + (((l32 & 7) ^ (row & 7)) * 4), v);   // k = 4 l32: k-tile l32 >> 3, slot l32 & 7
    if (keep) gst4_nt(keep + (long)(m0 + row) * CD + 4 * l32, v);
  };
  // accumulator element r: row 4 fg + r, column 16 wave + fi of the 128-column chunk
  const int ncol = wave * 16 + fi;

  // ================= stage OUT: x += LeakyReLU(O W_out^T + b_out) =================
  if (nk_out) {
    f32x4 c0 = zero, c1 = zero;
    read_a(fa0, Abig, 0);
    run_chunk(Abig, nk_out, c0, c1);
    const float bv = lds[p_bout + ncol];
    const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = v[r] + bv;
      p = p > 0.0f ? p : 0.01f * p;
      lds[xs + (4 * fg + r) * XP + ncol] += p;
    }
    __syncthreads();
  }
  CHAIN_PROF(3);

  // ---- training: the feed-forward block's input goes to the tape (the backward recomputes the block from it)
  if (a_x_mid && member == 0) {
    const int row = tid >> 5, l32 = tid & 31;
    gst4_nt(a_x_mid + (long)(m0 + row) * CD + 4 * l32, lld4(lds, xs + row * XP + 4 * l32));
  }

  // ================= stages FF1 / FF2: x += (a * gate(g)) W2^T + b2,  [a | g] = LN(x) W1^T + b1 =================
  if (a_has_ff) {
    layer_norm(a_f_nw != nullptr, p_fnw, p_fnb);
    __syncthreads();
    CHAIN_PROF(4);
    read_a(fa0, Ahat, 0);
    int hid_at[4];                          // hidden-tile element (row 4 fg + r, column ncol of chunk 0): A layout, 16-byte slots XOR-swizzled
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * fg + r;
      hid_at[r] = Abig + (ncol >> 5) * ATILE + row * WK + ((((ncol & 31) >> 2) ^ (row & 7)) * 4) + (ncol & 3);
    }
    for (int hci = 0; hci < my_chunks; ++hci) {
      const int hc = member + hci * a_C;
      f32x4 a0 = zero, a1 = zero, g0 = zero, g1 = zero;
      run_chunk_k128(Ahat, a0, a1);
      run_chunk_k128(Ahat, g0, g1);
      const int h0 = hc * WN + ncol;
      const float ba = lds[p_b1 + h0], bg = lds[p_b1 + CHID + h0];
      const float va[4] = {a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w};
      const float vg[4] = {g0.x + g1.x, g0.y + g1.y, g0.z + g1.z, g0.w + g1.w};
      // one gate, behind a scalar branch (a select between the two evaluates both on the VALU the MFMAs share), and the four
      // hidden-tile addresses of the lane precomputed: only the chunk offset is added here
      float gate[4];
      if (a_gate == HN_GATE_SELU) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) gate[r] = selu_f(vg[r] + bg);
      } else {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) gate[r] = gelu_erf(vg[r] + bg);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[hid_at[r] + hc * 4 * ATILE] = (va[r] + ba) * gate[r];
    }
    __syncthreads();
    CHAIN_PROF(5);
    if (a_C == 1) {
      f32x4 c0 = zero, c1 = zero;
      read_a(fa0, Abig, 0);
      run_chunk_k512(Abig, c0, c1);
      const float bv = lds[p_b2 + ncol];
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
      if (d_thr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[xs + (4 * fg + r) * XP + ncol] += v[r] + bv;
      } else {
        // dropout on the block output (:347): through the (dead) LN tile into the row layout, where a thread holds one aligned
        // column quad -- one generator call (common.h: the mask is a function of (row, quad) of the (rows, l_d) output)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[Ahat + (4 * fg + r) * CD + ncol] = v[r] + bv;
        __syncthreads();
        const int row = tid >> 5, l32 = tid & 31;
        const float4 f = lld4(lds, Ahat + row * CD + 4 * l32);
        float4 x = lld4(lds, xs + row * XP + 4 * l32);
        uint32_t w[4];
        philox4x32(args.ff_drop.seed_lo, args.ff_drop.seed_hi, (uint32_t)l32, (uint32_t)(m0 + row), args.ff_drop.sid, drop_counter(args.ff_drop), w);
        const float d_scale = args.ff_drop.scale;
        x.x += w[0] >= d_thr ? f.x * d_scale : 0.0f; x.y += w[1] >= d_thr ? f.y * d_scale : 0.0f;
        x.z += w[2] >= d_thr ? f.z * d_scale : 0.0f; x.w += w[3] >= d_thr ? f.w * d_scale : 0.0f;
        lst4(lds, xs + row * XP + 4 * l32, x);
      }
    } else {
      // cluster: the k-tiles of the own hidden chunks only -> a partial tile; exchange with the other members
      f32x4 c0 = zero, c1 = zero;
      read_a(fa0, Abig, 4 * member);
      for (int hci = 0; hci < my_chunks; ++hci) {
        const int hc = member + hci * a_C;
        run4n(Abig, 4 * hc, hci + 1 < my_chunks ? 4 * (hc + a_C) : 0, c0, c1);
      }
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
      // The members of a tile share an XCD (launcher: ntiles % 8 == 0), i.e. one L2: partials and flags move as RELAXED agent-scope
      // atomics, which are performed at the L2 -- no release fence (a device-scope release writes back the XCD's whole L2:
      // measured 12 us per exchange, more than the cluster saved from b = 2 on), no stale L1 lines on the reading side.
      float *mine = args.xchg + ((long)tile * a_C + member) * (CR * CD);
#pragma unroll
      for (int r = 0; r < 4; ++r) __hip_atomic_store(mine + (4 * fg + r) * CD + ncol, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // ... but ORDER still has to be made: a workgroup-scope fence emits no wait at all here (the waves of a workgroup share the
      // CU's L1), and stores to different L2 channels are not ordered among themselves -- the flag could overtake a partial.
      // Every wave waits until ITS stores have been acknowledged by the L2 (vmcnt counts stores on gfx9), then the barrier, then
      // the flag.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0 && !(late_kernarg<int>(offsetof(ChainArgs, inject_loss)) && member == a_C - 1))      // ... before the member's flag goes up (fault injection: the last member's never does)
        __hip_atomic_store(args.xflags + tile * a_C + member, args.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // one lane per member flag; bounded wait (cluster_wait, chain_common.h): the dispatch order of a cluster grid keeps the
      // members of a tile together, but a member that still never shows up must not hang the device.  A tile that gives up is NOT
      // allowed to carry on with incomplete sums silently: the whole tile becomes NaN, which reaches the logits / gradients of its
      // sample, and the launch reports itself in the device's status word (HN_E_CORESIDENCY).
      int timed_out = 0;
      if (tid < a_C)
        timed_out = cluster_wait(args.xflags + tile * a_C + tid, args.seq, late_kernarg<unsigned>(offsetof(ChainArgs, wait_ticks)), args.xflags + ntiles * a_C,
                                 late_kernarg<unsigned *>(offsetof(ChainArgs, status)), late_kernarg<unsigned>(offsetof(ChainArgs, token)));
      const bool lost = __syncthreads_or(timed_out) != 0;
      {
        const int row = tid >> 5, l32 = tid & 31;
        float4 x = lld4(lds, xs + row * XP + 4 * l32);
        // fixed order: every member ends with the same bits.  Without dropout the partials are added onto x + b2 one by one;
        // with it they are summed first (f = b2 + sum of the partials is what the mask thins)
        const float4 bb = lld4(lds, p_b2 + 4 * l32);
        float4 f = d_thr != 0 ? bb : make_float4(x.x + bb.x, x.y + bb.y, x.z + bb.z, x.w + bb.w);
        for (int c = 0; c < a_C; ++c) {
          const float *pp = args.xchg + ((long)tile * a_C + c) * (CR * CD) + row * CD + 4 * l32;
          f.x += __hip_atomic_load(pp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          f.y += __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          f.z += __hip_atomic_load(pp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          f.w += __hip_atomic_load(pp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (d_thr != 0) {                    // dropout on the block output: every member draws the same mask
          uint32_t w[4];
          philox4x32(args.ff_drop.seed_lo, args.ff_drop.seed_hi, (uint32_t)l32, (uint32_t)(m0 + row), args.ff_drop.sid, drop_counter(args.ff_drop), w);
          const float d_scale = args.ff_drop.scale;
          x.x += w[0] >= d_thr ? f.x * d_scale : 0.0f; x.y += w[1] >= d_thr ? f.y * d_scale : 0.0f;
          x.z += w[2] >= d_thr ? f.z * d_scale : 0.0f; x.w += w[3] >= d_thr ? f.w * d_scale : 0.0f;
        } else {
          x = f;
        }
        if (lost) x = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
        lst4(lds, xs + row * XP + 4 * l32, x);
      }
    }
    __syncthreads();
  }

  CHAIN_PROF(6);
  // ---- x is final: hand it to the next attention block (its input / residual, and the trace slot of hn_attn_probs)
  if (a_x_out && member == 0) {
    const int row = tid >> 5, l32 = tid & 31;
    gst4_nt(a_x_out + (long)(m0 + row) * CD + 4 * l32, lld4(lds, xs + row * XP + 4 * l32));
  }

  // ================= stages Q / KV: the next attention block's projections of LN'(x) =================
  if (my_proj > 0) {
    layer_norm(a_p_nw != nullptr, p_pnw, p_pnb, (args.xhat_out && member == 0) ? (gf32 *)args.xhat_out : nullptr);
    __syncthreads();
    CHAIN_PROF(7);
    read_a(fa0, Ahat, 0);
    const int stg = Abig + wave * 256;       // this wave's 16 x 16 output tile (the hidden tile is dead by now)
    const int a_q_cols = EXT && args.q_cols > 0 ? args.q_cols : a_nq, a_kv_cols = EXT && args.kv_cols > 0 ? args.kv_cols : a_nkv;
    for (int pj = member; pj < nq_ch + nkv_ch; pj += a_C) {
      const bool isq = pj < nq_ch;
      const int j = isq ? pj : pj - nq_ch;
      f32x4 c0 = zero, c1 = zero;
      run_chunk_k128(Ahat, c0, c1);
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
      if (isq && args.qf != nullptr) {
        // folded query of a rank-D block: this wave's 16 columns are head `wave`'s packed slots; rows of 64 contiguous bytes in the
        // head-major image, and the row's score bound from the squares summed over the 16 lanes that share it
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[stg + (4 * fg + r) * 16 + fi] = v[r];
        const int srow = lane >> 2, c4 = lane & 3;
        const int grow = m0 + srow, bi = grow / args.L, q = grow - bi * args.L;
        gst4_nt((gf32 *)args.qf + (((long)bi * args.qf_heads + wave) * args.L + q) * 16 + 4 * c4, lld4(lds, stg + srow * 16 + 4 * c4));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float ss = v[r] * v[r];
          ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 8);
          if (fi == 0) {
            const int gr = m0 + 4 * fg + r, b2 = gr / args.L, q2 = gr - b2 * args.L;
            const float bq = sqrtf(ss * (float)args.qf_D) * 1.00002f + 1e-6f;
            gst1((gf32 *)args.qf_bound + ((long)b2 * args.qf_heads + wave) * args.L + q2, bq);
            if (bq > 60.0f) atomicOr(args.qf_flag, 1);
          }
        }
        continue;
      }
      gf32 *C = isq ? a_Q : a_KV;
      const long ldc = isq ? a_ldq : a_ldkv;
      const float al = isq ? a_alpha_q : 1.0f;
      // through LDS to ONE 16-byte store per lane (16 rows x 64 contiguous bytes per wave) instead of four 4-byte ones: a store
      // counts against vmcnt like a load, so every outstanding store shortens the weight ring's lead
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[stg + (4 * fg + r) * 16 + fi] = al * v[r];
      const int srow = lane >> 2, c4 = lane & 3;
      if (!EXT || j * WN + wave * 16 < (isq ? a_q_cols : a_kv_cols)) // (staged models keep the leading columns only)
        gst4_nt(C + (long)(m0 + srow) * ldc + j * WN + wave * 16 + 4 * c4, lld4(lds, stg + srow * 16 + 4 * c4));
      if (pj + 1 == nq_ch) CHAIN_PROF(8);
    }
  }
  CHAIN_PROF(9);
  CHAIN_PROF(15);
#ifdef CHAIN_PROFILE
  if (blockIdx.x == 100 % gridDim.x && threadIdx.x == 0) {
    g_chain_prof[(g_chain_seq & 15) * 16 + 10] = (unsigned long long)nblocks | ((unsigned long long)nk_out << 16) | ((unsigned long long)nq_ch << 32) | ((unsigned long long)nkv_ch << 48);
    __threadfence();
    g_chain_seq = g_chain_seq + 1;
  }
#endif
}

#ifdef CHAIN_PROFILE
extern "C" __attribute__((visibility("default"))) int hn_debug_chain_prof(unsigned long long *out, int *seq) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_prof), sizeof(unsigned long long) * 256) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(seq, HIP_SYMBOL(g_chain_seq), sizeof(int)) != hipSuccess) return -1;
  return 0;
}
#endif

#include "vfold.h"

__global__ __launch_bounds__(256) void vfold_kernel(VfoldMulti v) {
  vfold_body(v, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.z);
}

// grid of the vfold roles (with the broadcast rows decided): (gx, gy, gz) and the argument block the kernel takes
int vfold_plan(const VfoldMulti &v, VfoldMulti *vv, int *gx, int *gy, int *gz) {
  HN_REQUIRE(v.n >= 1 && v.n <= 16 && v.out && v.D >= 1 && v.D <= 15 && v.heads >= 1, HN_E_SHAPE, "vfold: n=%d D=%d heads=%d", v.n, v.D, v.heads);
  HN_REQUIRE(v.qout == nullptr || (v.dh <= 128 && v.l_d >= 1), HN_E_SHAPE, "vfold: query fold dh=%d l_d=%d", v.dh, v.l_d);
  *vv = v;
  vv->bc_rows = 0;
  *gz = v.qout ? 1 + (v.l_d + 31) / 32 : 1;
  if (v.bc_dst) {
    HN_REQUIRE(v.bc_src && v.bc_per > 0 && v.bc_per % 4 == 0 && v.bc_total % v.bc_per == 0 && (((uintptr_t)v.bc_src | (uintptr_t)v.bc_dst) & 15) == 0, HN_E_SHAPE,
               "vfold: broadcast role per=%ld total=%ld", v.bc_per, v.bc_total);
    const long want = (v.bc_total / 4 + 256 * 4 - 1) / (256 * 4);      // ~4 passes per thread
    long rows = (want + (long)v.heads * *gz - 1) / ((long)v.heads * *gz);
    vv->bc_rows = (int)(rows < 1 ? 1 : rows > 512 ? 512 : rows);
  }
  *gx = v.heads;
  *gy = v.n + vv->bc_rows;
  return HN_OK;
}

int launch_vfold(const VfoldMulti &v, hipStream_t s) {
  VfoldMulti vv;
  int gx, gy, gz;
  const int rc = vfold_plan(v, &vv, &gx, &gy, &gz);
  if (rc != HN_OK) return rc;
  hipLaunchKernelGGL(vfold_kernel, dim3(gx, gy, gz), dim3(256), 0, s, vv);
  HN_LAUNCH_CHECK("vfold");
  return HN_OK;
}

bool latent_chain_supported(int rows, int d, int hidden) { return d == CD && hidden == CHID && rows > 0 && rows % CR == 0; }

int launch_latent_chain(const ChainArgs &a, hipStream_t s) {
  HN_REQUIRE(a.rows > 0 && a.L > 0, HN_E_SHAPE, "latent_chain: rows=%d L=%d", a.rows, a.L);
  HN_REQUIRE(a.dv >= 0 && a.dv <= CD && a.o_cols >= 0 && a.o_cols <= a.inner_o && a.o_cols % 4 == 0 && a.q_cols >= 0 && a.q_cols <= a.nq &&
                 a.q_cols % 16 == 0 && a.kv_cols >= 0 && a.kv_cols <= a.nkv && a.kv_cols % 16 == 0,
             HN_E_SHAPE, "latent_chain: valid widths dv=%d o_cols=%d q_cols=%d kv_cols=%d", a.dv, a.o_cols, a.q_cols, a.kv_cols);
  HN_REQUIRE(a.ff_drop.thr == 0 || a.has_ff, HN_E_SHAPE, "latent_chain: dropout without a feed-forward block");
  HN_REQUIRE(a.x_in, HN_E_NULL, "latent_chain: x_in is NULL");
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  if (a.head == 1) {
    HN_REQUIRE(a.O && a.w_out && a.b_out, HN_E_NULL, "latent_chain: out-projection operand is NULL");
    HN_REQUIRE(a.inner_o > 0 && a.inner_o % (4 * WK) == 0 && a.inner_o <= 16 * WK && a.ldo % 4 == 0 && al16(a.O) && al16(a.w_out), HN_E_SHAPE,
               "latent_chain: inner=%d ldo=%d", a.inner_o, a.ldo);
  } else if (a.head == 2) {
    HN_REQUIRE(a.y && al16(a.y), HN_E_NULL, "latent_chain: y is NULL / unaligned");
  } else if (a.head == 3) {
    HN_REQUIRE(a.Opart && a.Mpart && a.Lpart && a.wvf && a.w_out && a.b_out, HN_E_NULL, "latent_chain: merge operand is NULL");
    HN_REQUIRE(a.dp == 16 && a.heads >= 1 && a.heads <= 8 && (a.dh == 16 || a.dh == 32 || a.dh == 64) &&
                   a.inner_o == a.heads * a.dh && a.inner_o % (4 * WK) == 0 && a.inner_o <= 16 * WK && a.nsplit >= 1 &&
                   a.nsplit <= CHAIN_MERGE_MAX_SPLITS && a.L % CR == 0 && a.Lp >= a.L && al16(a.Opart) && al16(a.wvf) && al16(a.w_out),
               HN_E_SHAPE, "latent_chain: merge head dp=%d heads=%d dh=%d nsplit=%d L=%d", a.dp, a.heads, a.dh, a.nsplit, a.L);
  }
  if (a.has_ff) {
    HN_REQUIRE(a.w1 && a.b1 && a.w2 && a.b2, HN_E_NULL, "latent_chain: feed-forward operand is NULL");
    HN_REQUIRE(al16(a.w1) && al16(a.w2) && (!a.f_nw || (al16(a.f_nw) && al16(a.f_nb))), HN_E_SHAPE, "latent_chain: unaligned operand");
  }
  HN_REQUIRE(a.nq >= 0 && a.nkv >= 0 && a.nq % WN == 0 && a.nkv % WN == 0, HN_E_SHAPE, "latent_chain: nq=%d nkv=%d", a.nq, a.nkv);
  HN_REQUIRE(a.nq == 0 || (a.wq && (a.Q || a.qf) && al16(a.wq)), HN_E_NULL, "latent_chain: Q projection operand is NULL");
  HN_REQUIRE(a.qf == nullptr || (a.nq == WN && a.qf_heads * 16 == WN && a.qf_bound && a.qf_flag && a.qf_D >= 1 && a.L % CR == 0 &&
                                 a.rows % a.L == 0 && al16(a.qf) && (a.q_cols == 0 || a.q_cols == a.nq)),
             HN_E_SHAPE, "latent_chain: folded query stage nq=%d heads=%d L=%d", a.nq, a.qf_heads, a.L);
  HN_REQUIRE(a.nkv == 0 || (a.wkv && a.KV && al16(a.wkv)), HN_E_NULL, "latent_chain: KV projection operand is NULL");
  HN_REQUIRE(al16(a.x_in) && (!a.x_out || al16(a.x_out)) && (!a.x_mid || al16(a.x_mid)) && (!a.p_nw || (al16(a.p_nw) && al16(a.p_nb))), HN_E_SHAPE,
             "latent_chain: unaligned operand");
  // one-time opt-in per device to > 64 KB of dynamic LDS (a function attribute; setting it twice is harmless, so no lock)
  static bool configured[64] = {};
  const int lds_bytes = LDS_FLOATS * (int)sizeof(float);
  int dev = 0;
  HN_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    HN_HIP_CHECK(hipFuncSetAttribute((const void *)latent_chain_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    HN_HIP_CHECK(hipFuncSetAttribute((const void *)latent_chain_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  // cluster mode for small batches (see the kernel): 4 workgroups per row tile up to 64 tiles, 2 up to 128 -- at most 256
  // workgroups, all resident (the exchange spins on the other members' flags)
  const bool no_cluster = !cluster_enabled(dev);
  static const int max_tiles = tuning_env("HN_CHAIN_CLUSTER_TILES") ? atoi(tuning_env("HN_CHAIN_CLUSTER_TILES")) : 128;      // development knob
  ChainArgs ac = a;
  const int tiles = (a.rows + CR - 1) / CR;
  // members of a tile sit `gtiles` workgroups apart and must share an XCD (workgroups are dealt round-robin over the 8 XCDs):
  // the member rows of the grid are rounded up to a multiple of 8 tiles, the workgroups beyond `tiles` return at once
  const int gtiles = (tiles + 7) / 8 * 8;
  ac.cluster = 1;
  ac.tiles = tiles;
  if (!no_cluster && a.xchg && a.xflags && a.seq > 0 && a.has_ff && gtiles <= max_tiles && al16(a.xchg)) {
    // every member of every tile must be resident at once (the exchange spins): two 72 KB workgroups fit a CU
    static int cu_count[64] = {};
    if (dev >= 0 && dev < 64 && cu_count[dev] == 0) {
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 1;
      cu_count[dev] = n;
    }
    const int cus = (dev >= 0 && dev < 64) ? cu_count[dev] : 1, C = gtiles <= 64 ? 4 : 2;
    if (gtiles * C <= 2 * cus && gtiles * C <= CHAIN_XCHG_FLAGS - 1) ac.cluster = C;
  }
  if (ac.cluster > 1) {
    ClusterTicket t;
    cluster_before_launch(dev, s, &t);
    ac.status = t.status; ac.token = t.token; ac.wait_ticks = t.wait_ticks; ac.inject_loss = t.inject_loss;
    static const bool split_order = tuning_env("HN_FORCE_CLUSTER_SPLIT_ORDER") != nullptr;      // route switch (A/B): the former grid order
    ac.split_order = split_order ? 1 : 0;
  }
  const bool ext = a.rows % CR != 0 || (a.dv > 0 && a.dv < CD) || a.ff_drop.thr != 0 || (a.o_cols > 0 && a.o_cols < a.inner_o) ||
                   (a.q_cols > 0 && a.q_cols < a.nq) || (a.kv_cols > 0 && a.kv_cols < a.nkv) || (ac.cluster > 1 && gtiles != tiles);
  const dim3 grid(ac.cluster > 1 ? gtiles * ac.cluster : tiles);
  if (ext) hipLaunchKernelGGL(latent_chain_kernel<true>, grid, dim3(512), lds_bytes, s, ac);
  else hipLaunchKernelGGL(latent_chain_kernel<false>, grid, dim3(512), lds_bytes, s, ac);
  HN_LAUNCH_CHECK("latent_chain");
  if (ac.cluster > 1) cluster_after_launch(dev, s);
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// Host side of the cluster contract (common.h explains the three parts).  One record per device; everything under its mutex:
// submission order under the mutex is a total order of the device's cluster launches, and each is ordered behind its
// predecessor when that went to another stream, so no two cluster kernels of this process run at once.  While only ONE stream
// has ever carried cluster launches nothing is recorded (a lock and a capture query per launch); the first launch from a second
// stream drains the device once and switches to recording an event behind every cluster launch.  Streams under graph capture are
// left alone (an outside event must not enter a capture; a replayed graph is the caller's to order); other processes on the same
// GPU cannot be seen at all -- the dispatch order of the grid and the bounded, reported wait are what covers those.
// ------------------------------------------------------------------------------------------------
namespace {
struct ClusterDev {
  std::mutex mu;
  hipEvent_t ev = nullptr;
  hipStream_t last = nullptr;             // compared only, never passed to HIP again
  bool any = false, multi = false, ev_valid = false;
  unsigned *word_host = nullptr, *word_dev = nullptr;
  bool alloc_failed = false;
  bool disabled = false;
  bool reported = false;                  // the word's current token has been counted by a poll that could not clear it
  bool inject = false;                    // hn_cluster_config(enable = 2): test hook, every cluster launch loses an exchange
  unsigned next_token = 0, lost = 0, last_token = 0;
  int timeout_us = -1;                    // -1: HN_CLUSTER_TIMEOUT_US or the default
};
ClusterDev g_cluster[64];
constexpr int CLUSTER_TIMEOUT_US_DEFAULT = 100000;

bool stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return true; }      // unknown: treat as "leave alone"
  return st != hipStreamCaptureStatusNone;
}
// the status word: 64 bytes of host-coherent pinned memory, mapped into the device (the ONE allocation this library makes; a kernel
// only writes it when a wait has run into its bound).  Not attempted from inside a capture.
void ensure_status_word(ClusterDev &g) {
  if (g.word_host || g.alloc_failed) return;
  void *h = nullptr, *d = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || h == nullptr) {
    (void)hipGetLastError();
    g.alloc_failed = true;
    return;
  }
  memset(h, 0, 64);
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || d == nullptr) {
    (void)hipGetLastError();
    (void)hipHostFree(h);
    g.alloc_failed = true;
    return;
  }
  g.word_host = (unsigned *)h;
  g.word_dev = (unsigned *)d;
}
int timeout_us_of(const ClusterDev &g) {
  if (g.timeout_us > 0) return g.timeout_us;
  static const int env = getenv("HN_CLUSTER_TIMEOUT_US") ? atoi(getenv("HN_CLUSTER_TIMEOUT_US")) : 0;
  return env > 0 ? env : CLUSTER_TIMEOUT_US_DEFAULT;
}
}  // namespace

bool cluster_enabled(int dev) {
  static const bool no_cluster = getenv("HN_NO_CHAIN_CLUSTER") != nullptr;
  if (no_cluster || dev < 0 || dev >= 64) return false;
  ClusterDev &g = g_cluster[dev];
  std::lock_guard<std::mutex> lock(g.mu);
  return !g.disabled;
}

void cluster_before_launch(int dev, hipStream_t s, ClusterTicket *t) {
  t->status = nullptr; t->token = 1; t->wait_ticks = (unsigned)CLUSTER_TIMEOUT_US_DEFAULT * 100u; t->inject_loss = 0;
  if (dev < 0 || dev >= 64) return;
  ClusterDev &g = g_cluster[dev];
  std::lock_guard<std::mutex> lock(g.mu);
  const bool capturing = stream_capturing(s);
  if (!capturing) ensure_status_word(g);
  t->status = g.word_dev;
  if (++g.next_token == 0) g.next_token = 1;     // 0 means "clean"
  t->token = g.next_token;
  t->wait_ticks = (unsigned)timeout_us_of(g) * 100u;      // s_memrealtime: 100 MHz
  t->inject_loss = g.inject ? 1 : 0;
  if (capturing) return;
  if (g.any && g.last != s) {
    if (!g.multi) {
      // first cluster launch from a second stream: nothing was recorded behind the launches so far -- drain the device once
      g.multi = true;
      if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    } else if (g.ev_valid) {
      if (hipStreamWaitEvent(s, g.ev, 0) != hipSuccess) (void)hipGetLastError();
    }
  }
  g.last = s;
  g.any = true;
}

void cluster_after_launch(int dev, hipStream_t s) {
  if (dev < 0 || dev >= 64) return;
  ClusterDev &g = g_cluster[dev];
  std::lock_guard<std::mutex> lock(g.mu);
  if (!g.multi || stream_capturing(s)) return;
  if (g.ev == nullptr && hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); g.ev = nullptr; return; }
  g.ev_valid = hipEventRecord(g.ev, s) == hipSuccess;
  if (!g.ev_valid) (void)hipGetLastError();
}

int cluster_poll(const char *who, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return HN_OK; }
  if (dev < 0 || dev >= 64) return HN_OK;
  ClusterDev &g = g_cluster[dev];
  std::lock_guard<std::mutex> lock(g.mu);
  if (!g.word_host) return HN_OK;
  const unsigned v = *(volatile unsigned *)g.word_host;
  if (v == 0) return HN_OK;
  g.disabled = true;
  if (!g.reported || g.last_token != v) g.lost += 1;
  g.reported = true;
  g.last_token = v;
  // The word is cleared only once the device has DRAINED: an hn_l1_adam_step enqueued before this call (the host running ahead of
  // a stalled chain) decides on the device whether to skip by reading this word, and must still find it set when it executes.
  // Under stream capture nothing can be waited for: the word stays set -- every fused entry point keeps returning
  // HN_E_CORESIDENCY -- until the caller has drained the device itself and acknowledged through hn_cluster_status.
  if (!stream_capturing(s) && hipDeviceSynchronize() == hipSuccess) {
    *(volatile unsigned *)g.word_host = 0;
    g.reported = false;
  } else {
    (void)hipGetLastError();
  }
  return fail(HN_E_CORESIDENCY,
              "%s: a cluster-mode latent chain launched earlier on device %d (launch token %u) gave up waiting for a member "
              "workgroup (co-residency lost: busy / masked CUs); the rows of that tile -- and what was computed from them since "
              "-- are NaN.  Cluster mode is now OFF for this device; repeat the step",
              who, dev, v);
}

const unsigned *cluster_status_device_word(int dev) {
  if (dev < 0 || dev >= 64) return nullptr;
  ClusterDev &g = g_cluster[dev];
  std::lock_guard<std::mutex> lock(g.mu);
  return g.word_dev;
}

int cluster_status(int dev, int acknowledge, hn_cluster_info *info) {
  HN_REQUIRE(dev >= 0 && dev < 64, HN_E_SHAPE, "cluster_status: device %d", dev);
  ClusterDev &g = g_cluster[dev];
  std::lock_guard<std::mutex> lock(g.mu);
  int cur = 0;
  const bool on_dev = hipGetDevice(&cur) == hipSuccess && cur == dev;
  if (on_dev) ensure_status_word(g);             // (an eager caller asking early also makes sure the word exists before a capture)
  unsigned v = g.word_host ? *(volatile unsigned *)g.word_host : 0u;
  if (v != 0 && acknowledge) {
    g.disabled = true;
    if (!g.reported || g.last_token != v) g.lost += 1;      // (a poll that could not drain has counted this loss already)
    g.reported = false;
    g.last_token = v;
    *(volatile unsigned *)g.word_host = 0;
  }
  if (info) {
    info->pending = v != 0;
    info->lost = g.lost;
    info->last_token = v != 0 ? v : g.last_token;
    info->enabled = !g.disabled && getenv("HN_NO_CHAIN_CLUSTER") == nullptr;
    info->timeout_us = timeout_us_of(g);
    info->status_word = g.word_host;
  }
  return HN_OK;
}

int cluster_config(int dev, int enable, int timeout_us) {
  HN_REQUIRE(dev >= 0 && dev < 64, HN_E_SHAPE, "cluster_config: device %d", dev);
  ClusterDev &g = g_cluster[dev];
  std::lock_guard<std::mutex> lock(g.mu);
  if (enable >= 0) { g.disabled = enable == 0; g.inject = enable == 2; }
  if (timeout_us >= 0) g.timeout_us = timeout_us == 0 ? -1 : timeout_us;
  return HN_OK;
}

}  // namespace hn
