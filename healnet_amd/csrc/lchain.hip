// latent_layer_kernel -- the WHOLE latent side between two shared-context (image) attention cores as ONE persistent launch
// (SURVEY.md 7 step 6 / 8(b) hn_latent_block_fwd; healnet/models/healnet.py:236-237, :241-245 -- and the one-token cross block of
// the next layer iteration, :235-237, whose output vector does not depend on the latent array):
//
//   a launch is a list of SEGMENTS, each  [head] -> feed-forward block -> [projections of the next attention block -> that block]:
//     head 2   x += y[sample]                         one-token cross block (its output row was computed ahead of the layer loop)
//     head 3   x += LeakyReLU(O W_out^T + b_out)      O merged here from the split partials of the shared-context core (chain.hip)
//     head 4   x += LeakyReLU(O W_out^T + b_out)      O = the latent self-attention output the PREVIOUS segment left in LDS
//     FF       x += FF(LN(x))
//     proj 1   Q | K | V of the latent self-attention (:241-244), and the attention itself (below)
//     proj 2   the folded, packed query of the next shared-context block (chain.hip "qf") -- last segment only
//
// latent_chain_kernel (chain.hip) is one such segment per launch, with the self-attention core (self_attention.hip) as a launch of
// its own in between.  At cfg2 b = 32 that is 12 chain launches of 26-45 us, of which ~9 us each do not scale with the work
// (dispatch of 256 x 512 threads, block table, x tile, first weight round trip, write-back, drain), and 6 core launches of 16.9 us
// for 6.8 us of MFMA time.  Here the x tile never leaves LDS between two image cores, the weight ring runs
// across segment boundaries, and the self-attention is a STAGE FAMILY of the same ring:
//
//   * a workgroup still owns 16 latent rows (one row tile); the 8 tiles of a sample (l_c = 128) are the MEMBERS of a cluster in the
//     sense of chain_common.h: adjacent in their XCD's dispatch order, one L2.  The KV stage writes the tile's K rows head-major,
//     (b, 8, 128, 64), and V TRANSPOSED, (b, 8, 64, 128) -- straight from the accumulators: lane (g, j) holds four consecutive
//     tokens of column j -- and the tile raises its flag.  The Q stage then runs HEAD BY HEAD (wave w projects the 64 columns of head w:
//     the Q tile of a head is written and read by one wave, no barrier) and hides the flags' way to the siblings; every wave requests
//     the sample's 8 flags a chunk early and checks them itself (bounded, reported wait otherwise: HN_E_CORESIDENCY);
//   * both images are then ordinary "weight matrices" for the per-wave register ring (full 128-byte lines, private LDS transpose
//     slot, descriptor from the block table), with wave w = head w: K as rows = tokens (16 blocks of 16 tokens x 32 dims), V^T as
//     rows = dims (16 blocks of 16 dims x 32 tokens); the loads carry sc1 (agent scope: they never hit in the CU's L1);
//   * S^T = K_h Q_h^T with A = the K fragment from the ring and B = the wave's Q fragments (read ONCE from the LDS tile the Q stage
//     wrote: 16 registers) lands in the A-operand layout of P V (register r of lane (g, j) = token 16 t + 4 g + r of query row j),
//     so the plain two-pass softmax runs in registers (self_attention.hip's arithmetic, bit for bit the same order per row) and
//     O_h = P V_h takes the V^T blocks as B; the wave writes its 16 x 64 slice of O into the A tile of the out-projection that
//     follows (the same two k-tiles its Q fragments came from: no other wave touches them).
//   256 MFMAs per wave = what the separate core issues, at the ring's rate, without a launch, a ramp or the core's load phase.
//
// Shapes: l_d = 128, hidden 512, l_c = 128, self-attention heads 8 x dim_head 64, image blocks heads * dh = 512 with the folded
// value / query projections staged (vfold_kernel), no dropout, no tape: the inference forward at b * 8 > 128 row tiles.  Anything
// else runs the per-block chains (HN_NO_SELF_IN_CHAIN=1 forces them, HN_FORCE_SELF_IN_CHAIN=1 this kernel below its size gate: the A/B
// switches of tests/test_gpu_layer_chain.py).  Stage times, what was tried and what it cost: DESIGN.md 4.2, tools/lchain_profile.py.
#include "common.h"
#include <stddef.h>
#include <type_traits>

namespace hn {

namespace {

#include "chain_common.h"
constexpr int PRM = 128 + 2 * CHID + 128 + 4 * 128 + 128;     // b_out | b1 | b2 | ff gamma, beta | projection gamma, beta | y row (head 2)
constexpr int MAXBLK = LAYER_MAXBLK;     // blocks of a launch (two table entries per thread)
constexpr int STG = 8 * 256;             // per-wave 16 x 16 output staging tiles
constexpr int LDS_FIXED = 8 * WSLOT + 16 * ATILE + 4 * ATILE + CR * XP + 2 * MAXBLK + STG;
constexpr int NATT = 32;                 // K blocks + V^T blocks
constexpr int SC1 = 16;                  // cache-policy bit of a buffer load: agent scope (never served from the CU's L1)

template <int V>
using ic = std::integral_constant<int, V>;

#ifdef LCHAIN_PROFILE
// development only (tools/lchain_profile.py builds a private library with -DLCHAIN_PROFILE): 100 MHz time stamps of one workgroup
// at the stage boundaries of the last 8 launches, 64 stamps each
__device__ unsigned long long g_lchain_prof[8 * 64];
__device__ int g_lchain_seq;
#define LC_PROF(i)                                                                                              \
  do {                                                                                                          \
    if (blockIdx.x == 100 % gridDim.x && threadIdx.x == 0 && (i) < 63)                                          \
      g_lchain_prof[(g_lchain_seq & 7) * 64 + (i)] = __builtin_amdgcn_s_memrealtime();                          \
  } while (0)
#else
#define LC_PROF(i)
#endif

}  // namespace

__global__ __launch_bounds__(512) void latent_layer_kernel(const LayerChainArgs args) {
  const gf32 *const a_x_in = (const gf32 *)args.x_in;
  const gf32 *const a_Opart = (const gf32 *)args.Opart;
  const gf32 *const a_Mpart = (const gf32 *)args.Mpart;
  const gf32 *const a_Lpart = (const gf32 *)args.Lpart;
  const gf32 *const a_wvf = (const gf32 *)args.wvf;
  const int a_nsplit = args.nsplit, a_Lp = args.Lp, a_heads = args.heads, a_dh = args.dh;
  const int nseg = args.nseg;
  constexpr int a_L = 128;
  LC_PROF(0);
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  lf32 *lds = (lf32 *)lds_raw;
  constexpr int Wr = 0;                     // [8 waves][WSLOT]  per-wave transpose slot of the block stream
  constexpr int Abig = Wr + 8 * WSLOT;      // [16][ATILE]  A of the out-projection / FF hidden tile / Q tile of the self-attention
  constexpr int Ahat = Abig + 16 * ATILE;   // [4][ATILE]   LayerNorm-ed x
  constexpr int xs = Ahat + 4 * ATILE;      // [CR][XP]     the x tile
  constexpr int tbl = xs + CR * XP;         // [MAXBLK] descriptor words 0-1 of every block
  constexpr int stgb = tbl + 2 * MAXBLK;    // [8][256]     per-wave output staging
  constexpr int prm0 = stgb + STG;          // [nseg][PRM]  small parameters of every segment
  constexpr int o_bout = 0, o_b1 = 128, o_b2 = o_b1 + 2 * CHID, o_fnw = o_b2 + 128, o_fnb = o_fnw + 128, o_pnw = o_fnb + 128,
                o_pnb = o_pnw + 128, o_y = o_pnb + 128;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fg = lane >> 4, fi = lane & 15;
  // workgroup -> (row tile of the sample = member, sample): the 8 tiles of a sample share an XCD and are adjacent in its dispatch
  // order (cluster_decode with C = 8)
  const int nsamp_grid = gridDim.x >> 3;
  int member_, samp_;
  cluster_decode((int)blockIdx.x, 8, nsamp_grid, 0, member_, samp_);
  const int member = __builtin_amdgcn_readfirstlane(member_), samp = __builtin_amdgcn_readfirstlane(samp_);
  if (samp >= args.b) return;               // grids are rounded up to 8 samples: idle workgroups
  const int m0 = (samp * 8 + member) * CR;

  auto seg_i = [&](int s, size_t field) { return late_kernarg<int>(offsetof(LayerChainArgs, seg) + (size_t)s * sizeof(LSeg) + field); };
  auto seg_p = [&](int s, size_t field) {
    return (const gf32 *)late_kernarg<const float *>(offsetof(LayerChainArgs, seg) + (size_t)s * sizeof(LSeg) + field);
  };

  // ---- the block table: thread t describes block t of the launch.  Per segment: [OUT 16] [FF1 32] [FF2 16] then
  //   proj 1: [K|V chunks 8 x 4] [Q chunks 4 x 4, head by head] [K blocks 16] [V^T blocks 16]   (the ring runs from the Q blocks straight into the K blocks)
  //   proj 2: [Q 4]
  for (int bi = tid; bi < MAXBLK; bi += 512) {
    int sfound = 0, local = bi, found = 0;
    for (int s = 0; s < nseg; ++s) {
      const int hd = seg_i(s, offsetof(LSeg, head)), pj = seg_i(s, offsetof(LSeg, proj));
      const int n = ((hd == 3 || hd == 4) ? 16 : 0) + 48 + (pj == 1 ? 48 + NATT : pj == 2 ? 4 : 0);
      if (!found) {
        if (local < n) { sfound = s; found = 1; }
        else local -= n;
      }
    }
    if (!found) { sfound = nseg - 1; local = -1; }      // past the end: the launch's last block again (requested, never consumed)
    const int s = sfound;
    const int hd = seg_i(s, offsetof(LSeg, head)), pj = seg_i(s, offsetof(LSeg, proj));
    const int n_out = (hd == 3 || hd == 4) ? 16 : 0;
    const int n_all = n_out + 48 + (pj == 1 ? 48 + NATT : pj == 2 ? 4 : 0);
    if (local < 0) local = n_all - 1;
    const float *W;
    long rb;
    int k, ldw;
    unsigned long long desc;
    if (local < n_out) {
      W = (const float *)seg_p(s, offsetof(LSeg, w_out)); rb = 0; k = local; ldw = 512;
    } else if (local < n_out + 32) {
      const int l2 = local - n_out, hc = l2 >> 3;                      // per hidden chunk: value rows (4 k-blocks), gate rows (4)
      W = (const float *)seg_p(s, offsetof(LSeg, w1)); rb = ((l2 >> 2) & 1) * CHID + hc * WN; k = l2 & 3; ldw = CD;
    } else if (local < n_out + 48) {
      const int l2 = local - n_out - 32;
      W = (const float *)seg_p(s, offsetof(LSeg, w2)); rb = 0; k = l2; ldw = CHID;
    } else {
      int l2 = local - n_out - 48;
      if (pj == 2) {
        W = (const float *)seg_p(s, offsetof(LSeg, wq)); rb = 0; k = l2 & 3; ldw = CD;
      } else {
        if (l2 < 32) { W = (const float *)seg_p(s, offsetof(LSeg, wkv)); rb = (long)(l2 >> 2) * WN; k = l2 & 3; ldw = CD; }
        else if (l2 < 48) { W = (const float *)seg_p(s, offsetof(LSeg, wq)); rb = (long)((l2 - 32) >> 2) * 16; k = l2 & 3; ldw = CD; }      // wave w: rows 64 w + 16 c + ..
        else {
          // attention blocks: slot of this segment's K / V^T images, this sample
          const int slot = seg_i(s, offsetof(LSeg, kv_slot));
          const int a2 = l2 - 48;
          if (a2 < 16) {                     // K block (token tile tt, k-half kk): rows = tokens of head `wave`, 64 floats apart
            const int tt = a2 >> 1, kk = a2 & 1;
            W = args.kbuf + (long)slot * args.kv_stride + (long)samp * (8 * 128 * 64) + tt * 16 * 64; rb = 0; k = kk; ldw = 64;
          } else {                           // V^T block (dim tile dt, token chunk tc): rows = dims of head `wave`, 128 floats apart
            const int dt = (a2 - 16) >> 2, tc = (a2 - 16) & 3;
            W = args.vtbuf + (long)slot * args.kv_stride + (long)samp * (8 * 64 * 128) + dt * 16 * 128; rb = 0; k = tc; ldw = 128;
          }
        }
      }
    }
    const unsigned long long addr = (unsigned long long)(W + rb * ldw + k * WK);
    desc = (addr & 0x0000ffffffffffffull) | ((unsigned long long)(ldw * 4) << 48);
    *(__attribute__((address_space(3))) unsigned long long *)(lds + tbl + 2 * bi) = desc;
  }
  const int r8 = lane >> 3, pos = lane & 7;
  const int wslot = Wr + wave * WSLOT + r8 * WK + ((pos ^ (r8 & 7)) * 4);
  const int pos16 = pos * 16;
  const int vrow = wave * 16 + r8;          // weights: the wave's 16 rows of the 128-row block
  const int vrowK = wave * 128 + r8;        // K image: head `wave`, token r8 of the block's 16
  const int vrowV = wave * 64 + r8;         // V^T image: head `wave`, dim r8 of the block's 16 (and the Q weights: row 64 w + 16 c + r8)
  unsigned long long ent = 0;
  int tp = tbl;
  asm volatile("" : "+v"(tp));
  auto fetch_entry = [&]() {
    ent = *(const __attribute__((address_space(3))) unsigned long long *)(lds + tp);
    tp += 2;
  };
  auto load2 = [&](float4 (&r)[2], int vidx, auto aux) {
    i32x4 rs;
    rs.x = (int)__builtin_amdgcn_readfirstlane((unsigned)ent);
    rs.y = (int)__builtin_amdgcn_readfirstlane((unsigned)(ent >> 32));
    rs.z = 1024;                            // records: 128 weight rows, 8 x 128 K rows, 8 x 64 V^T rows
    rs.w = 0x00020000;
    const f32x4 v0 = hn_sbuffer_load_x4(rs, vidx, pos16, 0, decltype(aux)::value), v1 = hn_sbuffer_load_x4(rs, vidx + 8, pos16, 0, decltype(aux)::value);
    r[0] = make_float4(v0.x, v0.y, v0.z, v0.w);
    r[1] = make_float4(v1.x, v1.y, v1.z, v1.w);
  };
  auto park = [&](const float4 (&r)[2]) {
    lst4(lds, wslot, r[0]);
    lst4(lds, wslot + 8 * WK, r[1]);
  };

  __syncthreads();                           // the block table
  LC_PROF(1);
  fetch_entry();
  float4 Bp[2], B0[2], B1[2], B2[2], B3[2];
  auto issue_w = [&](float4 (&r)[2]) { load2(r, vrow, ic<0>{}); fetch_entry(); };
  issue_w(Bp);
  issue_w(B0);
  issue_w(B1);
  issue_w(B2);
  issue_w(B3);
  // small parameters of every segment (PRM / 4 = 480 pieces each: thread q < 480 takes piece q of EVERY segment -- the segment index is
  // uniform, so the pointers are scalar loads from the argument segment) and the x tile: all requests first, then the LDS stores.  As
  // a strided loop with a per-lane segment index this was two dependent round trips per pass (pointer, then data), four passes for a
  // launch of four segments: 4.6 us of prologue against 2.5 for two segments (tools/lchain_profile.py, round 6)
  {
    float4 pv[LSEG_MAX];
    bool have[LSEG_MAX];
    const int q = tid;
#pragma unroll
    for (int sg2 = 0; sg2 < LSEG_MAX; ++sg2) {
      have[sg2] = false;
      if (sg2 < nseg && q < PRM / 4) {
        const gf32 *src = nullptr;
        if (q < 32) { const gf32 *p = seg_p(sg2, offsetof(LSeg, b_out)); src = p ? p + 4 * q : nullptr; }
        else if (q < 288) src = seg_p(sg2, offsetof(LSeg, b1)) + 4 * (q - 32);
        else if (q < 320) src = seg_p(sg2, offsetof(LSeg, b2)) + 4 * (q - 288);
        else if (q < 352) { const gf32 *p = seg_p(sg2, offsetof(LSeg, f_nw)); src = p ? p + 4 * (q - 320) : nullptr; }
        else if (q < 384) { const gf32 *p = seg_p(sg2, offsetof(LSeg, f_nb)); src = p ? p + 4 * (q - 352) : nullptr; }
        else if (q < 416) { const gf32 *p = seg_p(sg2, offsetof(LSeg, p_nw)); src = p ? p + 4 * (q - 384) : nullptr; }
        else if (q < 448) { const gf32 *p = seg_p(sg2, offsetof(LSeg, p_nb)); src = p ? p + 4 * (q - 416) : nullptr; }
        else { const gf32 *p = seg_p(sg2, offsetof(LSeg, y)); src = p ? p + (long)samp * CD + 4 * (q - 448) : nullptr; }
        if (src) { pv[sg2] = gld4(src); have[sg2] = true; }
      }
    }
    const int row = tid >> 5, l32 = tid & 31;
    const float4 xv = gld4(a_x_in + (long)(m0 + row) * CD + 4 * l32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sg2 = 0; sg2 < LSEG_MAX; ++sg2)
      if (have[sg2]) lst4(lds, prm0 + sg2 * PRM + 4 * q, pv[sg2]);
    lst4(lds, xs + row * XP + 4 * l32, xv);
  }
  // ---- head 3 of segment 0: merge the split partials of the shared-context core, folded value projection (chain.hip).  Behind the
  // ring start and the parameter requests, not in front of them as in chain.hip (whose budget is 128 registers): one round trip for
  // all three instead of two in a row (entry -> table 5.0 us with the merge against 1.3 without: tools/lchain_profile.py)
  const int head0 = seg_i(0, offsetof(LSeg, head));
  if (head0 == 3 && wave < a_heads) {
    const int i = lane & 15, gq = lane >> 4;
    const int q = member * CR + i;
    const long prow = ((long)(samp * a_heads + wave) * a_nsplit) * a_Lp + q;
    float M = -3.0e38f, l = 0.0f;
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bw[4];
    {
      const gf32 *wbase = a_wvf + ((long)(wave * a_dh + i) * 16 + 4 * gq);
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
      for (int s = 0; s < CHAIN_MERGE_GROUP; ++s) {
        const long pr = prow + (long)min(s0 + s, a_nsplit - 1) * a_Lp;
        mv[s] = gld1(a_Mpart + pr);
        lv[s] = gld1(a_Lpart + pr);
        ov[s] = gld4(a_Opart + pr * 16 + 4 * gq);
      }
      __builtin_amdgcn_sched_barrier(0);
      float Mn = M;
#pragma unroll
      for (int s = 0; s < CHAIN_MERGE_GROUP; ++s) Mn = fmaxf(Mn, s0 + s < a_nsplit ? mv[s] : -3.0e38f);
      const float sc = __builtin_amdgcn_exp2f(M - Mn);
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
    if (args.stats3 && gq == 0) {
      gf32 *st = (gf32 *)args.stats3 + ((long)(samp * a_heads + wave) * a_L + q) * 2;
      gst1(st, M);
      gst1(st + 1, l);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      if (ct >= (a_dh >> 4)) break;
      const float4 b0 = bw[ct];
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(o0.x, b0.x, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(o0.y, b0.y, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(o0.z, b0.z, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(o0.w, b0.w, c, 0, 0, 0);
      const int col = wave * a_dh + 16 * ct + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * gq + r;
        lds[Abig + (col >> 5) * ATILE + row * WK + ((((col & 31) >> 2) ^ (row & 7)) * 4) + (col & 3)] = c[r];
      }
    }
  }
  park(Bp);
  __syncthreads();
  LC_PROF(2);

  float4 fa0[2], fa1[2], fb0[2], fb1[2];
  auto read_a = [&](float4 (&f)[2], int A, int kt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) f[s2] = lld4(lds, A + kt * ATILE + fi * WK + (((4 * s2 + fg) ^ (fi & 7)) * 4));
  };
  auto read_b = [&](float4 (&f)[2]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) f[s2] = lld4(lds, Wr + wave * WSLOT + fi * WK + (((4 * s2 + fg) ^ (fi & 7)) * 4));
  };
  read_b(fb0);

  // One block of a GEMM stage (chain.hip `step`, slot by slot).  MODE 0: A fragments from an LDS tile, re-read for the next block;
  // MODE 1: operands swapped (A = the streamed block: S^T = K Q^T), the other side's fragments live in registers; MODE 2: A = the
  // probabilities in registers, B = the streamed block.  `vidx` / AUX: row index register and cache policy of the request for
  // block t + 5 (weights, K image, V^T image).
#define LC_SB __builtin_amdgcn_sched_barrier(0)
#define LC_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)
  auto step = [&](auto mode, auto aux, float4 (&Bq)[2], const float4 (&fa)[2], const float4 (&fb)[2], float4 (&fan)[2], float4 (&fbn)[2], int A,
                  int kt_next, int vidx, f32x4 &c0, f32x4 &c1) {
    constexpr int MODE = decltype(mode)::value;
    constexpr int AUX = decltype(aux)::value;
    if constexpr (MODE == 1) { LC_MFMA(c0, fb[0].x, fa[0].x); } else { LC_MFMA(c0, fa[0].x, fb[0].x); }
    LC_SB;
    lst4(lds, wslot, Bq[0]); LC_SB;
    if constexpr (MODE == 1) { LC_MFMA(c1, fb[1].x, fa[1].x); } else { LC_MFMA(c1, fa[1].x, fb[1].x); }
    LC_SB;
    lst4(lds, wslot + 8 * WK, Bq[1]); LC_SB;
    if constexpr (MODE == 1) { LC_MFMA(c0, fb[0].y, fa[0].y); } else { LC_MFMA(c0, fa[0].y, fb[0].y); }
    LC_SB;
    read_b(fbn); LC_SB;
    if constexpr (MODE == 1) { LC_MFMA(c1, fb[1].y, fa[1].y); } else { LC_MFMA(c1, fa[1].y, fb[1].y); }
    LC_SB;
    if constexpr (MODE == 0) { read_a(fan, A, kt_next); LC_SB; }
    if constexpr (MODE == 1) { LC_MFMA(c0, fb[0].z, fa[0].z); } else { LC_MFMA(c0, fa[0].z, fb[0].z); }
    LC_SB;
    i32x4 rs;
    rs.x = (int)__builtin_amdgcn_readfirstlane((unsigned)ent);
    rs.y = (int)__builtin_amdgcn_readfirstlane((unsigned)(ent >> 32));
    rs.z = 1024;
    rs.w = 0x00020000;
    {
      const f32x4 v0 = hn_sbuffer_load_x4(rs, vidx, pos16, 0, AUX);
      Bq[0] = make_float4(v0.x, v0.y, v0.z, v0.w);
    }
    LC_SB;
    if constexpr (MODE == 1) { LC_MFMA(c1, fb[1].z, fa[1].z); } else { LC_MFMA(c1, fa[1].z, fb[1].z); }
    LC_SB;
    {
      const f32x4 v1 = hn_sbuffer_load_x4(rs, vidx + 8, pos16, 0, AUX);
      Bq[1] = make_float4(v1.x, v1.y, v1.z, v1.w);
    }
    fetch_entry(); LC_SB;
    if constexpr (MODE == 1) { LC_MFMA(c0, fb[0].w, fa[0].w); } else { LC_MFMA(c0, fa[0].w, fb[0].w); }
    LC_SB;
    if constexpr (MODE == 1) { LC_MFMA(c1, fb[1].w, fa[1].w); } else { LC_MFMA(c1, fa[1].w, fb[1].w); }
    LC_SB;
  };
  auto run4 = [&](int A, int kc, int nk, f32x4 &c0, f32x4 &c1) {
    step(ic<0>{}, ic<0>{}, B0, fa0, fb0, fa1, fb1, A, kc + 1, vrow, c0, c1);
    step(ic<0>{}, ic<0>{}, B1, fa1, fb1, fa0, fb0, A, kc + 2, vrow, c0, c1);
    step(ic<0>{}, ic<0>{}, B2, fa0, fb0, fa1, fb1, A, kc + 3, vrow, c0, c1);
    step(ic<0>{}, ic<0>{}, B3, fa1, fb1, fa0, fb0, A, kc + 4 == nk ? 0 : kc + 4, vrow, c0, c1);
  };
  // ... with the row index register / cache policy of each step's request given (stage transitions into other images)
  auto run4x = [&](auto x0, auto x1, auto x2, auto x3, int A, int v0, int v1, int v2, int v3, f32x4 &c0, f32x4 &c1) {
    step(ic<0>{}, x0, B0, fa0, fb0, fa1, fb1, A, 1, v0, c0, c1);
    step(ic<0>{}, x1, B1, fa1, fb1, fa0, fb0, A, 2, v1, c0, c1);
    step(ic<0>{}, x2, B2, fa0, fb0, fa1, fb1, A, 3, v2, c0, c1);
    step(ic<0>{}, x3, B3, fa1, fb1, fa0, fb0, A, 0, v3, c0, c1);
  };
  auto run_chunk_k128 = [&](int A, f32x4 &c0, f32x4 &c1) { run4(A, 0, 4, c0, c1); };
  auto run_chunk_k512 = [&](int A, f32x4 &c0, f32x4 &c1) {
#pragma unroll
    for (int kc = 0; kc < 16; kc += 4) run4(A, kc, 16, c0, c1);
  };
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  auto layer_norm = [&](bool affine, int gamma, int beta, int yoff) {
    const int row = tid >> 5, l32 = tid & 31;
    float4 v = lld4(lds, xs + row * XP + 4 * l32);
    if (yoff >= 0) {                         // head 2: the one-token block's output row, added where the tile is read anyway
      const float4 y0 = lld4(lds, yoff + 4 * l32);
      v.x += y0.x; v.y += y0.y; v.z += y0.z; v.w += y0.w;
      lst4(lds, xs + row * XP + 4 * l32, v);
    }
    if (affine) {
      const float mu = half_wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / CD);
      v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
      const float q = half_wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
      const float rs = 1.0f / sqrtf(q * (1.0f / CD) + 1e-5f);
      const float4 g0 = lld4(lds, gamma + 4 * l32), b0 = lld4(lds, beta + 4 * l32);
      v.x = v.x * rs * g0.x + b0.x; v.y = v.y * rs * g0.y + b0.y; v.z = v.z * rs * g0.z + b0.z; v.w = v.w * rs * g0.w + b0.w;
    }
    lst4(lds, Ahat + (l32 >> 3) * ATILE + row * WK + (((l32 & 7) ^ (row & 7)) * 4), v);
  };
  const int ncol = wave * 16 + fi;
  int hid_at[4];                            // A-layout address of element (row 4 fg + r, column ncol) of a 16 x 512 tile's first 128 columns
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * fg + r;
    hid_at[r] = Abig + (ncol >> 5) * ATILE + row * WK + ((((ncol & 31) >> 2) ^ (row & 7)) * 4) + (ncol & 3);
  }
  const int stg = stgb + wave * 256;
  int seq = args.seq;
  int lostw = 0;                            // this wave gave up waiting for a sibling's K / V (sticky to the end of the launch)

  for (int sgi = 0; sgi < nseg; ++sgi) {
    const int prm = prm0 + sgi * PRM;
    const int hd = seg_i(sgi, offsetof(LSeg, head));
#define LC_PS(j) LC_PROF(3 + sgi * 14 + (j))
    LC_PS(0);
    // ================= OUT: x += LeakyReLU(O W_out^T + b_out), O in the A tile (head 3: merged above; head 4: the attention below) =====
    if (hd == 3 || hd == 4) {
      f32x4 c0 = zero, c1 = zero;
      read_a(fa0, Abig, 0);
      run_chunk_k512(Abig, c0, c1);
      const float bv = lds[prm + o_bout + ncol];
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = v[r] + bv;
        p = p > 0.0f ? p : 0.01f * p;
        lds[xs + (4 * fg + r) * XP + ncol] += p;
      }
      __syncthreads();
    }
    LC_PS(1);
    // ================= FF: x += (a * gate(g)) W2^T + b2,  [a | g] = LN(x) W1^T + b1 =================
    {
      layer_norm(seg_p(sgi, offsetof(LSeg, f_nw)) != nullptr, prm + o_fnw, prm + o_fnb, hd == 2 ? prm + o_y : -1);
      __syncthreads();
      LC_PS(2);
      read_a(fa0, Ahat, 0);
      const int a_gate = seg_i(sgi, offsetof(LSeg, gate));
      for (int hc = 0; hc < 4; ++hc) {
        f32x4 a0 = zero, a1 = zero, g0 = zero, g1 = zero;
        run_chunk_k128(Ahat, a0, a1);
        run_chunk_k128(Ahat, g0, g1);
        const int h0 = hc * WN + ncol;
        const float ba = lds[prm + o_b1 + h0], bg = lds[prm + o_b1 + CHID + h0];
        const float va[4] = {a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w};
        const float vg[4] = {g0.x + g1.x, g0.y + g1.y, g0.z + g1.z, g0.w + g1.w};
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
      LC_PS(3);
      f32x4 c0 = zero, c1 = zero;
      read_a(fa0, Abig, 0);
      run_chunk_k512(Abig, c0, c1);
      const float bv = lds[prm + o_b2 + ncol];
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[xs + (4 * fg + r) * XP + ncol] += v[r] + bv;
      __syncthreads();
      LC_PS(4);
    }
    {
      gf32 *xo = (gf32 *)seg_p(sgi, offsetof(LSeg, x_out));
      if (xo) {
        const int row = tid >> 5, l32 = tid & 31;
        gst4_nt(xo + (long)(m0 + row) * CD + 4 * l32, lld4(lds, xs + row * XP + 4 * l32));
      }
    }
    const int pj = seg_i(sgi, offsetof(LSeg, proj));
    if (pj == 0) continue;
    layer_norm(seg_p(sgi, offsetof(LSeg, p_nw)) != nullptr, prm + o_pnw, prm + o_pnb, -1);
    __syncthreads();
    LC_PS(5);
    read_a(fa0, Ahat, 0);
    if (pj == 2) {
      // ---- folded query of the next shared-context block (chain.hip): wave w = head w's 16 packed slots + the row's score bound
      f32x4 c0 = zero, c1 = zero;
      run_chunk_k128(Ahat, c0, c1);
      const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[stg + (4 * fg + r) * 16 + fi] = v[r];
      const int srow = lane >> 2, c4 = lane & 3;
      const int q = member * CR + srow;
      gst4_nt((gf32 *)args.qf + (((long)samp * 8 + wave) * a_L + q) * 16 + 4 * c4, lld4(lds, stg + srow * 16 + 4 * c4));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float ss = v[r] * v[r];
        ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4); ss += __shfl_xor(ss, 8);
        if (fi == 0) {
          const int q2 = member * CR + 4 * fg + r;
          const float bq = sqrtf(ss * (float)args.qf_D) * 1.00002f + 1e-6f;
          gst1((gf32 *)args.qf_bound + ((long)samp * 8 + wave) * a_L + q2, bq);
          if (bq > 60.0f) atomicOr(args.qf_flag, 1);
        }
      }
      LC_PS(6);
      continue;
    }
    // ================= latent self-attention: K | V chunks, Q chunks, exchange, S^T = K Q^T, softmax, O = P V =================
    const int slot = seg_i(sgi, offsetof(LSeg, kv_slot));
    {
      gf32 *kb = (gf32 *)args.kbuf + (long)slot * args.kv_stride + (long)samp * (8 * 128 * 64);
      gf32 *vb = (gf32 *)args.vtbuf + (long)slot * args.kv_stride + (long)samp * (8 * 64 * 128);
      const int t0 = member * CR;
      for (int j = 0; j < 8; ++j) {
        f32x4 c0 = zero, c1 = zero;
        // (the requests of the last five steps are Q blocks: wave w streams the rows of ITS head, 64 w + 16 c + ..)
        const int vi3 = j == 7 ? vrowV : vrow, vi1 = j >= 6 ? vrowV : vrow;
        run4x(ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, Ahat, vi3, vi3, vi3, vi1, c0, c1);
        const float4 v = make_float4(c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w);
        const int hh = 2 * (j & 3) + (wave >> 2), d0 = (wave & 3) * 16;      // head and first dim of this wave's 16 columns
        if (j < 4) {
          // K rows head-major: through the staging tile to one 16-byte store per lane (16 rows x 64 contiguous bytes per wave)
          lds[stg + (4 * fg + 0) * 16 + fi] = v.x; lds[stg + (4 * fg + 1) * 16 + fi] = v.y;
          lds[stg + (4 * fg + 2) * 16 + fi] = v.z; lds[stg + (4 * fg + 3) * 16 + fi] = v.w;
          int ln = lane;
          asm volatile("" : "+v"(ln));       // (address arithmetic stays here: see the statistics store below)
          const int srow = ln >> 2, c4 = ln & 3;
          gst4(kb + ((long)hh * 128 + t0 + srow) * 64 + d0 + 4 * c4, lld4(lds, stg + srow * 16 + 4 * c4));
        } else {
          // V transposed: the accumulator IS four consecutive tokens of dim fi
          int ln = lane;
          asm volatile("" : "+v"(ln));
          gst4(vb + ((long)hh * 64 + d0 + (ln & 15)) * 128 + t0 + 4 * (ln >> 4), v);
        }
      }
      LC_PS(6);
      // ---- exchange, first half: every store of this wave acknowledged by the L2 (vmcnt counts stores; the ring's requests in flight
      // are waited for with them: one load latency), barrier, the TILE's flag.  (One flag per wave instead -- 64 per sample, no barrier
      // here -- was slower: 512 lanes per workgroup polling two cache lines of one L2 channel held up the very stores they waited
      // for: 2.612 against 2.580 ms per forward.)  The Q chunks below are the work the flag's way to the siblings hides behind.
      LC_PS(7);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0 && !(late_kernarg<int>(offsetof(LayerChainArgs, inject_loss)) && member == 7))
        __hip_atomic_store(args.xflags + samp * 8 + member, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      LC_PS(8);
      // Q, head by head: wave w projects the 64 columns of head w (chunk c = its columns 16 c .. 16 c + 15: weight rows
      // 64 w + 16 c + .. through the V^T row index register), so the Q tile of a head is written and read by ONE wave: no barrier
      const float al = late_kernarg<float>(offsetof(LayerChainArgs, seg) + (size_t)sgi * sizeof(LSeg) + offsetof(LSeg, alpha_q));
      auto q_epilogue = [&](int c, const f32x4 &c0, const f32x4 &c1) {
        const float v[4] = {c0.x + c1.x, c0.y + c1.y, c0.z + c1.z, c0.w + c1.w};
        const int kt = 2 * wave + (c >> 1), cc = 16 * (c & 1) + fi;      // column 64 w + 16 c + fi: k-tile, column inside it
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * fg + r;
          lds[Abig + kt * ATILE + row * WK + (((cc >> 2) ^ (row & 7)) * 4) + (cc & 3)] = al * v[r];
        }
      };
      {
        f32x4 c0 = zero, c1 = zero;
        run4x(ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, Ahat, vrowV, vrowV, vrowV, vrowV, c0, c1);
        q_epilogue(0, c0, c1);
      }
      // the sample's 8 flags are REQUESTED here (lanes 0-7 of every wave: each wave decides for itself, no barrier) and looked at a
      // chunk later: a poll is an L2 round trip, and by then the siblings' flags are usually up
      int early = seq;
      if (lane < 8) early = __hip_atomic_load(args.xflags + samp * 8 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      {
        f32x4 c0 = zero, c1 = zero;
        run4x(ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{}, Ahat, vrowV, vrowV, vrowV, vrowV, c0, c1);
        q_epilogue(1, c0, c1);
      }
      // ---- exchange, second half: all 8 flags up (else the bounded wait of chain_common.h, lane by lane), BEFORE the first request of
      // a K block leaves -- the last step of Q chunk 2 asks for K block 0 (the ring's lead is five blocks).  A wave that gives up
      // poisons its slice of O below: every row of the tile becomes NaN in the out-projection, never a silently incomplete sum.
      LC_PS(9);
      if (__builtin_amdgcn_ballot_w64(early < seq) != 0) {
        int timed_out = 0;
        if (early < seq)
          timed_out = cluster_wait(args.xflags + samp * 8 + lane, seq, late_kernarg<unsigned>(offsetof(LayerChainArgs, wait_ticks)), args.xflags + args.flag_marker,
                                   late_kernarg<unsigned *>(offsetof(LayerChainArgs, status)), late_kernarg<unsigned>(offsetof(LayerChainArgs, token)));
        if (__builtin_amdgcn_ballot_w64(timed_out != 0) != 0) lostw = 1;
      }
      seq += 1;
      LC_PS(10);
      {
        f32x4 c0 = zero, c1 = zero;
        run4x(ic<0>{}, ic<0>{}, ic<0>{}, ic<SC1>{}, Ahat, vrowV, vrowV, vrowV, vrowK, c0, c1);
        q_epilogue(2, c0, c1);
        c0 = zero; c1 = zero;
        run4x(ic<SC1>{}, ic<SC1>{}, ic<SC1>{}, ic<SC1>{}, Ahat, vrowK, vrowK, vrowK, vrowK, c0, c1);
        q_epilogue(3, c0, c1);
      }
    }
    // the wave's Q fragments, both k-halves (B operand of S^T): k-tiles 2 w and 2 w + 1 of the Q tile, which this wave wrote itself;
    // fb0 holds K block 0's fragments
    float4 q0[2], q1[2];
    read_a(q0, Abig + 2 * wave * ATILE, 0);
    read_a(q1, Abig + 2 * wave * ATILE, 1);
    LC_PS(11);
    f32x4 S[8];
    {
      f32x4 e0, e1;
#define LC_S2(T, V0, V1, V2, V3)                                                                            \
      e0 = zero; e1 = zero;                                                                                 \
      step(ic<1>{}, ic<SC1>{}, B0, q0, fb0, fa1, fb1, 0, 0, V0, e0, e1);                                      \
      step(ic<1>{}, ic<SC1>{}, B1, q1, fb1, fa0, fb0, 0, 0, V1, e0, e1);                                      \
      S[T] = e0 + e1;                                                                                       \
      e0 = zero; e1 = zero;                                                                                 \
      step(ic<1>{}, ic<SC1>{}, B2, q0, fb0, fa1, fb1, 0, 0, V2, e0, e1);                                      \
      step(ic<1>{}, ic<SC1>{}, B3, q1, fb1, fa0, fb0, 0, 0, V3, e0, e1);                                      \
      S[T + 1] = e0 + e1;
      // the request of step t is block t + 5: K blocks up to t = 10, then the first V^T blocks
      LC_S2(0, vrowK, vrowK, vrowK, vrowK)
      LC_S2(2, vrowK, vrowK, vrowK, vrowK)
      LC_S2(4, vrowK, vrowK, vrowK, vrowV)
      LC_S2(6, vrowV, vrowV, vrowV, vrowV)
#undef LC_S2
    }
    LC_PS(12);
    // ---- softmax over the row's 128 tokens: 32 values here, the other 96 in the lanes (g', j) (self_attention.hip)
    float mx = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, S[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float l = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[t][r] = __builtin_amdgcn_exp2f(S[t][r] - mx);
        l += S[t][r];
      }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    {
      gf32 *st = (gf32 *)seg_p(sgi, offsetof(LSeg, stats));
      if (st && fg == 0) {                   // lane (0, j): query row j of the tile, head `wave`
        int fj = fi;
        asm volatile("" : "+v"(fj));         // (keeps the lane's 64-bit address out of the prologue: hoisted there it was SPILLED, and its
                                             // reload here -- a scratch load -- waited for the whole ring with vmcnt(0))
        st += ((long)(samp * 8 + wave) * a_L + member * CR + fj) * 2;
        gst1(st, mx);
        gst1(st + 1, l);
      }
    }
    float4 P[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) P[t] = make_float4(S[t][0] * inv, S[t][1] * inv, S[t][2] * inv, S[t][3] * inv);
    LC_PS(13);
    // ---- O = P V: per dim tile four blocks of 32 tokens; A = P[2 tc], P[2 tc + 1], B = the V^T block.  The requests of the last
    // five steps are the first weight blocks of the next segment.
    f32x4 O[4];
    {
      f32x4 e0, e1;
      float4 pa[2], pb[2];
#define LC_O4(D, V0, V1, V2, V3, X0, X1, X2, X3)                                                              \
      e0 = zero; e1 = zero;                                                                                 \
      pa[0] = P[0]; pa[1] = P[1];                                                                           \
      step(ic<2>{}, X0, B0, pa, fb0, fa1, fb1, 0, 0, V0, e0, e1);                                             \
      pb[0] = P[2]; pb[1] = P[3];                                                                           \
      step(ic<2>{}, X1, B1, pb, fb1, fa0, fb0, 0, 0, V1, e0, e1);                                             \
      pa[0] = P[4]; pa[1] = P[5];                                                                           \
      step(ic<2>{}, X2, B2, pa, fb0, fa1, fb1, 0, 0, V2, e0, e1);                                             \
      pb[0] = P[6]; pb[1] = P[7];                                                                           \
      step(ic<2>{}, X3, B3, pb, fb1, fa0, fb0, 0, 0, V3, e0, e1);                                             \
      O[D] = e0 + e1;
      LC_O4(0, vrowV, vrowV, vrowV, vrowV, ic<SC1>{}, ic<SC1>{}, ic<SC1>{}, ic<SC1>{})
      LC_O4(1, vrowV, vrowV, vrowV, vrowV, ic<SC1>{}, ic<SC1>{}, ic<SC1>{}, ic<SC1>{})
      LC_O4(2, vrowV, vrowV, vrowV, vrow, ic<SC1>{}, ic<SC1>{}, ic<SC1>{}, ic<0>{})
      LC_O4(3, vrow, vrow, vrow, vrow, ic<0>{}, ic<0>{}, ic<0>{}, ic<0>{})
#undef LC_O4
    }
    // ---- accumulator register r of lane (g, n): query row 4 g + r, column 64 w + 16 d + n of the out-projection's A tile (the
    // wave's own two k-tiles: nobody else reads or writes them)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int col = wave * 64 + 16 * d + fi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * fg + r;
        lds[Abig + (col >> 5) * ATILE + row * WK + ((((col & 31) >> 2) ^ (row & 7)) * 4) + (col & 3)] = lostw ? __builtin_nanf("") : O[d][r];
      }
    }
    __syncthreads();
  }
#ifdef LCHAIN_PROFILE
  if (blockIdx.x == 100 % gridDim.x && threadIdx.x == 0) {
    g_lchain_prof[(g_lchain_seq & 7) * 64 + 63] = (unsigned long long)nseg;
    __threadfence();
    g_lchain_seq = g_lchain_seq + 1;
  }
#endif
#undef LC_SB
#undef LC_MFMA
}

#ifdef LCHAIN_PROFILE
extern "C" __attribute__((visibility("default"))) int hn_debug_lchain_prof(unsigned long long *out, int *seq) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lchain_prof), sizeof(unsigned long long) * 512) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(seq, HIP_SYMBOL(g_lchain_seq), sizeof(int)) != hipSuccess) return -1;
  return 0;
}
#endif

int latent_layer_segment_blocks(int head, int proj) {
  return ((head == 3 || head == 4) ? 16 : 0) + 48 + (proj == 1 ? 48 + NATT : proj == 2 ? 4 : 0);
}

bool latent_layer_enabled() {
  static const bool off = getenv("HN_NO_SELF_IN_CHAIN") != nullptr;      // route switch (A/B): the per-block chains + the self core
  return !off;
}

size_t latent_layer_lds_bytes(int nseg) { return (size_t)(LDS_FIXED + nseg * PRM) * sizeof(float); }

int launch_latent_layer(const LayerChainArgs &a, hipStream_t s) {
  HN_REQUIRE(a.b >= 1 && a.nseg >= 1 && a.nseg <= LSEG_MAX && a.x_in && a.xflags, HN_E_SHAPE, "latent_layer: b=%d nseg=%d", a.b, a.nseg);
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  int nblk = 0, nself = 0;
  for (int i = 0; i < a.nseg; ++i) {
    const LSeg &g = a.seg[i];
    HN_REQUIRE(g.head == 0 || g.head == 2 || (g.head == 3 && i == 0) || (g.head == 4 && i > 0 && a.seg[i - 1].proj == 1), HN_E_SHAPE,
               "latent_layer: segment %d head=%d", i, g.head);
    HN_REQUIRE(g.w1 && g.b1 && g.w2 && g.b2 && al16(g.w1) && al16(g.w2) && al16(g.b1) && al16(g.b2), HN_E_NULL, "latent_layer: feed-forward operand of segment %d", i);
    HN_REQUIRE((g.f_nw == nullptr) == (g.f_nb == nullptr) && al16(g.f_nw) && al16(g.f_nb), HN_E_SHAPE, "latent_layer: feed-forward LayerNorm of segment %d", i);
    if (g.head == 3 || g.head == 4) HN_REQUIRE(g.w_out && g.b_out && al16(g.w_out) && al16(g.b_out), HN_E_NULL, "latent_layer: out-projection of segment %d", i);
    if (g.head == 2) HN_REQUIRE(g.y && al16(g.y), HN_E_NULL, "latent_layer: y of segment %d", i);
    HN_REQUIRE(g.proj == 0 || g.proj == 1 || (g.proj == 2 && i == a.nseg - 1), HN_E_SHAPE, "latent_layer: segment %d proj=%d", i, g.proj);
    HN_REQUIRE(g.proj != 1 || i + 1 < a.nseg, HN_E_SHAPE, "latent_layer: a self-attention segment must be followed by its out-projection");
    if (g.proj) HN_REQUIRE(g.wq && al16(g.wq) && (g.p_nw == nullptr) == (g.p_nb == nullptr) && al16(g.p_nw) && al16(g.p_nb), HN_E_NULL, "latent_layer: projection of segment %d", i);
    if (g.proj == 1) {
      HN_REQUIRE(g.wkv && al16(g.wkv) && a.kbuf && a.vtbuf && al16(a.kbuf) && al16(a.vtbuf) && g.kv_slot == nself && a.kv_stride % 4 == 0, HN_E_NULL,
                 "latent_layer: self-attention operands of segment %d", i);
      ++nself;
    }
    if (g.proj == 2) HN_REQUIRE(a.qf && a.qf_bound && a.qf_flag && a.qf_D >= 1 && al16(a.qf), HN_E_NULL, "latent_layer: folded query operands");
    if (g.x_out) HN_REQUIRE(al16(g.x_out), HN_E_SHAPE, "latent_layer: unaligned x_out");
    nblk += ((g.head == 3 || g.head == 4) ? 16 : 0) + 48 + (g.proj == 1 ? 48 + NATT : g.proj == 2 ? 4 : 0);
  }
  HN_REQUIRE(nblk + 6 <= MAXBLK, HN_E_SHAPE, "latent_layer: %d blocks", nblk);      // (the caller plans with latent_layer_segment_blocks)
  if (a.seg[0].head == 3)
    HN_REQUIRE(a.Opart && a.Mpart && a.Lpart && a.wvf && a.heads >= 1 && a.heads <= 8 && (a.dh == 16 || a.dh == 32 || a.dh == 64) && a.heads * a.dh == 512 &&
                   a.nsplit >= 1 && a.nsplit <= CHAIN_MERGE_MAX_SPLITS && a.Lp >= 128 && al16(a.Opart) && al16(a.wvf),
               HN_E_SHAPE, "latent_layer: merge head heads=%d dh=%d nsplit=%d", a.heads, a.dh, a.nsplit);
  const int nsamp = (a.b + 7) / 8 * 8;
  HN_REQUIRE(nsamp * 8 + 1 <= a.flag_count, HN_E_SHAPE, "latent_layer: %d flags for %d samples", a.flag_count, a.b);
  static bool configured[64] = {};
  const int lds_bytes = (int)latent_layer_lds_bytes(LSEG_MAX);
  int dev = 0;
  HN_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !configured[dev]) {
    HN_HIP_CHECK(hipFuncSetAttribute((const void *)latent_layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    if (dev >= 0 && dev < 64) configured[dev] = true;
  }
  LayerChainArgs ac = a;
  ClusterTicket t;
  cluster_before_launch(dev, s, &t);
  ac.status = t.status; ac.token = t.token; ac.wait_ticks = t.wait_ticks; ac.inject_loss = t.inject_loss;
  ac.flag_marker = nsamp * 8;
  hipLaunchKernelGGL(latent_layer_kernel, dim3(nsamp * 8), dim3(512), latent_layer_lds_bytes(a.nseg), s, ac);
  HN_LAUNCH_CHECK("latent_layer");
  cluster_after_launch(dev, s);
  return HN_OK;
}

}  // namespace hn
