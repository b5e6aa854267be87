// The dQ half of the explicit-binding attention backward (dp = 64: patch bags; autograd of healnet/models/healnet.py:409-424) on a
// workgroup-shared LDS ring of K / V tiles (the dK / dV half with the QUERY side in LDS is attn_bwd_dkv_lds_kernel, attention_bwd.hip).
//
// attn_bwd_dq_kernel (attention_bwd.hip) is register-only: every wave fetches the K / V fragments of every 16-token step itself, as
// fragment-shaped loads (16 rows x 64 B per instruction -- the pattern that runs at a third of the full-line rate),
// although the four waves of a workgroup (its query-tile groups) walk the SAME tokens: 0.60 of the fp32 MFMA peak at cfg4 (134 us).
// Here the workgroup lands each 32-token K / V tile ONCE by LDS-DMA in full 256-byte rows (gemm_nt.hip's skeleton: two ring slots,
// one barrier per tile in the MIDDLE of the tile, inline-asm DMA with hand-counted waits) and its waves read their fragments from
// LDS: 134 -> 115 us.  (The FORWARD core was built on the same ring and measured: 81.8 against 79.2 us for attn_core_kernel -- its
// softmax needs the third resident wave per SIMD that 175 VGPRs do not leave -- and is not kept.)
//   LDS image of a tile: 32 rows (tokens) of 16 slots of 16 bytes, K rows then V rows; slot c of row r is stored at
//   c ^ sw(r & 15), sw(x) = ((x & 3) << 2) | (x >> 2).  With the channel contraction of S (and dP) ordered k = 16 g + 4 s + e
//   (lane group g reads slot 4 g + s of its token row) and the B fragments of P V / dS K taking ALL 16 slots of rows 4 g + r (lane j:
//   channels 4 j .. 4 j + 3), both read patterns are conflict-free, and MFMA e of a quad produces output channels {4 j + e}: a lane
//   ends up with four consecutive channels of its query rows (float4 partial stores in the natural layout).
#include "common.h"
#include <stdlib.h>
#pragma clang diagnostic ignored "-Winline-asm"

namespace hn {
namespace {

constexpr int TT = 32, DP = 64, DT = 4;
constexpr int SLOT = 2 * TT * DP;                 // floats per ring slot: K rows, then V rows (16 KB)

__device__ __forceinline__ void kv_glds16(const i32x4 &rsrc, unsigned lds_byte, int voffset, unsigned soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_byte), "v"(voffset), "s"(rsrc), "s"(soffset)
               : "memory", "m0");
}
template <int N> __device__ __forceinline__ void kv_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ float lexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// The ring: every wave of the workgroup issues 4 one-KB pieces per tile (K pieces w and w + 4, V pieces w and w + 4; a piece is 4
// token rows).  Lane (r4 = lane >> 4, p = lane & 15) fetches slot p ^ sw(row & 15) of its row and lands at slot p.
struct KvRing {
  i32x4 krs, vrs;
  unsigned lds_base, kbytes, vbytes;
  int vok, vov, ldk4, ldv4, wave;
  __device__ __forceinline__ void issue(int tile, int t_begin) const {
    const unsigned dst = lds_base + (unsigned)((tile & 1) * SLOT * 4 + wave * 1024);
    const int t0 = t_begin + tile * TT;
    // the scalar offset must stay inside the descriptor (the range check subtracts it from num_records): clamp, so a piece that
    // starts past the context reads zeros instead of wrapping
    const unsigned ks0 = min((unsigned)t0 * (unsigned)ldk4, kbytes), ks1 = min((unsigned)(t0 + 16) * (unsigned)ldk4, kbytes);
    const unsigned vs0 = min((unsigned)t0 * (unsigned)ldv4, vbytes), vs1 = min((unsigned)(t0 + 16) * (unsigned)ldv4, vbytes);
    kv_glds16(krs, dst, vok, ks0);
    kv_glds16(krs, dst + 4096, vok, ks1);
    kv_glds16(vrs, dst + TT * DP * 4, vov, vs0);
    kv_glds16(vrs, dst + TT * DP * 4 + 4096, vov, vs1);
  }
};

__device__ __forceinline__ KvRing make_ring(const float *kbase, const float *vbase, int N, int ldk, int ldv, const float *lds, int lane, int wave) {
  KvRing r;
  r.kbytes = (unsigned)(((long)(N - 1) * ldk + DP) * 4);
  r.vbytes = (unsigned)(((long)(N - 1) * ldv + DP) * 4);
  r.krs = make_rsrc(kbase, r.kbytes);
  r.vrs = make_rsrc(vbase, r.vbytes);
  r.lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds);
  const int r4 = lane >> 4, p = lane & 15, rl = 4 * wave + r4;      // row within a 16-row half tile (pieces w and w + 4: same rl)
  const int sw = ((rl & 3) << 2) | (rl >> 2);
  r.vok = (rl * ldk + ((p ^ sw) << 2)) * 4;
  r.vov = (rl * ldv + ((p ^ sw) << 2)) * 4;
  r.ldk4 = ldk * 4;
  r.ldv4 = ldv * 4;
  r.wave = wave;
  return r;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// dQ.  S^T = K Q^T and dP^T = V dO^T share the fragment shape (A = token rows from the ring, B = Q / dO fragments held in
// registers); l dS = 2^(s - m) (dP - D) stays in registers as the A operand of dQ += dS K, whose B fragments are rows 4 g + r of
// the SAME landed K tile (all 16 slots).  1 / l is applied to the finished rows.
// ------------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(256) void attn_bwd_dq_lds_kernel(AttnBwdArgs a, int ngroups, int gy, int waves_per_block) {
  __shared__ __attribute__((aligned(16))) float lds[2 * SLOT];
  const int L = a.Lq;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, j = lane & 15;
  long total = (long)gridDim.x, id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int split = (int)(id % a.nsplit);
  const int yb = (int)((id / a.nsplit) % gy);
  const int bh = (int)(id / ((long)a.nsplit * gy));
  const int qg = yb * waves_per_block + wave;
  const bool active = qg < ngroups;
  const int bi = bh / a.h, hi = bh % a.h;
  const int t_begin = split * a.chunk, t_end = min(a.N, t_begin + a.chunk);
  const int ntiles = (t_end - t_begin + TT - 1) / TT;

  const KvRing ring = make_ring(a.Kp + (long)bi * a.k_b + (long)hi * a.k_h, a.Vp + (long)bi * a.v_b + (long)hi * a.v_h, a.N, a.ldk, a.ldv,
                                lds, lane, wave);
  if (ntiles > 0) ring.issue(0, t_begin);

  float4 qf[NQ][DT], gf[NQ][DT];
  float invl[NQ];
  f32x4 negm[NQ], negd[NQ];
  {
    const i32x4 rsQ = make_rsrc(a.Q + (long)bi * a.q_b + (long)hi * a.q_h, rsrc_bytes(L, a.ldq, DP));
    const i32x4 rsG = make_rsrc(a.dO + (long)bi * a.do_b + (long)hi * a.do_h, rsrc_bytes(L, a.lddo, DP));
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int row = (qg * NQ + i) * 16 + j;
#pragma unroll
      for (int s = 0; s < DT; ++s) {
        qf[i][s] = buf4(rsQ, active ? (row * a.ldq + 16 * g + 4 * s) * 4 : 0x7ffffff0);
        gf[i][s] = buf4(rsG, active ? (row * a.lddo + 16 * g + 4 * s) * 4 : 0x7ffffff0);
      }
      const int rc = min(row, L - 1);
      const float mrow = a.stats[((long)bh * L + rc) * 2 + 0];
      invl[i] = 1.0f / a.stats[((long)bh * L + rc) * 2 + 1];
      const float drow = a.delta[(long)bh * L + rc];
      negm[i] = (f32x4){-mrow, -mrow, -mrow, -mrow};
      negd[i] = (f32x4){-drow, -drow, -drow, -drow};
    }
  }
  kv_wait_vmcnt<0>();
#pragma unroll
  for (int i = 0; i < NQ; ++i) asm volatile("" ::"v"(qf[i][DT - 1].w), "v"(gf[i][DT - 1].w), "v"(invl[i]), "v"(negd[i][0]), "v"(negm[i][0]));
  if (ntiles > 1) ring.issue(1, t_begin);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  f32x4 dQ[NQ][4];
#pragma unroll
  for (int i = 0; i < NQ; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) dQ[i][e] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragments of a half tile: K and V token rows as A operands (ka, va), K rows 4 g + r as the B operand of dQ (kb)
  struct DqFrags { float4 ka[DT], va[DT], kb[4]; };
  auto read_dq = [&](const float *slot, int sub, DqFrags &f) {
    const int swj = ((j & 3) << 2) | (j >> 2);
    const float *K = slot + (sub * 16) * DP, *V = slot + TT * DP + (sub * 16) * DP;
#pragma unroll
    for (int s = 0; s < DT; ++s) {
      f.ka[s] = *(const float4 *)&K[j * DP + (((4 * g + s) ^ swj) << 2)];
      f.va[s] = *(const float4 *)&V[j * DP + (((4 * g + s) ^ swj) << 2)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) f.kb[r] = *(const float4 *)&K[(4 * g + r) * DP + ((j ^ ((r << 2) | g)) << 2)];
  };
  auto half_step = [&](int t0, const DqFrags &f) {
    f32x4 S[NQ], dP[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) { S[i] = negm[i]; dP[i] = negd[i]; }
#pragma unroll
    for (int s = 0; s < DT; ++s) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        S[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ka[s].x, qf[i][s].x, S[i], 0, 0, 0);
        dP[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.va[s].x, gf[i][s].x, dP[i], 0, 0, 0);
        S[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ka[s].y, qf[i][s].y, S[i], 0, 0, 0);
        dP[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.va[s].y, gf[i][s].y, dP[i], 0, 0, 0);
        S[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ka[s].z, qf[i][s].z, S[i], 0, 0, 0);
        dP[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.va[s].z, gf[i][s].z, dP[i], 0, 0, 0);
        S[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ka[s].w, qf[i][s].w, S[i], 0, 0, 0);
        dP[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.va[s].w, gf[i][s].w, dP[i], 0, 0, 0);
      }
    }
    if (t0 + 16 > t_end) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (t0 + 4 * g + r >= t_end) {
#pragma unroll
          for (int i = 0; i < NQ; ++i) S[i][r] = -__builtin_inff();
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {          // l * dS  (2^-inf = 0 kills invalid tokens)
      const f32x4 e = {lexp2(S[i][0]), lexp2(S[i][1]), lexp2(S[i][2]), lexp2(S[i][3])};
      S[i] = e * dP[i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float kv[4] = {f.kb[r].x, f.kb[r].y, f.kb[r].z, f.kb[r].w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < NQ; ++i) dQ[i][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(S[i][r], kv[e], dQ[i][e], 0, 0, 0);
    }
  };

  DqFrags fa, fb;
  if (ntiles > 0) read_dq(lds, 0, fa);
  for (int kt = 0; kt < ntiles; ++kt) {
    const float *slot = lds + (kt & 1) * SLOT;
    const int t0 = t_begin + kt * TT;
    read_dq(slot, 1, fb);
    __builtin_amdgcn_sched_barrier(0);
    if (active) half_step(t0, fa);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < ntiles) {
      kv_wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 2 < ntiles) ring.issue(kt + 2, t_begin);
      read_dq(lds + ((kt + 1) & 1) * SLOT, 0, fa);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (active && t0 + 16 < t_end) half_step(t0 + 16, fb);
  }
  if (!active) return;
  const long prow = ((long)bh * a.nsplit + split) * a.Lp;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int tile = qg * NQ + i;
    if (tile * 16 < a.Lp) {
#pragma unroll
      for (int r = 0; r < 4; ++r)         // accumulator reg r of lane (g, .) belongs to query row 4 g + r, whose 1/l lives in lane 4 g + r
        *(f32x4 *)&a.dQpart[(prow + tile * 16 + 4 * g + r) * DP + 4 * j] =
            (f32x4){dQ[i][0][r], dQ[i][1][r], dQ[i][2][r], dQ[i][3][r]} * __shfl(invl[i], 4 * g + r);
    }
  }
}

bool attn_bwd_dq_lds_eligible(const AttnBwdArgs &a) {
  static const bool off = tuning_env("HN_NO_ATTN_LDS") != nullptr;
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  return !off && a.dp == 64 && a.Kp != a.Vp && a.drop.thr == 0 && a.mask == nullptr && a.qk_steps == 0 && a.N >= 256 && a.ldk % 4 == 0 &&
         a.ldv % 4 == 0 && a.k_b % 4 == 0 && a.k_h % 4 == 0 && a.v_b % 4 == 0 && a.v_h % 4 == 0 && a.ldq % 4 == 0 && a.lddo % 4 == 0 &&
         a.q_b % 4 == 0 && a.q_h % 4 == 0 && a.do_b % 4 == 0 && a.do_h % 4 == 0 && al16(a.Kp) && al16(a.Vp) && al16(a.Q) && al16(a.dO) &&
         al16(a.dQpart) && ((long)a.N * a.ldk + DP) * 4 < (1L << 31) && ((long)a.N * a.ldv + DP) * 4 < (1L << 31);
}

int launch_attn_bwd_dq_lds(const AttnBwdArgs &a, hipStream_t s) {
  const int tiles = a.Lp / 16;
  const int nq = tiles >= 8 ? 2 : 1;
  const int ngroups = ceil_div(tiles, nq);
  const int gy = ceil_div(ngroups, 4);
  const long blocks = (long)a.nsplit * gy * a.b * a.h;
  HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "attn_bwd_dq_lds: grid too large");
  if (nq == 2) hipLaunchKernelGGL(attn_bwd_dq_lds_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, a, ngroups, gy, 4);
  else hipLaunchKernelGGL(attn_bwd_dq_lds_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, a, ngroups, gy, 4);
  HN_LAUNCH_CHECK("attn_bwd_dq_lds");
  return HN_OK;
}

}  // namespace hn
