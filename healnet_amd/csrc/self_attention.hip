// self_core_lds_kernel -- the attention core of the latent self-attention block (healnet/models/healnet.py:241-245 through
// Attention.forward :407-425) when the whole problem of a (sample, head) fits one workgroup: N = L <= 128 tokens, dim_head 64,
// no mask, no dropout, single split.
//
// The split-KV core (attention.hip) is built for one short query block against 10^4 .. 10^6 tokens: register-only, every wave
// fetches its K / V fragments from global memory.  Here that shape is wrong: 128 x 128 scores per (sample, head), 32 x 8 pairs
// at cfg2 b = 32 -- with two query tiles per wave that is ONE wave per SIMD, 20 dependent global loads per 16-token step and
// nothing to hide them behind (17.4 us per call, 0.39 of the fp32 MFMA rate of its 1.07 GF).  This kernel gives a pair to a
// workgroup of 8 waves (two per SIMD), stages K and V in LDS once (64 KB, 16-byte slots XOR-swizzled by token & 15) and lets wave
// w own query tile w:
//   S^T = K Q^T for all 8 token tiles (128 MFMAs, K fragments by ds_read_b128), every score of a row in registers, so the
//   softmax is the plain two-pass form (max, exp2, sum; no running max, no rescale) and P is normalised BEFORE the second
//   product; O = P V (128 MFMAs, V by ds_read_b32: lane (g, n) feeds V[4g + r][16 d + n]); O lands in its final layout.
// Same operand conventions as attn_core_kernel: queries pre-scaled to log2 units, A = K tile / B = Q tile for S^T so that the
// score registers are the A operand of P V as they are; stats = (row maximum, row sum) for hn_attn_probs.
#include "common.h"

namespace hn {

namespace {

constexpr int ST = 128;                 // tokens (and query rows) per (sample, head)

__device__ __forceinline__ float sel4(const float4 &v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

__global__ __launch_bounds__(512) void self_core_lds_kernel(AttnCoreArgs a) {
  __shared__ __attribute__((aligned(16))) float Ks[ST * 64];
  __shared__ __attribute__((aligned(16))) float Vs[ST * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int bh = blockIdx.x, bi = bh / a.h, hi = bh % a.h;
  const int L = a.Lq;
  const float *qbase = a.Q + (long)bi * a.q_b + (long)hi * a.q_h;
  const float *kbase = a.Kp + (long)bi * a.k_b + (long)hi * a.k_h;
  const float *vbase = a.Vp + (long)bi * a.v_b + (long)hi * a.v_h;

  // ---- requests first: the wave's query fragments (B operand: lane (g, j) holds Q[16 w + j][16 s + 4 g + c]) and the
  // workgroup's share of K / V (thread -> token tid >> 4 (+ 32 i), 16-byte piece tid & 15: full 256-byte rows per 16 lanes).
  // Everything goes through range-checked descriptors: rows past L / N read 0, no predicate exists.
  const i32x4 qrs = make_rsrc(qbase, (unsigned)(((long)(L - 1) * a.ldq + 64) * 4));
  const i32x4 krs = make_rsrc(kbase, (unsigned)(((long)(a.N - 1) * a.ldk + 64) * 4));
  const i32x4 vrs = make_rsrc(vbase, (unsigned)(((long)(a.N - 1) * a.ldv + 64) * 4));
  float4 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = buf4(qrs, ((wave * 16 + j) * a.ldq + 16 * s + 4 * g) * 4);
  {
    const int c = tid & 15, t0 = tid >> 4;
    float4 kk[4], vv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kk[i] = buf4(krs, ((t0 + 32 * i) * a.ldk + 4 * c) * 4);
      vv[i] = buf4(vrs, ((t0 + 32 * i) * a.ldv + 4 * c) * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tok = t0 + 32 * i;
      const int at = tok * 64 + ((c ^ (tok & 15)) * 4);
      *(float4 *)&Ks[at] = kk[i];
      *(float4 *)&Vs[at] = vv[i];
    }
  }
  __syncthreads();
  if (wave * 16 >= L) return;

  // ---- S^T = K Q^T: token tile t -> S[t]; lane (g, j) then holds the scores of query row j for tokens 16 t + 4 g + r.
  // Two tiles at a time: two independent chains of 16 MFMAs.
  f32x4 S[8];
#pragma unroll
  for (int t = 0; t < 8; t += 2) {
    float4 k0[4], k1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      k0[s] = *(const float4 *)&Ks[(16 * t + j) * 64 + (((4 * s + g) ^ j) * 4)];
      k1[s] = *(const float4 *)&Ks[(16 * t + 16 + j) * 64 + (((4 * s + g) ^ j) * 4)];
    }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sel4(k0[st >> 2], st & 3), sel4(qf[st >> 2], st & 3), s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sel4(k1[st >> 2], st & 3), sel4(qf[st >> 2], st & 3), s1, 0, 0, 0);
    }
    S[t] = s0;
    S[t + 1] = s1;
  }

  // ---- softmax over the 128 tokens of the row: 32 values here, the other 96 in the lanes (g', j)
  float mx = -__builtin_inff();
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (a.N < ST && 16 * t + 4 * g + r >= a.N) S[t][r] = -__builtin_inff();
      mx = fmaxf(mx, S[t][r]);
    }
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

  // ---- O = P V: A = P straight from the score registers (k-chunk of MFMA r = tokens {16 t + 4 g + r}), B = V rows 16 t + 4 g + r
  f32x4 O[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) O[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * t + 4 * g + r;       // row & 15 == 4 g + r
      const float p = S[t][r] * inv;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float v = Vs[row * 64 + (((4 * d + (j >> 2)) ^ (4 * g + r)) * 4) + (j & 3)];
        O[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, v, O[d], 0, 0, 0);
      }
    }
  }

  // ---- accumulator register r of lane (g, n) is query row 16 w + 4 g + r, column 16 d + n of the head
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = wave * 16 + 4 * g + r;
    if (q < L) {
      float *orow = a.Ofinal + ((long)bi * L + q) * a.ldo + hi * a.dh;
#pragma unroll
      for (int d = 0; d < 4; ++d) orow[16 * d + j] = O[d][r];
    }
  }
  if (a.stats && g == 0 && wave * 16 + j < L) {
    a.stats[((long)bh * L + wave * 16 + j) * 2 + 0] = mx;
    a.stats[((long)bh * L + wave * 16 + j) * 2 + 1] = l;
  }
}

}  // namespace

bool self_core_lds_eligible(const AttnCoreArgs &a) {
  static const bool force = getenv("HN_FORCE_SELF_LDS") != nullptr;
  static const bool off = tuning_env("HN_NO_SELF_LDS") != nullptr;       // development switch: the split-KV core instead
  // one workgroup per (sample, head): below ~3/4 of the CUs the split-KV core (one query tile per wave, spread over the chip)
  // wins -- cfg2 forward with / without: b = 8 1.232 / 1.216 ms, b = 32 2.966 / 2.974, b = 64 5.813 / 5.834, b = 128 11.48 / 11.51
  const bool enough = (long)a.b * a.h >= 192 || force;
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  return !off && enough && a.dp == 64 && a.dh == 64 && a.nsplit == 1 && a.Ofinal != nullptr && !a.ones_col && a.drop.thr == 0 && a.mask == nullptr &&
         a.N >= 1 && a.N <= ST && a.Lq >= 1 && a.Lq <= ST && a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldk >= 64 &&
         a.ldv >= 64 && a.ldq >= 64 && a.q_b % 4 == 0 && a.q_h % 4 == 0 && a.k_b % 4 == 0 && a.k_h % 4 == 0 && a.v_b % 4 == 0 &&
         a.v_h % 4 == 0 && al16(a.Q) && al16(a.Kp) && al16(a.Vp) && (long)a.b * a.h < (1L << 31);
}

int launch_self_core_lds(const AttnCoreArgs &a, hipStream_t s) {
  hipLaunchKernelGGL(self_core_lds_kernel, dim3((unsigned)(a.b * a.h)), dim3(512), 0, s, a);
  HN_LAUNCH_CHECK("self_core_lds");
  return HN_OK;
}

}  // namespace hn
