// K1: Fourier positional encode + concat + flatten (+ optional affine-free LayerNorm) for one modality.
//
// Replaces HealNet.forward :200-222 and fourier_encode :292-302 of the reference
// (healnet/models/healnet.py).  Pure HBM-bound elementwise work: every token is read once and
// written once; positions are a function of the token index, so no position tensor exists.
//
// Two kernels:
//   encode_token_kernel : one thread per token, for narrow modalities (D <= 32: images, volumes).
//                         Consecutive lanes own consecutive tokens, so the C-float reads and the
//                         ld_out-float writes of a wave form one contiguous span each.
//   encode_wave_kernel  : one 64-lane wave per token for wide modalities (tabular / patch bags,
//                         C up to thousands): lanes stride over channels, wave-level reductions for
//                         the LayerNorm moments.
#include "common.h"

namespace hn {

struct EncGeom {
  int n_axes;
  int S[HN_MAX_AXES];
  int C, F, D, ld_out;
  long N;
  float max_freq;
  int fourier, normalize;
  float eps;
  int ones_col;   // >= 0: this padding column is written as 1.0 (the attention core's synthetic ones column)
  // window along axis 0 (context split over ranks, healnet_amd.dist.context_parallel_forward): the data holds rows
  // [off0, off0 + S[0]) of an axis of full0 positions; off0 = 0, full0 = S[0] for a whole modality
  int off0, full0;
};

// torch.linspace(-1, 1, S)[i] in fp32 (ATen's symmetric formulation: ascending from the start for the
// first half, descending from the end for the second), healnet.py:212.
__device__ __forceinline__ float axis_pos(int i, int S) {
  if (S == 1) return -1.0f;
  float step = __fdiv_rn(2.0f, (float)(S - 1));
  if (i < S / 2) return __fadd_rn(-1.0f, __fmul_rn(step, (float)i));
  return __fsub_rn(1.0f, __fmul_rn(step, (float)(S - 1 - i)));
}

// torch.linspace(1, max_freq/2, F)[f], healnet.py:296
__device__ __forceinline__ float band_scale(int f, int F, float max_freq) {
  float end = max_freq * 0.5f;
  if (F == 1) return 1.0f;
  float step = __fdiv_rn(end - 1.0f, (float)(F - 1));
  if (f < F / 2) return __fadd_rn(1.0f, __fmul_rn(step, (float)f));
  return __fsub_rn(end, __fmul_rn(step, (float)(F - 1 - f)));
}

__device__ __forceinline__ void token_coords(long n, const EncGeom &g, int *idx) {
#pragma unroll
  for (int a = HN_MAX_AXES - 1; a >= 0; --a) {
    if (a < g.n_axes) {
      idx[a] = (int)(n % g.S[a]);
      n /= g.S[a];
    } else {
      idx[a] = 0;
    }
  }
}

// position of coordinate idx[a] on axis a (axis 0 may be a window of a longer axis)
__device__ __forceinline__ float axis_pos_of(int a, const int *idx, const EncGeom &g) {
  return a == 0 ? axis_pos(idx[0] + g.off0, g.full0) : axis_pos(idx[a], g.S[a]);
}

// positional feature j (0 <= j < n_axes*(2F+1)) of a token with coordinates idx
__device__ __forceinline__ float pos_feature(int j, const int *idx, const EncGeom &g) {
  const int per = 2 * g.F + 1;
  int a = j / per, w = j - a * per;
  float p = axis_pos_of(a, idx, g);
  if (w == 2 * g.F) return p;
  int f = (w < g.F) ? w : w - g.F;
  float arg = __fmul_rn(__fmul_rn(p, band_scale(f, g.F, g.max_freq)), 3.14159265358979323846f);  // (p*s)*pi, :299
  return (w < g.F) ? sinf(arg) : cosf(arg);
}

constexpr int kMaxNarrow = 32;

// modality elements: fp32 or bf16 (hn_modality_input.dtype); all arithmetic is fp32 either way
__device__ __forceinline__ float in_at(const float *p, long i) { return p[i]; }
__device__ __forceinline__ float in_at(const uint16_t *p, long i) { return __uint_as_float((unsigned)p[i] << 16); }
// HN_U8: 8-bit image transport; the value is byte / 255 in fp32, bit-identical to torchvision's ToTensor (`.div(255)`)
__device__ __forceinline__ float in_at(const uint8_t *p, long i) { return __fdiv_rn((float)p[i], 255.0f); }
__device__ __forceinline__ uint16_t to_bf16(float f) {     // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// PACK > 0: packed context layout (common.h): the D-1 kept channels go to packed_slot(c, PACK), the last channel is
// dropped (it is minus the sum of the others after normalisation).
// LD: compile-time bound of the output row (16 or 32 columns): every loop below runs over LD candidate channels
// SC / SA / SF > 0: channel count, axes and frequency bands known at compile time (RGB image: 3, 2, 2; RGB volume: 3, 3, 2) --
// every predicate and index computation of the generic formulation folds away
template <int PACK, typename IN, int LD, int SC = 0, int SA = 0, int SF = 0>
__device__ __forceinline__ void encode_token_body(const IN *__restrict__ data, float *__restrict__ out, EncGeom g, long total, const long block) {
  if (SC > 0) { g.C = SC; g.n_axes = SA; g.F = SF; g.D = SC + SA * (2 * SF + 1); }
  long gid = block * blockDim.x + threadIdx.x;
  // 16-float rows leave through LDS (below): every lane of a wave takes part, the ones past the end on the last token's data
  const bool staged_store = (LD == 16 || LD == 32) && g.ld_out == LD;
  const bool live = gid < total;
  if (!live) { if (!staged_store) return; gid = total - 1; }
  long n = gid % g.N;
  int idx[HN_MAX_AXES];
  token_coords(n, g, idx);
  float v[LD];
  const IN *src = data + gid * g.C;
  if (SC > 0) {
    // static shapes: one sincosf per (axis, band) -- the argument reduction is shared -- instead of a sinf and a cosf of the same
    // argument through pos_feature (whose per-column selects hide the pairing from the compiler)
#pragma unroll
    for (int c = 0; c < LD; ++c) v[c] = c < SC ? in_at(src, c) : 0.0f;
#pragma unroll
    for (int a = 0; a < SA; ++a) {
      const float p = axis_pos_of(a, idx, g);
#pragma unroll
      for (int f = 0; f < SF; ++f) {
        const float arg = __fmul_rn(__fmul_rn(p, band_scale(f, SF, g.max_freq)), 3.14159265358979323846f);  // (p*s)*pi, :299
        float sn, cs;
        sincosf(arg, &sn, &cs);
        v[SC + a * (2 * SF + 1) + f] = sn;
        v[SC + a * (2 * SF + 1) + SF + f] = cs;
      }
      v[SC + a * (2 * SF + 1) + 2 * SF] = p;
    }
  } else {
#pragma unroll
    for (int c = 0; c < LD; ++c) {
      float x = 0.0f;
      if (c < g.C) x = in_at(src, c);
      else if (c < g.D) x = pos_feature(c - g.C, idx, g);
      v[c] = x;
    }
  }
  if (g.normalize) {
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < LD; ++c) sum += (c < g.D) ? v[c] : 0.0f;
    float mean = sum / (float)g.D;
    float sq = 0.0f;
#pragma unroll
    for (int c = 0; c < LD; ++c) {
      float d = (c < g.D) ? v[c] - mean : 0.0f;
      sq += d * d;
    }
    float rstd = 1.0f / sqrtf(sq / (float)g.D + g.eps);
#pragma unroll
    for (int c = 0; c < LD; ++c) v[c] = (c < g.D) ? (v[c] - mean) * rstd : 0.0f;
  }
  if (PACK > 0) {
    float o[LD];
#pragma unroll
    for (int c = 0; c < LD; ++c) o[c] = 0.0f;
#pragma unroll
    for (int c = 0; c < (PACK <= 4 ? 4 * PACK : 16 + 4 * (PACK - 4)); ++c)
      if (c < g.D - 1) o[packed_slot(c, PACK)] = v[c];
#pragma unroll
    for (int c = 0; c < LD; ++c) v[c] = o[c];
  }
#pragma unroll
  for (int c = 0; c < LD; ++c)
    if (c == g.ones_col) v[c] = 1.0f;
  if ((LD == 16 || LD == 32) && staged_store) {
    // A lane holds one token's 64- / 128-byte row: stored directly, a wave instruction would write 64 pieces 64 / 128 bytes apart.
    // Through a per-wave LDS image instead (quad q of lane l at slot Q l + (q ^ sw(l)), Q = LD / 4 quads per row, sw = l >> 2 & 3 for
    // Q = 4, l >> 1 & 7 for Q = 8: conflict-free both ways), read back linearly, so that an instruction writes whole rows = 1 KB of
    // consecutive memory (cfg2: 36.1 -> 20.5 us for 19 MB in + 103 MB out = 6 TB/s).
    constexpr int Q = LD / 4;
    __shared__ float4 stage[4][64 * Q];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sw = Q == 4 ? (lane >> 2) & 3 : (lane >> 1) & 7;
#pragma unroll
    for (int q = 0; q < Q; ++q) stage[wv][Q * lane + (q ^ sw)] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    __builtin_amdgcn_wave_barrier();
    const long wave_tok0 = block * blockDim.x + 64 * wv;      // first token of this wave
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int slot = lane + 64 * k, tl = slot / Q;                     // token tl of the wave, position slot % Q of its row image
      const int q = (slot % Q) ^ (Q == 4 ? (tl >> 2) & 3 : (tl >> 1) & 7);
      const float4 val = stage[wv][slot];
      if (wave_tok0 + tl < total) *(float4 *)(out + (wave_tok0 + tl) * LD + 4 * q) = val;
    }
    return;
  }
  if (!live) return;
  float *dst = out + gid * (long)g.ld_out;
  if ((g.ld_out & 3) == 0) {
#pragma unroll
    for (int c = 0; c < LD; c += 4)
      if (c < g.ld_out) *(float4 *)(dst + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
  } else {
#pragma unroll
    for (int c = 0; c < LD; ++c)
      if (c < g.ld_out) dst[c] = v[c];
  }
}

template <int PACK, typename IN, int LD, int SC = 0, int SA = 0, int SF = 0>
__global__ __launch_bounds__(256) void encode_token_kernel(const IN *__restrict__ data, float *__restrict__ out,
                                                           EncGeom g, long total) {
  encode_token_body<PACK, IN, LD, SC, SA, SF>(data, out, g, total, (long)blockIdx.x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename IN>
__global__ __launch_bounds__(256) void encode_wave_kernel(const IN *__restrict__ data, float *__restrict__ out,
                                                          EncGeom g, long total) {
  const int lane = threadIdx.x & 63;
  long tok = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tok >= total) return;
  long n = tok % g.N;
  int idx[HN_MAX_AXES];
  token_coords(n, g, idx);
  const IN *src = data + tok * g.C;
  float *dst = out + tok * (long)g.ld_out;
  const int n_pos = g.D - g.C;
  float mean = 0.0f, rstd = 1.0f;
  if (g.normalize) {
    float s = 0.0f;
    for (int c = lane; c < g.C; c += 64) s += in_at(src, c);
    for (int j = lane; j < n_pos; j += 64) s += pos_feature(j, idx, g);
    mean = wave_sum(s) / (float)g.D;
    float q = 0.0f;
    for (int c = lane; c < g.C; c += 64) { float d = in_at(src, c) - mean; q += d * d; }
    for (int j = lane; j < n_pos; j += 64) { float d = pos_feature(j, idx, g) - mean; q += d * d; }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)g.D + g.eps);
  }
  for (int c = lane; c < g.C; c += 64) dst[c] = g.normalize ? (in_at(src, c) - mean) * rstd : in_at(src, c);
  for (int j = lane; j < n_pos; j += 64) {
    float p = pos_feature(j, idx, g);
    dst[g.C + j] = g.normalize ? (p - mean) * rstd : p;
  }
  for (int c = g.D + lane; c < g.ld_out; c += 64) dst[c] = c == g.ones_col ? 1.0f : 0.0f;
}

// The same with the token's row held in registers (D <= 64 * NV): the row is read from memory ONCE instead of three times and
// every positional feature (an accurate sinf / cosf) is evaluated once instead of three times.  Same summation order as
// encode_wave_kernel (lane-strided partial sums, then the butterfly), so the results are bit-identical.  Patch bags: 32 768
// tokens x 773 channels, 101 MB in and out -- 66 -> ~45 us per call (it runs in the forward AND in the backward's recompute).
template <typename IN, int NV>
__global__ __launch_bounds__(256) void encode_wave_reg_kernel(const IN *__restrict__ data, float *__restrict__ out,
                                                              EncGeom g, long total) {
  const int lane = threadIdx.x & 63;
  long tok = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tok >= total) return;
  long n = tok % g.N;
  int idx[HN_MAX_AXES];
  token_coords(n, g, idx);
  const IN *src = data + tok * g.C;
  float *dst = out + tok * (long)g.ld_out;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < g.C ? in_at(src, c) : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {                        // positional columns: a handful of lanes, behind a wave-uniform test
    const int c = lane + 64 * i;
    if (64 * i + 63 >= g.C && 64 * i < g.D) {
      if (c >= g.C && c < g.D) v[i] = pos_feature(c - g.C, idx, g);
    }
  }
  if (g.normalize) {
    // (the original sums the channels first, then the positional features: keep that order -- a lane holds at most one
    // positional value per register, added after its channel values)
    float s = 0.0f, sp = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < g.C) s += v[i];
      else if (c < g.D) sp += v[i];
    }
    // encode_wave_kernel: lane l accumulates channels l, l + 64, ... and then positional features j = l, l + 64, ... (column
    // C + j): a different lane assignment for the positional part, same set of addends per wave -- the butterfly sum of
    // per-lane partials is order dependent, so mirror it exactly: move each positional value to the lane that owns it there
    float spm = 0.0f;
    {
      const int n_pos = g.D - g.C;
      // value of positional feature j lives in lane (C + j) & 63 of register (C + j) >> 6; the reference lane for it is j & 63
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int base = 64 * i;                          // columns base .. base + 63 sit in register i
        if (base + 63 >= g.C && base < g.D) {
          // lane l wants feature j = l (+ 64 k): its column is C + j -> source lane (C + j) & 63 if that column is in this register
          for (int k = 0; k * 64 < n_pos; ++k) {
            const int j = lane + 64 * k, col = g.C + j;
            const float got = __shfl(v[i], col & 63);
            if (j < n_pos && (col >> 6) == i) spm += got;
          }
        }
      }
      (void)sp;
    }
    const float mean = wave_sum(s + spm) / (float)g.D;
    float q = 0.0f, qp = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < g.C) { const float d = v[i] - mean; q += d * d; }
    }
    {
      const int n_pos = g.D - g.C;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int base = 64 * i;
        if (base + 63 >= g.C && base < g.D) {
          for (int k = 0; k * 64 < n_pos; ++k) {
            const int j = lane + 64 * k, col = g.C + j;
            const float got = __shfl(v[i], col & 63);
            if (j < n_pos && (col >> 6) == i) { const float d = got - mean; qp += d * d; }
          }
        }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q + qp) / (float)g.D + g.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = (v[i] - mean) * rstd;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < g.D) dst[c] = v[i];
    else if (c < g.ld_out) dst[c] = c == g.ones_col ? 1.0f : 0.0f;
  }
  for (int c = 64 * NV + lane; c < g.ld_out; c += 64) dst[c] = c == g.ones_col ? 1.0f : 0.0f;
}

// Few, wide tokens (the tabular / omic modality: b tokens of 2005 channels): one 256-thread workgroup per token instead of
// one wave, so the three passes over the row (sum, variance, write) run 4x wider; workgroup sums through LDS.
__device__ __forceinline__ float block_sum(float v, float *red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename IN>
__device__ __forceinline__ void encode_block_body(const IN *__restrict__ data, float *__restrict__ out, const EncGeom &g, const long tok) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  long n = tok % g.N;
  int idx[HN_MAX_AXES];
  token_coords(n, g, idx);
  const IN *src = data + tok * g.C;
  float *dst = out + tok * (long)g.ld_out;
  const int n_pos = g.D - g.C;
  float mean = 0.0f, rstd = 1.0f;
  if (g.normalize) {
    float s = 0.0f;
    for (int c = tid; c < g.C; c += 256) s += in_at(src, c);
    for (int j = tid; j < n_pos; j += 256) s += pos_feature(j, idx, g);
    mean = block_sum(s, red) / (float)g.D;
    float q = 0.0f;
    for (int c = tid; c < g.C; c += 256) { float d = in_at(src, c) - mean; q += d * d; }
    for (int j = tid; j < n_pos; j += 256) { float d = pos_feature(j, idx, g) - mean; q += d * d; }
    rstd = 1.0f / sqrtf(block_sum(q, red) / (float)g.D + g.eps);
  }
  for (int c = tid; c < g.C; c += 256) dst[c] = g.normalize ? (in_at(src, c) - mean) * rstd : in_at(src, c);
  for (int j = tid; j < n_pos; j += 256) {
    float p = pos_feature(j, idx, g);
    dst[g.C + j] = g.normalize ? (p - mean) * rstd : p;
  }
  for (int c = g.D + tid; c < g.ld_out; c += 256) dst[c] = c == g.ones_col ? 1.0f : 0.0f;
}

template <typename IN>
__global__ __launch_bounds__(256) void encode_block_kernel(const IN *__restrict__ data, float *__restrict__ out, EncGeom g) {
  encode_block_body<IN>(data, out, g, (long)blockIdx.x);
}

// The prelude of the inference forward as ONE launch (round 6): the one-token modality's encode, the folded projections + latent
// broadcast (vfold.h) and the image encode are mutually independent, and until now ran as three launches one behind the other
// (8.6 + 8.0 + 21.4 us at cfg2 b = 32) in front of the skinny products that only need the first.  Roles by workgroup index, the small
// ones first (they are dispatched first and finish under the image encode, which is an HBM stream over all CUs): fp32 inputs, the
// static RGB-image instance of the token encode.
#include "vfold.h"
struct PreludeArgs {
  const float *tab; float *ztab; EncGeom gt; int tab_tokens;
  VfoldMulti v; int vf_gx, vf_gy, vf_gz;
  const float *img; float *zimg; EncGeom gi; long total;
};
template <int PACK>
__global__ __launch_bounds__(256) void prelude_kernel(PreludeArgs a) {
  int bid = (int)blockIdx.x;
  if (bid < a.tab_tokens) { encode_block_body<float>(a.tab, a.ztab, a.gt, (long)bid); return; }
  bid -= a.tab_tokens;
  const int nvf = a.vf_gx * a.vf_gy * a.vf_gz;
  if (bid < nvf) {
    const int bx = bid % a.vf_gx, by = (bid / a.vf_gx) % a.vf_gy, bz = bid / (a.vf_gx * a.vf_gy);
    vfold_body(a.v, bx, by, bz, a.vf_gx, a.vf_gz);
    return;
  }
  bid -= nvf;
  encode_token_body<PACK, float, 16, 3, 2, 2>(a.img, a.zimg, a.gi, a.total, (long)bid);
}

static int fill_geom(EncGeom *g, int b, int n_axes, const int *spatial, int C, int F, float max_freq, int fourier, int normalize,
                     float eps) {
  HN_REQUIRE(spatial, HN_E_NULL, "encode: NULL pointer");
  HN_REQUIRE(b > 0 && C > 0 && n_axes >= 1 && n_axes <= HN_MAX_AXES, HN_E_SHAPE,
             "encode: b=%d C=%d n_axes=%d (1..%d axes supported)", b, C, n_axes, HN_MAX_AXES);
  HN_REQUIRE(!fourier || F >= 1, HN_E_SHAPE, "encode: num_freq_bands=%d", F);
  g->n_axes = n_axes;
  g->N = 1;
  for (int a = 0; a < HN_MAX_AXES; ++a) {
    g->S[a] = a < n_axes ? spatial[a] : 1;
    HN_REQUIRE(g->S[a] > 0, HN_E_SHAPE, "encode: spatial[%d]=%d", a, g->S[a]);
    g->N *= g->S[a];
  }
  g->C = C;
  g->F = F;
  g->D = C + (fourier ? n_axes * (2 * F + 1) : 0);
  g->max_freq = max_freq;
  g->fourier = fourier;
  g->normalize = normalize;
  g->eps = eps;
  g->ones_col = -1;
  g->ld_out = 0;
  g->off0 = 0;
  g->full0 = g->S[0];
  return HN_OK;
}

template <typename IN>
static int launch_encode_t(const IN *data, EncGeom g, int b, float *out, int ld_out, hipStream_t s, int pack_ks) {
  long total = (long)b * g.N;
  if (g.D <= kMaxNarrow && ld_out <= kMaxNarrow) {
    long blocks = ceil_div_ll(total, 256);
    HN_REQUIRE(pack_ks == 0 || (g.normalize && pack_ks == packed_steps(g.D, ld_out)), HN_E_SHAPE, "encode: pack_ks=%d", pack_ks);
#define HN_ENC(P_, LD_) hipLaunchKernelGGL((encode_token_kernel<P_, IN, LD_>), dim3((unsigned)blocks), dim3(256), 0, s, data, out, g, total)
    const bool narrow16 = g.D <= 16 && ld_out <= 16;
    // the two shapes of the BASELINE configs get fully static kernels
#define HN_ENC_S(P_, LD_, C_, A_, F_) hipLaunchKernelGGL((encode_token_kernel<P_, IN, LD_, C_, A_, F_>), dim3((unsigned)blocks), dim3(256), 0, s, data, out, g, total)
    if (g.fourier && g.C == 3 && g.F == 2 && g.n_axes == 2 && ld_out == 16 && (pack_ks == 0 || pack_ks == 3)) {
      if (pack_ks == 3) HN_ENC_S(3, 16, 3, 2, 2); else HN_ENC_S(0, 16, 3, 2, 2);
      HN_LAUNCH_CHECK("encode(static image)");
      return HN_OK;
    }
    if (g.fourier && g.C == 3 && g.F == 2 && g.n_axes == 3 && ld_out == 32 && (pack_ks == 0 || pack_ks == 5)) {
      if (pack_ks == 5) HN_ENC_S(5, 32, 3, 3, 2); else HN_ENC_S(0, 32, 3, 3, 2);
      HN_LAUNCH_CHECK("encode(static volume)");
      return HN_OK;
    }
#undef HN_ENC_S
    switch (pack_ks) {
      case 1: HN_ENC(1, 16); break;
      case 2: HN_ENC(2, 16); break;
      case 3: HN_ENC(3, 16); break;
      case 4: HN_ENC(4, 32); break;      // D = 16 / 17 on a 32-column row
      case 5: HN_ENC(5, 32); break;
      case 6: HN_ENC(6, 32); break;
      case 7: HN_ENC(7, 32); break;
      default: if (narrow16) HN_ENC(0, 16); else HN_ENC(0, 32); break;
    }
#undef HN_ENC
  } else {
    HN_REQUIRE(pack_ks == 0, HN_E_SHAPE, "encode: packed layout needs a narrow modality");
    if (total <= 2048 && g.D >= 512) {      // few wide tokens: a workgroup per token
      hipLaunchKernelGGL((encode_block_kernel<IN>), dim3((unsigned)total), dim3(256), 0, s, data, out, g);
    } else {
      long blocks = ceil_div_ll(total, 4);
      if (g.D <= 64 * 13 && g.D > 64 * 8)
        hipLaunchKernelGGL((encode_wave_reg_kernel<IN, 13>), dim3((unsigned)blocks), dim3(256), 0, s, data, out, g, total);
      else if (g.D <= 64 * 8 && g.D > 64 * 2)
        hipLaunchKernelGGL((encode_wave_reg_kernel<IN, 8>), dim3((unsigned)blocks), dim3(256), 0, s, data, out, g, total);
      else
        hipLaunchKernelGGL((encode_wave_kernel<IN>), dim3((unsigned)blocks), dim3(256), 0, s, data, out, g, total);
    }
  }
  HN_LAUNCH_CHECK("encode");
  return HN_OK;
}

int launch_encode(const void *data, int in_dtype, int b, int n_axes, const int *spatial, int C, int F, float max_freq,
                  int fourier, int normalize, float eps, float *out, int ld_out, hipStream_t s, int ones_col, int pack_ks,
                  int axis0_begin, int axis0_total) {
  HN_REQUIRE(data && out, HN_E_NULL, "encode: NULL pointer");
  HN_REQUIRE(in_dtype == HN_F32 || in_dtype == HN_BF16 || in_dtype == HN_U8, HN_E_UNSUPPORTED,
             "encode: dtype=%d (0 = fp32, 1 = bf16, 2 = uint8)", in_dtype);
  EncGeom g;
  int rc = fill_geom(&g, b, n_axes, spatial, C, F, max_freq, fourier, normalize, eps);
  if (rc != HN_OK) return rc;
  g.ld_out = ld_out;
  g.ones_col = (ones_col >= g.D && ones_col < ld_out) ? ones_col : -1;
  HN_REQUIRE(ld_out >= g.D, HN_E_SHAPE, "encode: ld_out=%d < D=%d", ld_out, g.D);
  if (axis0_total > 0) {
    HN_REQUIRE(axis0_begin >= 0 && axis0_begin + g.S[0] <= axis0_total, HN_E_SHAPE, "encode: axis-0 window [%d, %d) of %d", axis0_begin,
               axis0_begin + g.S[0], axis0_total);
    g.off0 = axis0_begin; g.full0 = axis0_total;
  }
  if (in_dtype == HN_BF16) return launch_encode_t((const uint16_t *)data, g, b, out, ld_out, s, pack_ks);
  if (in_dtype == HN_U8) return launch_encode_t((const uint8_t *)data, g, b, out, ld_out, s, pack_ks);
  return launch_encode_t((const float *)data, g, b, out, ld_out, s, pack_ks);
}

// The prelude as one launch (prelude_kernel above).  `tab` / `img`: the arguments launch_encode would get for the one-token modality
// and for the RGB image (fp32, no axis window); false from encode_prelude_eligible = run the launches separately.
bool encode_prelude_eligible(const EncodeCall &tab, const EncodeCall &img) {
  auto tokens = [](const EncodeCall &c) { long n = 1; for (int a = 0; a < c.n_axes; ++a) n *= c.spatial[a]; return n; };
  const int Dt = tab.C + (tab.fourier ? tab.n_axes * (2 * tab.F + 1) : 0);
  const long total_t = (long)tab.b * tokens(tab);
  return tab.data && tab.out && img.data && img.out && tab.dtype == HN_F32 && img.dtype == HN_F32 && tab.n_axes >= 1 && tab.n_axes <= HN_MAX_AXES &&
         Dt > kMaxNarrow && Dt >= 512 && total_t >= 1 && total_t <= 2048 && tab.pack_ks == 0 && tab.ld_out >= Dt && tab.normalize && img.normalize &&
         img.fourier && img.C == 3 && img.F == 2 && img.n_axes == 2 && img.ld_out == 16 && (img.pack_ks == 0 || img.pack_ks == 3) &&
         (long)img.b * tokens(img) < (1L << 30);
}

int launch_encode_prelude(const EncodeCall &tab, const EncodeCall &img, const VfoldMulti &v, hipStream_t s) {
  HN_REQUIRE(encode_prelude_eligible(tab, img), HN_E_UNSUPPORTED, "encode_prelude: shapes not eligible");
  PreludeArgs a;
  int rc = fill_geom(&a.gt, tab.b, tab.n_axes, tab.spatial, tab.C, tab.F, tab.max_freq, tab.fourier, tab.normalize, tab.eps);
  if (rc != HN_OK) return rc;
  a.gt.ld_out = tab.ld_out;
  a.gt.ones_col = (tab.ones_col >= a.gt.D && tab.ones_col < tab.ld_out) ? tab.ones_col : -1;
  if ((rc = fill_geom(&a.gi, img.b, img.n_axes, img.spatial, img.C, img.F, img.max_freq, img.fourier, img.normalize, img.eps)) != HN_OK) return rc;
  a.gi.ld_out = img.ld_out;
  a.gi.ones_col = (img.ones_col >= a.gi.D && img.ones_col < img.ld_out) ? img.ones_col : -1;
  HN_REQUIRE(img.ld_out >= a.gi.D && (img.pack_ks == 0 || img.pack_ks == packed_steps(a.gi.D, img.ld_out)), HN_E_SHAPE, "encode_prelude: image row D=%d pack=%d", a.gi.D, img.pack_ks);
  a.tab = (const float *)tab.data; a.ztab = tab.out; a.tab_tokens = (int)((long)tab.b * a.gt.N);
  a.img = (const float *)img.data; a.zimg = img.out; a.total = (long)img.b * a.gi.N;
  if ((rc = vfold_plan(v, &a.v, &a.vf_gx, &a.vf_gy, &a.vf_gz)) != HN_OK) return rc;
  const long token_blocks = ceil_div_ll(a.total, 256);
  const long blocks = a.tab_tokens + (long)a.vf_gx * a.vf_gy * a.vf_gz + token_blocks;
  HN_REQUIRE(blocks < (1L << 31), HN_E_UNSUPPORTED, "encode_prelude: grid too large");
  if (img.pack_ks == 3) hipLaunchKernelGGL(prelude_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(prelude_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  HN_LAUNCH_CHECK("prelude");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// bf16 context images for attn_core_bf16_kernel (attention_bf16.hip): per sample
//   zb (Np, ZP)      token-major QK^T operand, ZP = bf16_row_slots(DV, ns); zero rows for the padding tokens n >= N
//                    ns == 1: [zh(32)]   ns == 2, DV == 16: [zh(16) zl(16)] [zh(16) 0]   ns == 2, DV == 32: [zh] [zl] [zh]
//   zT (ns, Np / 32, DV / 16, 4, 16, 8)  fragment-major P V operand planes (hi, lo); channel DV-1 of the hi plane = 1.0 on valid tokens
//                    (softmax denominator), zero on padding tokens and in the lo plane
// zh = bf16(z), zl = bf16(z - zh) of the affine-free LayerNorm z of the encoded token.
// ------------------------------------------------------------------------------------------------
// SC / SA / SF > 0 and SDV: channel count, axes, frequency bands and row width known at compile time (RGB image: 3, 2, 2, 16;
// RGB volume: 3, 3, 2, 32), as in encode_token_kernel: one sincosf per (axis, band), every predicate folds away.
template <typename IN, int NS, int SC = 0, int SA = 0, int SF = 0, int SDV = 0>
__global__ __launch_bounds__(256) void encode_bf16ctx_kernel(const IN *__restrict__ data, uint16_t *__restrict__ zb,
                                                             uint16_t *__restrict__ zT, EncGeom g, int Np, int DV, long total) {
  if (SC > 0) { g.C = SC; g.n_axes = SA; g.F = SF; g.D = SC + SA * (2 * SF + 1); DV = SDV; }
  // (total = b * Np is a multiple of 32: the last workgroup may be partly idle, but every 32-token block is whole)
  const long gid_raw = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = gid_raw < total;
  const long gid = active ? gid_raw : total - 1;
  const long bi = gid / Np;
  const int n = (int)(gid % Np);
  float v[kMaxNarrow];
#pragma unroll
  for (int c = 0; c < kMaxNarrow; ++c) v[c] = 0.0f;
  const bool valid = n < g.N;
  if (valid) {
    int idx[HN_MAX_AXES];
    token_coords(n, g, idx);
    const IN *src = data + (bi * g.N + n) * g.C;
    if (SC > 0) {
#pragma unroll
      for (int c = 0; c < SC; ++c) v[c] = in_at(src, c);
#pragma unroll
      for (int a = 0; a < SA; ++a) {
        const float p = axis_pos_of(a, idx, g);
#pragma unroll
        for (int f = 0; f < SF; ++f) {
          const float arg = __fmul_rn(__fmul_rn(p, band_scale(f, SF, g.max_freq)), 3.14159265358979323846f);  // (p*s)*pi, :299
          float sn, cs;
          sincosf(arg, &sn, &cs);
          v[SC + a * (2 * SF + 1) + f] = sn;
          v[SC + a * (2 * SF + 1) + SF + f] = cs;
        }
        v[SC + a * (2 * SF + 1) + 2 * SF] = p;
      }
    } else {
#pragma unroll
      for (int c = 0; c < kMaxNarrow; ++c) {
        float x = 0.0f;
        if (c < g.C) x = in_at(src, c);
        else if (c < g.D) x = pos_feature(c - g.C, idx, g);
        v[c] = x;
      }
    }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < kMaxNarrow; ++c) sum += (c < g.D) ? v[c] : 0.0f;
    const float mean = sum / (float)g.D;
    float sq = 0.0f;
#pragma unroll
    for (int c = 0; c < kMaxNarrow; ++c) {
      float d = (c < g.D) ? v[c] - mean : 0.0f;
      sq += d * d;
    }
    const float rstd = 1.0f / sqrtf(sq / (float)g.D + g.eps);
#pragma unroll
    for (int c = 0; c < kMaxNarrow; ++c) v[c] = (c < g.D) ? (v[c] - mean) * rstd : 0.0f;
  }
  uint16_t hi[kMaxNarrow], lo[kMaxNarrow];
#pragma unroll
  for (int c = 0; c < kMaxNarrow; ++c) {
    hi[c] = to_bf16(v[c]);
    lo[c] = NS == 2 ? to_bf16(v[c] - __uint_as_float((unsigned)hi[c] << 16)) : (uint16_t)0;
  }
  // token-major row: slot s of the row image
  const int zp = bf16_row_slots(DV, NS);
  uint4 *dst = (uint4 *)(zb + gid * zp);
  auto slot = [&](int s) -> unsigned {         // s is a compile-time constant after unrolling
    if (NS == 1) return hi[s];
    if (DV == 16) return s < 16 ? hi[s] : (s < 32 ? lo[s - 16] : (s < 48 ? hi[s - 32] : 0));
    return s < 32 ? hi[s] : (s < 64 ? lo[s - 32] : hi[s - 64]);
  };
  if (active) {
    if (NS == 1 || DV == 16) {
#pragma unroll
      for (int q = 0; q < (NS == 1 ? 4 : 8); ++q)
        dst[q] = make_uint4(slot(8 * q) | (slot(8 * q + 1) << 16), slot(8 * q + 2) | (slot(8 * q + 3) << 16),
                            slot(8 * q + 4) | (slot(8 * q + 5) << 16), slot(8 * q + 6) | (slot(8 * q + 7) << 16));
    } else {
#pragma unroll
      for (int q = 0; q < 12; ++q)
        dst[q] = make_uint4(slot(8 * q) | (slot(8 * q + 1) << 16), slot(8 * q + 2) | (slot(8 * q + 3) << 16),
                            slot(8 * q + 4) | (slot(8 * q + 5) << 16), slot(8 * q + 6) | (slot(8 * q + 7) << 16));
    }
  }
  // P V operand planes in FRAGMENT-MAJOR tiles: per 32-token block and 16-channel group one contiguous 1 KB tile laid out as the
  // MFMA B fragment reads it -- [g = token / 8][j = channel % 16][8 tokens] -- so that a wave's operand load is eight full
  // 128-byte lines.  (Channel-major rows (DV, Np) made every load touch 16 lines 1.2 MB apart, half of each used: the slow
  // "16 rows x 64 B" pattern of DESIGN 4.7.)  The tiles are written in 16-byte pieces (8 consecutive tokens of one channel)
  // after a transpose through LDS: per-token 2-byte stores made this kernel 267 us at cfg3 (9.6 M tokens x 32 channels).
  __shared__ uint16_t tr[NS][256][kMaxNarrow + 2];
  {
#pragma unroll
    for (int c = 0; c < kMaxNarrow; ++c)
      if (c < DV) {
        tr[0][threadIdx.x][c] = c == DV - 1 ? (valid ? (uint16_t)0x3f80 : (uint16_t)0) : hi[c];
        if (NS == 2) tr[NS - 1][threadIdx.x][c] = c == DV - 1 ? (uint16_t)0 : lo[c];
      }
  }
  __syncthreads();
  const int DTV = DV >> 4;
  const int pieces = 8 * DTV * 64;             // 8 token blocks of 32 per workgroup x DTV channel groups x 64 fragment slots
  for (int p = 0; p < NS; ++p)
    for (int q = threadIdx.x; q < pieces; q += 256) {
      const int tb = q / (DTV * 64), rem = q - tb * (DTV * 64), dg = rem >> 6, sl = rem & 63, fg = sl >> 4, fj = sl & 15;
      const long gid0 = (long)blockIdx.x * 256 + tb * 32;
      if (gid0 >= total) continue;
      const long sb = gid0 / Np;
      const int blk = (int)(gid0 - sb * Np) >> 5, c = dg * 16 + fj, tl = tb * 32 + fg * 8;
      unsigned w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (unsigned)tr[p][tl + 2 * e][c] | ((unsigned)tr[p][tl + 2 * e + 1][c] << 16);
      uint16_t *plane = zT + (sb * NS + p) * (long)DV * Np;
      *(uint4 *)(plane + ((((long)blk * DTV + dg) * 4 + fg) * 16 + fj) * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

int launch_encode_bf16ctx(const void *data, int in_dtype, int b, int n_axes, const int *spatial, int C, int F, float max_freq,
                          int fourier, float eps, uint16_t *zb, uint16_t *zT, int Np, int DV, int ns, hipStream_t s) {
  HN_REQUIRE(data && zb && zT, HN_E_NULL, "encode_bf16ctx: NULL pointer");
  HN_REQUIRE(in_dtype == HN_F32 || in_dtype == HN_BF16 || in_dtype == HN_U8, HN_E_UNSUPPORTED,
             "encode: dtype=%d (0 = fp32, 1 = bf16, 2 = uint8)", in_dtype);
  HN_REQUIRE(ns == 1 || ns == 2, HN_E_UNSUPPORTED, "encode_bf16ctx: ns=%d", ns);
  EncGeom g;
  int rc = fill_geom(&g, b, n_axes, spatial, C, F, max_freq, fourier, 1, eps);
  if (rc != HN_OK) return rc;
  HN_REQUIRE(g.D <= DV - 1 && (DV == 16 || DV == 32) && Np % 32 == 0 && Np >= g.N, HN_E_SHAPE,
             "encode_bf16ctx: D=%d DV=%d Np=%d N=%ld", g.D, DV, Np, g.N);
  const long total = (long)b * Np;
  const dim3 grid((unsigned)ceil_div_ll(total, 256)), block(256);
#define HN_ENC16(T_, NS_) hipLaunchKernelGGL((encode_bf16ctx_kernel<T_, NS_>), grid, block, 0, s, (const T_ *)data, zb, zT, g, Np, DV, total)
#define HN_ENC16_S(T_, A_, DV_) hipLaunchKernelGGL((encode_bf16ctx_kernel<T_, 1, 3, A_, 2, DV_>), grid, block, 0, s, (const T_ *)data, zb, zT, g, Np, DV, total)
  // static variants for the two BASELINE shapes (plain bf16 plane): RGB image (2 axes, 2 bands, 16-wide row), RGB volume (3 axes, 32-wide)
  const bool rgb = g.fourier && g.C == 3 && g.F == 2 && ns == 1;
  if (rgb && g.n_axes == 2 && DV == 16 && (in_dtype == HN_BF16 || in_dtype == HN_F32)) {
    if (in_dtype == HN_BF16) HN_ENC16_S(uint16_t, 2, 16); else HN_ENC16_S(float, 2, 16);
  } else if (rgb && g.n_axes == 3 && DV == 32 && (in_dtype == HN_BF16 || in_dtype == HN_F32)) {
    if (in_dtype == HN_BF16) HN_ENC16_S(uint16_t, 3, 32); else HN_ENC16_S(float, 3, 32);
  } else
  if (in_dtype == HN_BF16) { if (ns == 1) HN_ENC16(uint16_t, 1); else HN_ENC16(uint16_t, 2); }
  else if (in_dtype == HN_U8) { if (ns == 1) HN_ENC16(uint8_t, 1); else HN_ENC16(uint8_t, 2); }
  else { if (ns == 1) HN_ENC16(float, 1); else HN_ENC16(float, 2); }
#undef HN_ENC16
#undef HN_ENC16_S
  HN_LAUNCH_CHECK("encode_bf16ctx");
  return HN_OK;
}

}  // namespace hn
