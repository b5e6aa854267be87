// Small support kernels: latent broadcast (healnet.py:225), classifier head (to_logits :181-185).
#include "common.h"
#include <stdlib.h>

namespace hn {

// (also clears `nzero` ints at `zero`: the score-bound fallback flags of a forward ride on this launch)
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                             long n_per, long total, int *__restrict__ zero, int nzero) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    dst[i] = src[i % n_per];
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < nzero; i += blockDim.x) zero[i] = 0;
}

int launch_broadcast_rows(const float *src, float *dst, long n_per, int b, hipStream_t s, int *zero, int nzero) {
  long total = n_per * b;
  long blocks = ceil_div_ll(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n_per, total, zero, zero ? nzero : 0);
  HN_LAUNCH_CHECK("broadcast_rows");
  return HN_OK;
}

__global__ __launch_bounds__(256) void pad_rows_kernel(const float *__restrict__ src, int ld_src, float *__restrict__ dst,
                                                       int ld_dst, long rows, int cols) {
  long total = rows * ld_dst;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / ld_dst;
    int c = (int)(i % ld_dst);
    dst[i] = c < cols ? src[r * ld_src + c] : 0.0f;
  }
}

int launch_pad_rows(const float *src, int ld_src, float *dst, int ld_dst, long rows, int cols, hipStream_t s) {
  long blocks = ceil_div_ll(rows * ld_dst, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, ld_src, dst, ld_dst, rows, cols);
  HN_LAUNCH_CHECK("pad_rows");
  return HN_OK;
}

// blockIdx.y = piece; the piece's destination rectangle in 256-thread strides (common.h: StageTable)
__global__ __launch_bounds__(256) void stage_kernel(const StageTable t) {
  const StagePiece &p = t.p[blockIdx.y];
  const long total = (long)p.rows_dst * p.cols_dst;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / p.cols_dst), c = (int)(i - (long)r * p.cols_dst);
    const float v = (r < p.rows_src && c < p.cols_src) ? p.src[(long)r * p.ld_src + c] : 0.0f;
    float *d = p.dst + (long)r * p.ld_dst + c;
    *d = t.accumulate ? *d + v : v;
  }
}

int launch_stage(const StageTable &t, hipStream_t s) {
  HN_REQUIRE(t.n >= 0 && t.n <= STAGE_MAX, HN_E_SHAPE, "stage: %d pieces", t.n);
  if (t.n == 0) return HN_OK;
  long biggest = 1;
  for (int i = 0; i < t.n; ++i) {
    const long e = (long)t.p[i].rows_dst * t.p[i].cols_dst;
    if (e > biggest) biggest = e;
  }
  long blocks = ceil_div_ll(biggest, 1024);
  if (blocks > 64) blocks = 64;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(stage_kernel, dim3((unsigned)blocks, t.n), dim3(256), 0, s, t);
  HN_LAUNCH_CHECK("stage");
  return HN_OK;
}

// x_out[b, l, :] = y[b, :] (+ x_in[b, l, :]) -- epilogue of the one-token attention path
__global__ __launch_bounds__(256) void add_row_broadcast_kernel(const float *__restrict__ y, const float *x_in,
                                                                float *x_out, int L, int d, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / d;
    const int c = (int)(i - row * d);
    const float v = y[(row / L) * d + c];
    x_out[i] = x_in ? v + x_in[i] : v;
  }
}

int launch_add_row_broadcast(const float *y, const float *x_in, float *x_out, int b, int L, int d, hipStream_t s) {
  const long total = (long)b * L * d;
  long blocks = ceil_div_ll(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_row_broadcast_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y, x_in, x_out, L, d, total);
  HN_LAUNCH_CHECK("add_row_broadcast");
  return HN_OK;
}

__global__ __launch_bounds__(256) void fill_kernel(float *__restrict__ dst, float value, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = value;
}

int launch_fill(float *dst, float value, long n, hipStream_t s) {
  long blocks = ceil_div_ll(n, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, value, n);
  HN_LAUNCH_CHECK("fill");
  return HN_OK;
}

// Device-to-device copy as an ordinary kernel.  The library never uses hipMemcpyAsync / hipMemsetAsync between its kernels:
// on this stack (ROCm 7.2, MI355X) results that depended on a D2D hipMemcpyAsync issued between two kernels of the same
// stream were wrong in a few percent of fresh processes (tools/flake_probe.py); plain kernels are ordered like any other.
__global__ __launch_bounds__(256) void copy_kernel(const float *__restrict__ src, float *__restrict__ dst, long n) {
  const long n4 = n >> 2;
  const bool al = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
  if (al) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
      ((float4 *)dst)[i] = ((const float4 *)src)[i];
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
  }
}

int launch_copy(float *dst, const float *src, long n, hipStream_t s) {
  if (n <= 0 || dst == src) return HN_OK;
  long blocks = ceil_div_ll((n + 3) / 4, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n);
  HN_LAUNCH_CHECK("copy");
  return HN_OK;
}

__global__ __launch_bounds__(256) void fill_bytes_kernel(uint8_t *__restrict__ dst, uint8_t value, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = value;
}

int launch_fill_bytes(uint8_t *dst, uint8_t value, long n, hipStream_t s) {
  long blocks = ceil_div_ll(n, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, value, n);
  HN_LAUNCH_CHECK("fill_bytes");
  return HN_OK;
}

// One workgroup per sample: column means over the L latent rows -> LayerNorm(d) -> Linear(d, out).
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ x, int L, int d, const float *__restrict__ nw,
                                                   const float *__restrict__ nb, const float *__restrict__ w,
                                                   const float *__restrict__ bias, int out_dims, float *__restrict__ logits, int dv) {
  // dv: LayerNorm statistics over the first dv columns (staged models: the pad columns of x, gamma, beta, w are zero)
  extern __shared__ float sm[];  // pooled[d] + 8 scratch + 4 x d partial column sums
  float *pooled = sm;
  float *red = sm + d;
  float *part = sm + d + 8;
  const int bi = blockIdx.x, tid = threadIdx.x;
  const float *xb = x + (long)bi * L * d;
  if ((d & 3) == 0 && d <= 256) {
    // column means, 16-byte loads: d/4 column quads x (256 / (d/4)) row groups, four independent accumulations in flight
    const int nq = d >> 2, groups = 256 / nq;
    const int cq = tid % nq, gr = tid / nq;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (gr < groups) {
      int r = gr;
      for (; r + 3 * groups < L; r += 4 * groups) {          // four rows in flight per trip (L = 128, 8 groups: 4 trips instead of 8)
        const float4 v0 = *(const float4 *)(xb + (long)r * d + 4 * cq), v1 = *(const float4 *)(xb + (long)(r + groups) * d + 4 * cq);
        const float4 v2 = *(const float4 *)(xb + (long)(r + 2 * groups) * d + 4 * cq), v3 = *(const float4 *)(xb + (long)(r + 3 * groups) * d + 4 * cq);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        a0.x += v2.x; a0.y += v2.y; a0.z += v2.z; a0.w += v2.w;
        a1.x += v3.x; a1.y += v3.y; a1.z += v3.z; a1.w += v3.w;
      }
      for (; r + groups < L; r += 2 * groups) {
        const float4 v0 = *(const float4 *)(xb + (long)r * d + 4 * cq), v1 = *(const float4 *)(xb + (long)(r + groups) * d + 4 * cq);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      }
      if (r < L) { const float4 v0 = *(const float4 *)(xb + (long)r * d + 4 * cq); a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w; }
      float *dst = part + (long)gr * d + 4 * cq;        // groups * d <= 1024 floats (launch_head sizes the LDS for it)
      dst[0] = a0.x + a1.x; dst[1] = a0.y + a1.y; dst[2] = a0.z + a1.z; dst[3] = a0.w + a1.w;
    }
    __syncthreads();
    for (int c = tid; c < d; c += blockDim.x) {
      float sacc = 0.0f;
      for (int k = 0; k < groups; ++k) sacc += part[(long)k * d + c];
      pooled[c] = sacc / (float)L;
    }
  } else {   // column means: the four waves take rows w, w+4, ...; lanes run along the columns (coalesced rows)
    const int w = tid >> 6, ln = tid & 63;
    for (int c = ln; c < d; c += 64) {
      float s = 0.0f;
      for (int r = w; r < L; r += 4) s += xb[(long)r * d + c];
      part[w * d + c] = s;
    }
    __syncthreads();
    for (int c = tid; c < d; c += blockDim.x) pooled[c] = (part[c] + part[d + c] + part[2 * d + c] + part[3 * d + c]) / (float)L;
  }
  __syncthreads();
  // mean
  float s = 0.0f;
  for (int c = tid; c < d; c += blockDim.x) s += pooled[c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)dv;
  __syncthreads();
  float q = 0.0f;
  for (int c = tid; c < dv; c += blockDim.x) { float t = pooled[c] - mean; q += t * t; }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if ((tid & 63) == 0) red[tid >> 6] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)dv + 1e-5f);
  __syncthreads();
  for (int c = tid; c < d; c += blockDim.x) pooled[c] = c < dv ? (pooled[c] - mean) * rstd * nw[c] + nb[c] : 0.0f;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int o = wave; o < out_dims; o += 4) {
    float acc = 0.0f;
    for (int c = lane; c < d; c += 64) acc = fmaf(pooled[c], w[(long)o * d + c], acc);
    for (int k = 32; k > 0; k >>= 1) acc += __shfl_xor(acc, k);
    if (lane == 0) logits[(long)bi * out_dims + o] = acc + (bias ? bias[o] : 0.0f);
  }
}

int launch_head(const float *x, int b, int L, int d, const float *nw, const float *nb, const float *w, const float *bias,
                int out_dims, float *logits, hipStream_t s, int dv) {
  HN_REQUIRE(x && nw && nb && w && logits, HN_E_NULL, "head: NULL pointer");
  HN_REQUIRE(b > 0 && L > 0 && d > 0 && out_dims > 0 && dv >= 0 && dv <= d, HN_E_SHAPE, "head: b=%d L=%d d=%d out=%d valid=%d", b, L, d, out_dims, dv);
  if (dv == 0) dv = d;
  size_t lds = (size_t)(d + 8 + (4 * d > 1024 ? 4 * d : 1024)) * sizeof(float);     // pooled + scratch + row-group partials
  HN_REQUIRE(lds <= 64 * 1024, HN_E_UNSUPPORTED, "head: l_d=%d too large", d);
  hipLaunchKernelGGL(head_kernel, dim3(b), dim3(256), lds, s, x, L, d, nw, nb, w, bias, out_dims, logits, dv);
  HN_LAUNCH_CHECK("head");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// dropout helpers (see DropCfg in common.h)
// ------------------------------------------------------------------------------------------------
// out[r, c] = (add ? add[r, c] : 0) + src[r, c] * keepscale(r, c).  One thread per aligned column quad (the unit of one Philox
// call); VEC: cols % 4 == 0, 16-byte accesses.  Otherwise (the reference's tuned widths: l_d = 119, 126, 62, 65) rows are not
// 16-byte aligned and the last quad is partial: scalar accesses, same masks.
template <bool VEC>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float *__restrict__ src, const float *add, float *out, long rows,
                                                            int cols, DropCfg d) {
  const int q4 = (cols + 3) >> 2;
  const long total = rows * q4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / q4;
    const int q = (int)(i - r * q4);
    float m[4];
    drop_quad(d, (uint32_t)q, (uint32_t)r, m);
    if (VEC) {
      const float4 v = ((const float4 *)src)[i];
      float4 o = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
      if (add) { const float4 a = ((const float4 *)add)[i]; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
      ((float4 *)out)[i] = o;
    } else {
      const long base = r * cols + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * q + e < cols) out[base + e] = (add ? add[base + e] : 0.0f) + src[base + e] * m[e];
    }
  }
}

int launch_dropout_apply(const float *src, const float *add, float *out, long rows, int cols, const DropCfg &d, hipStream_t s) {
  HN_REQUIRE(cols > 0 && rows > 0, HN_E_SHAPE, "dropout: rows=%ld cols=%d", rows, cols);
  long blocks = ceil_div_ll(rows * ((cols + 3) >> 2), 256);
  if (blocks > 8192) blocks = 8192;
  const bool vec = (cols & 3) == 0 && (((uintptr_t)src | (uintptr_t)out | (uintptr_t)add) & 15) == 0;
  if (vec) hipLaunchKernelGGL(dropout_apply_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, src, add, out, rows, cols, d);
  else hipLaunchKernelGGL(dropout_apply_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, src, add, out, rows, cols, d);
  HN_LAUNCH_CHECK("dropout_apply");
  return HN_OK;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t *__restrict__ mask, long rows, int cols, DropCfg d) {
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const int c = (int)(i - r * cols);
    mask[i] = drop_one(d, (uint32_t)c, (uint32_t)r) != 0.0f ? 1 : 0;
  }
}

int launch_dropout_mask(uint8_t *mask, long rows, int cols, const DropCfg &d, hipStream_t s) {
  long blocks = ceil_div_ll(rows * cols, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, s, mask, rows, cols, d);
  HN_LAUNCH_CHECK("dropout_mask");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// Stand-alone pieces of the reference's module surface (inside the fused path they live in K1 / the GEMM epilogues):
//   fourier_encode(x, max_freq, num_bands)  healnet.py:292-302   out[i, :] = [sin(x s_f pi).., cos(x s_f pi).., x]
//   GELU / SELU gate modules                healnet.py:323-331   out = a * act(g), (a | g) = chunk(x, 2, dim=-1)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fourier_encode_kernel(const float *__restrict__ x, float *__restrict__ out, long n, int F,
                                                             float max_freq) {
  const int per = 2 * F + 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n * per; i += (long)gridDim.x * blockDim.x) {
    const long e = i / per;
    const int w = (int)(i - e * per);
    const float p = x[e];
    float v = p;
    if (w < 2 * F) {
      const int f = w < F ? w : w - F;
      const float end = max_freq * 0.5f;                   // torch.linspace(1, max_freq / 2, F)[f], ATen's symmetric form
      float sc = 1.0f;
      if (F > 1) {
        const float step = __fdiv_rn(end - 1.0f, (float)(F - 1));
        sc = f < F / 2 ? __fadd_rn(1.0f, __fmul_rn(step, (float)f)) : __fsub_rn(end, __fmul_rn(step, (float)(F - 1 - f)));
      }
      const float arg = __fmul_rn(__fmul_rn(p, sc), 3.14159265358979323846f);
      v = w < F ? sinf(arg) : cosf(arg);
    }
    out[i] = v;
  }
}

int launch_fourier_encode(const float *x, float *out, long n, int F, float max_freq, hipStream_t s) {
  HN_REQUIRE(x && out && n > 0 && F >= 1, HN_E_SHAPE, "fourier_encode: n=%ld num_bands=%d", n, F);
  long blocks = ceil_div_ll(n * (2 * F + 1), 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fourier_encode_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, out, n, F, max_freq);
  HN_LAUNCH_CHECK("fourier_encode");
  return HN_OK;
}

__global__ __launch_bounds__(256) void glu_gate_kernel(const float *__restrict__ x, float *__restrict__ out, long rows, int hid, int gelu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * hid; i += (long)gridDim.x * blockDim.x) {
    const long r = i / hid;
    const int c = (int)(i - r * hid);
    const float a = x[r * 2 * hid + c], g = x[r * 2 * hid + hid + c];
    float act;
    if (gelu) act = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
    else act = 1.0507009873554804934193349852946f * (g > 0.0f ? g : 1.6732632423543772848170429916717f * expm1f(g));
    out[i] = a * act;
  }
}

int launch_glu_gate(const float *x, float *out, long rows, int hid, int gelu, hipStream_t s) {
  HN_REQUIRE(x && out && rows > 0 && hid > 0, HN_E_SHAPE, "glu_gate: rows=%ld hid=%d", rows, hid);
  long blocks = ceil_div_ll(rows * hid, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(glu_gate_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, out, rows, hid, gelu);
  HN_LAUNCH_CHECK("glu_gate");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// temperature_softmax(logits, temperature, dim = -1) (healnet/models/healnet.py:354-365): F.softmax(logits / T) over the
// contiguous last dimension, one wave per row (rows of any length; three passes over the row: max, sum, write).
// Inside Attention.forward the same function is fused into the split-KV core; this entry point is the stand-alone op.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void temperature_softmax_kernel(const float *__restrict__ x, float *__restrict__ y, long rows,
                                                                  int n, float inv_t) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *xr = x + row * n;
  float *yr = y + row * n;
  float m = -__builtin_inff();
  for (int i = lane; i < n; i += 64) m = fmaxf(m, xr[i] * inv_t);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.0f;
  for (int i = lane; i < n; i += 64) s += expf(xr[i] * inv_t - m);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.0f / s;
  for (int i = lane; i < n; i += 64) yr[i] = expf(xr[i] * inv_t - m) * inv;
}

int launch_temperature_softmax(const float *x, float *y, long rows, int n, float temperature, hipStream_t s) {
  HN_REQUIRE(x && y, HN_E_NULL, "temperature_softmax: NULL pointer");
  HN_REQUIRE(rows > 0 && n > 0 && temperature > 0.0f, HN_E_SHAPE, "temperature_softmax: rows=%ld n=%d T=%g", rows, n, (double)temperature);
  hipLaunchKernelGGL(temperature_softmax_kernel, dim3((unsigned)ceil_div_ll(rows, 4)), dim3(256), 0, s, x, y, rows, n, 1.0f / temperature);
  HN_LAUNCH_CHECK("temperature_softmax");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lds_poison_kernel(float *sink) {
  __shared__ float buf[16000];                       // 62.5 KB: two such workgroups per CU touch 125 of the 160 KB
  for (int i = threadIdx.x; i < 16000; i += 256) buf[i] = __uint_as_float(0x7fc00000u + i);
  __syncthreads();
  if (sink && buf[threadIdx.x] == 1.0f) sink[0] = 1.0f;   // keep the stores alive
}

void debug_after_launch(hipStream_t s) {
  static int enabled = -1;
  if (enabled < 0) { const char *e = tuning_env("HN_POISON_LDS"); enabled = (e && e[0] == '1') ? 1 : 0; }
  if (enabled) hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), 0, s, (float *)nullptr);
}

}  // namespace hn
