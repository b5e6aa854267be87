// Training-step tail of the fusion path: what the reference's loop runs right after
// `logits = model.forward(features)` (healnet/main.py:432-467):
//
//   surv_nll_kernel   hazards = sigmoid(logits), survival = cumprod(1 - hazards), risk = -sum(survival)  (main.py:439-441)
//                     + nll_loss(hazards, S, Y, c, weights, alpha = 0.4, eps = 1e-7)   (healnet/models/survival_loss.py:9-43)
//                     + its gradient w.r.t. the logits in closed form, one thread per sample, fixed-order batch mean
//   l1_adam_kernel    g = grad_scale * grad + l1 * sign(p)   (calc_reg_loss, healnet/utils/train_utils.py:5-14: the L1
//                     norm over ALL parameters enters the loss, its autograd gradient is l1 * sign(p))
//                     + torch.optim.Adam's update (main.py:390; betas / lr are per-step arguments because OneCycleLR
//                     cycles both) + sum |p| for the logged reg_loss: ONE pass over the flat parameter / gradient /
//                     moment buffers instead of abs().sum() + sign() + the optimizer's foreach passes (each a full
//                     HBM sweep over 9.6 M parameters).  HBM-bound: 16 B read + 12 B written per parameter.
#include "common.h"

namespace hn {

constexpr int kMaxBins = 64;

__global__ __launch_bounds__(256) void surv_nll_kernel(const float *__restrict__ logits, const long long *__restrict__ y,
                                                       const float *__restrict__ cens, const float *__restrict__ weights,
                                                       int b, int K, float alpha, float eps, float grad_scale,
                                                       float *__restrict__ loss, float *__restrict__ dlogits,
                                                       float *__restrict__ hazards, float *__restrict__ survival,
                                                       float *__restrict__ risk) {
  __shared__ float red[256];
  __shared__ float wsum_s;
  if (threadIdx.x == 0) {
    float ws = 0.0f;
    if (weights) for (int k = 0; k < K; ++k) ws += weights[k];      // weights / sum(weights), survival_loss.py:35
    wsum_s = ws;
  }
  __syncthreads();
  float acc = 0.0f;
  for (int i = threadIdx.x; i < b; i += blockDim.x) {
    const float *l = logits + (long)i * K;
    const long long yraw = y[i];
    // a label outside [0, K) makes the reference's torch.gather raise (survival_loss.py:27-31); a kernel cannot raise, so the
    // sample poisons the loss and its gradient row with NaN instead of reading weights[] / S_pad[] out of bounds
    const bool bad_label = yraw < 0 || yraw >= K;
    const int yi = bad_label ? 0 : (int)yraw;
    const float c = cens[i];
    float S = 1.0f, A = 1.0f, B = 1.0f, H = 0.0f, rsum = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float h = 1.0f / (1.0f + expf(-l[k]));
      if (k == yi) { A = S; H = h; }
      S *= (1.0f - h);
      if (k == yi) B = S;
      rsum += S;
      if (hazards) hazards[(long)i * K + k] = h;
      if (survival) survival[(long)i * K + k] = S;
    }
    if (risk) risk[i] = -rsum;
    const float w = weights ? weights[yi] / wsum_s : 1.0f;
    const float unc = -(1.0f - c) * (logf(fmaxf(A, eps)) + logf(fmaxf(H, eps)));
    const float cen = -c * logf(fmaxf(B, eps));
    const float neg = (cen + unc) * w;
    acc += bad_label ? __int_as_float(0x7fc00000) : (1.0f - alpha) * neg + alpha * unc;
    if (dlogits) {
      const float a_c = (1.0f - alpha) * w * c * (B >= eps ? 1.0f : 0.0f);
      const float a_u = ((1.0f - alpha) * w + alpha) * (1.0f - c);
      const float gA = A >= eps ? 1.0f : 0.0f, gH = H >= eps ? 1.0f : 0.0f;
      const float sc = grad_scale / (float)b;
      for (int k = 0; k < K; ++k) {
        const float h = 1.0f / (1.0f + expf(-l[k]));
        float g = 0.0f;
        if (k <= yi) g += a_c * h;                  // d(-log S_pad[y+1]) / dl_k = h_k, k <= y
        if (k < yi) g += a_u * gA * h;              // d(-log S_pad[y])   / dl_k = h_k, k <  y
        if (k == yi) g -= a_u * gH * (1.0f - h);    // d(-log h_y)        / dl_y = -(1 - h_y)
        dlogits[(long)i * K + k] = bad_label ? __int_as_float(0x7fc00000) : g * sc;
      }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss) loss[0] = red[0] / (float)b;
}

int launch_surv_nll(const float *logits, const long long *y, const float *cens, const float *weights, int b, int K, float alpha,
                    float eps, float grad_scale, float *loss, float *dlogits, float *hazards, float *survival, float *risk,
                    hipStream_t s) {
  HN_REQUIRE(logits && y && cens, HN_E_NULL, "surv_nll: NULL pointer");
  HN_REQUIRE(b >= 1 && K >= 1 && K <= kMaxBins, HN_E_SHAPE, "surv_nll: b=%d bins=%d (1..%d)", b, K, kMaxBins);
  hipLaunchKernelGGL(surv_nll_kernel, dim3(1), dim3(256), 0, s, logits, y, cens, weights, b, K, alpha, eps, grad_scale, loss,
                     dlogits, hazards, survival, risk);
  HN_LAUNCH_CHECK("surv_nll");
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
constexpr int ADAM_BLOCKS = 1024;

struct AdamArgs {
  float l1, grad_scale, beta1, beta2, one_minus_beta1, one_minus_beta2, step_size, bc2_sqrt, eps;
};

__global__ __launch_bounds__(256) void l1_adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                      float *__restrict__ v, long n, AdamArgs a, float *__restrict__ partial,
                                                      const unsigned *__restrict__ status_word) {
  float asum = 0.0f;
  const long n4 = n >> 2;
  // The device's cluster status word (common.h): non-zero = a latent chain of this step lost an exchange and NaN has been on its
  // way into these gradients since.  The update is then SKIPPED (only l1 * sum |p| is still produced), like a loss-scaler skips an
  // overflowed step: the host learns of it at its next entry-point call, after this kernel was enqueued.  `status_word` is the
  // step's SNAPSHOT of the host-mapped word in device memory (status_snapshot_kernel, one PCIe read): reading the mapped word
  // here, from every wave, cost 750 us per step (r05b: ~90 ns per read, serialised).
  if (status_word != nullptr && *status_word != 0u) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
      const float4 pp = ((const float4 *)p)[i];
      asum += (fabsf(pp.x) + fabsf(pp.y)) + (fabsf(pp.z) + fabsf(pp.w));
    }
    if (blockIdx.x == 0)
      for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) asum += fabsf(p[i]);
    n = 0;                                    // (both update loops below see an empty range)
  }
  const long n4u = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4u; i += (long)gridDim.x * blockDim.x) {
    float4 pp = ((float4 *)p)[i], gg = ((const float4 *)g)[i], mm = ((float4 *)m)[i], vv = ((float4 *)v)[i];
    float *pa = (float *)&pp, *ga = (float *)&gg, *ma = (float *)&mm, *va = (float *)&vv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float x = pa[k];
      asum += fabsf(x);
      const float sgn = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
      const float gt = ga[k] * a.grad_scale + a.l1 * sgn;
      const float mn = ma[k] + a.one_minus_beta1 * (gt - ma[k]);                 // exp_avg.lerp_(grad, 1 - beta1)
      const float vn = va[k] * a.beta2 + (a.one_minus_beta2 * gt) * gt;          // mul_(beta2).addcmul_(g, g, 1 - beta2)
      const float denom = sqrtf(vn) / a.bc2_sqrt + a.eps;
      pa[k] = x + (-a.step_size * mn) / denom;                                   // addcdiv_(exp_avg, denom, value=-step_size)
      ma[k] = mn;
      va[k] = vn;
    }
    ((float4 *)p)[i] = pp;
    ((float4 *)m)[i] = mm;
    ((float4 *)v)[i] = vv;
  }
  if (blockIdx.x == 0) {
    for (long i = (n4u << 2) + threadIdx.x; i < n; i += blockDim.x) {
      const float x = p[i];
      asum += fabsf(x);
      const float sgn = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
      const float gt = g[i] * a.grad_scale + a.l1 * sgn;
      const float mn = m[i] + a.one_minus_beta1 * (gt - m[i]);
      const float vn = v[i] * a.beta2 + (a.one_minus_beta2 * gt) * gt;
      const float denom = sqrtf(vn) / a.bc2_sqrt + a.eps;
      p[i] = x + (-a.step_size * mn) / denom;
      m[i] = mn;
      v[i] = vn;
    }
  }
  __shared__ float red[256];
  red[threadIdx.x] = asum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void status_snapshot_kernel(const unsigned *__restrict__ host_word, unsigned *__restrict__ snap) {
  snap[0] = __hip_atomic_load(host_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void l1_norm_reduce_kernel(const float *__restrict__ partial, int n, float scale,
                                                             float *__restrict__ out) {
  __shared__ float red[256];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

int launch_l1_adam(float *p, const float *g, float *m, float *v, long n, double l1, double grad_scale, double lr, double beta1,
                   double beta2, double eps, int step, float *reg_loss, float *partial, hipStream_t s) {
  HN_REQUIRE(p && g && m && v && partial, HN_E_NULL, "l1_adam: NULL pointer");
  HN_REQUIRE(n >= 1 && step >= 1, HN_E_SHAPE, "l1_adam: n=%ld step=%d", n, step);
  HN_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, HN_E_SHAPE, "l1_adam: buffers must be 16-byte aligned");
  // the scalars follow torch/optim/adam.py (_single_tensor_adam): python-double arithmetic, then one rounding to fp32
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  AdamArgs a;
  a.l1 = (float)l1; a.grad_scale = (float)grad_scale; a.beta1 = (float)beta1; a.beta2 = (float)beta2;
  a.one_minus_beta1 = (float)(1.0 - beta1);
  a.one_minus_beta2 = (float)(1.0 - beta2);
  a.step_size = (float)(lr / bc1);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  long want = (n / 4 + 255) / 256;
  int blocks = (int)(want < 1 ? 1 : (want > ADAM_BLOCKS ? ADAM_BLOCKS : want));
  int dev = 0;
  HN_HIP_CHECK(hipGetDevice(&dev));
  const unsigned *word = cluster_status_device_word(dev);
  unsigned *snap = nullptr;
  if (word != nullptr) {
    snap = (unsigned *)(partial + ADAM_BLOCKS);
    hipLaunchKernelGGL(status_snapshot_kernel, dim3(1), dim3(1), 0, s, word, snap);
    HN_LAUNCH_CHECK("status_snapshot");
  }
  hipLaunchKernelGGL(l1_adam_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, a, partial, (const unsigned *)snap);
  HN_LAUNCH_CHECK("l1_adam");
  if (reg_loss) {
    hipLaunchKernelGGL(l1_norm_reduce_kernel, dim3(1), dim3(256), 0, s, partial, blocks, (float)l1, reg_loss);
    HN_LAUNCH_CHECK("l1_norm_reduce");
  }
  return HN_OK;
}

}  // namespace hn
