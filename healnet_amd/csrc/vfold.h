// vfold_body -- the folded value / query projections of a shared-context modality's layers, and (second role) the latent array's
// broadcast + the forward's pre-zeroed flags.  A device function so that it can be a launch of its own (vfold_kernel, chain.hip) or
// a ROLE of the one-launch prelude (prelude_kernel, encode.hip): workgroup (bx, by, bz) of a (gx, n + bc_rows, gz) grid.
#pragma once
// Folded value projection of a shared-context (rank-D) block, (heads * dh, dp): column d of row (h, e) multiplies slot d of the
// merged context average -- gamma folded in, the packed channel order of the context (common.h), and in column dp-1 (the ones
// column of the average) the beta term.  What merge_vproj_kernel builds per workgroup, once per forward here.
__device__ __forceinline__ void vfold_body(const VfoldMulti &v, const int bx, const int by, const int bz, const int gx, const int gz) {
  if (by >= v.n) {               // broadcast role (common.h): x[i] = latents[i mod (l_c * l_d)], 16 bytes per thread and pass
    const long nb = (long)gx * v.bc_rows * gz;
    const long bid = ((long)(by - v.n) * gz + bz) * gx + bx;
    const long n4 = v.bc_total >> 2, per4 = v.bc_per >> 2;
    for (long i = bid * blockDim.x + threadIdx.x; i < n4; i += nb * blockDim.x)
      ((f32x4 *)v.bc_dst)[i] = ((const f32x4 *)v.bc_src)[i % per4];
    if (bid == 0)
      for (int i = threadIdx.x; i < v.bc_nzero; i += blockDim.x) v.bc_zero[i] = 0;
    return;
  }
  const int hi = bx, z = by, dp = 16;
  const float *w_v = v.w_v[z], *gamma = v.gamma[z], *beta = v.beta[z];
  float *out = v.out + (long)z * v.out_stride;
  if (bz > 0) {
    // query side (grid.z = 1 + slices of 32 input columns): the head's folded key weights (what qfold_mfma_kernel stages per
    // workgroup: scale, gamma, packed channel order), then W_f[hi * 16 + d][c] = sum_e wk[e][d] * W_q[hi * dh + e][c]
    __shared__ float wk[128 * 16];
    const float *w_k = v.w_k[z], *w_q = v.w_q[z];
    for (int idx = threadIdx.x; idx < v.dh * dp; idx += blockDim.x) {
      const int e = idx / dp, d = idx % dp;
      const float *wr = w_k + (long)(hi * v.dh + e) * v.D;
      float w = 0.0f;
      if (v.pack_ks == 0) {
        if (d < v.D) w = wr[d] * (gamma ? gamma[d] : 1.0f);
      } else {
        const int c = packed_chan(d, v.pack_ks);
        if (c >= 0 && c < v.D - 1) w = wr[c] * (gamma ? gamma[c] : 1.0f) - wr[v.D - 1] * (gamma ? gamma[v.D - 1] : 1.0f);
      }
      wk[idx] = w * v.cscale;
    }
    __syncthreads();
    // thread: input column c of this block's slice, two of the 16 slots; the W_q column is read 16 rows at a time
    const int c = (bz - 1) * 32 + (threadIdx.x & 31), d0 = (threadIdx.x >> 5) * 2;
    if (c >= v.l_d) return;
    const float *wq = w_q + (long)hi * v.dh * v.l_d + c;
    float a0 = 0.0f, a1 = 0.0f;
    for (int e0 = 0; e0 < v.dh; e0 += 16) {
      float x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = e0 + k < v.dh ? wq[(long)(e0 + k) * v.l_d] : 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int e = e0 + k < v.dh ? e0 + k : 0;
        a0 = fmaf(wk[e * dp + d0], x[k], a0);
        a1 = fmaf(wk[e * dp + d0 + 1], x[k], a1);
      }
    }
    float *qo = v.qout + (long)z * v.qout_stride + (long)hi * dp * v.l_d;
    qo[(long)d0 * v.l_d + c] = a0;
    qo[(long)(d0 + 1) * v.l_d + c] = a1;
    return;
  }
  for (int idx = threadIdx.x; idx < v.dh * dp; idx += blockDim.x) {
    const int e = idx / dp, d = idx % dp;
    const float *wr = w_v + (long)(hi * v.dh + e) * v.D;
    float w = 0.0f;
    if (d == dp - 1) {
      float bc[15], wc[15];                   // all requests first (D <= 15): one round trip instead of D
#pragma unroll
      for (int c = 0; c < 15; ++c) {
        const int cc = c < v.D ? c : 0;
        bc[c] = beta ? beta[cc] : 0.0f;
        wc[c] = wr[cc];
      }
#pragma unroll
      for (int c = 0; c < 15; ++c) w = c < v.D ? fmaf(bc[c], wc[c], w) : w;
    } else if (v.pack_ks == 0) {
      if (d < v.D) w = wr[d] * (gamma ? gamma[d] : 1.0f);
    } else {
      const int c = packed_chan(d, v.pack_ks);
      if (c >= 0 && c < v.D - 1) w = wr[c] * (gamma ? gamma[c] : 1.0f) - wr[v.D - 1] * (gamma ? gamma[v.D - 1] : 1.0f);
    }
    out[(long)(hi * v.dh + e) * dp + d] = w;
  }
}

