// A/B bench of the patch-bag K/V projection (cfg4: 32 768 x 773 -> 1024): the fp32-MFMA LDS-DMA kernel of gemm_nt.hip against
// the three-plane bf16 kernel of gemm_x6.hip (fp32-exact: six bf16 products per fp32 product), interleaved in one process on the
// same operands.  Both are checked against an fp64 host reference on sampled elements; the point of the comparison is that the
// split kernel's error is NOT larger than the fp32 MFMA's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -o gemm_x6_bench gemm_x6_bench.hip
#define HN_GEMM_NT_BENCH 1
#include "../../healnet_amd/csrc/gemm_nt.hip"
#include "../../healnet_amd/csrc/gemm_x6.hip"
#include <vector>
#include <algorithm>
#include <random>
#include <string>
#include <string.h>

namespace hn {
void debug_after_launch(hipStream_t) {}
KernelTimerScope::KernelTimerScope(const char *, hipStream_t stream) : stop(nullptr), s(stream) {}
void set_error(const char *, ...) {}
int fail(int code, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
  return code;
}
}  // namespace hn

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 773;
  const int rounds = argc > 4 ? atoi(argv[4]) : 5, iters = 10;
  const int dist = argc > 5 ? atoi(argv[5]) : 0;      // 0: uniform, 1: normal x wide log-scale (stress for the split)
  const int lda = (K + 3) / 4 * 4, ldws = hn::gemm_nt_ldws(K);
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::normal_distribution<float> G(0.f, 1.f);
  std::vector<float> hA((size_t)M * lda), hW((size_t)N * K), hg(K), hb(K);
  for (auto &v : hA) v = dist ? G(rng) * expf(3.0f * U(rng)) : U(rng);
  for (auto &v : hW) v = (dist ? G(rng) : U(rng)) * 0.05f;
  for (auto &v : hg) v = 1.0f + 0.3f * U(rng);
  for (auto &v : hb) v = 0.2f * U(rng);
  float *A, *W, *gam, *bet, *C0, *C1, *Ws, *bs;
  CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&gam, K * 4)); CK(hipMalloc(&bet, K * 4));
  CK(hipMalloc(&C0, (size_t)M * N * 4)); CK(hipMalloc(&C1, (size_t)M * N * 4));
  CK(hipMalloc(&Ws, (size_t)N * ldws * 4 + 4096)); CK(hipMalloc(&bs, N * 4));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(gam, hg.data(), K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bet, hb.data(), K * 4, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned short *Ap, *Wp;
  const size_t a_bytes = hn::x6_plane_bytes(M, K, 8), w_bytes = hn::x6_plane_bytes(N, K, 8);
  CK(hipMalloc(&Ap, a_bytes)); CK(hipMalloc(&Wp, w_bytes));

  hn::GemmNtArgs g1{};
  g1.A = A; g1.lda = lda; g1.W = Ws; g1.ldw = ldws; g1.bias = bs; g1.C = C0; g1.ldc = N; g1.M = M; g1.N = N; g1.K = K; g1.alpha = 1.0f;
  hn::GemmX6Args g2{};
  g2.Ap = Ap; g2.a_rt = (int)(a_bytes / 3072 / ((K + 15) / 16)); g2.Wp = Wp; g2.w_rt = (int)(w_bytes / 3072 / ((K + 15) / 16));
  g2.bias = bs; g2.C = C1; g2.ldc = N; g2.M = M; g2.N = N; g2.KT = (K + 15) / 16; g2.alpha = 1.0f;

  if (hn::launch_gemm_nt_stage(W, K, gam, bet, nullptr, N, K, Ws, bs, s) != 0) return 1;
  auto split = [&]() {
    if (hn::launch_x6_split(A, lda, nullptr, M, K, 8, Ap, s) != 0) exit(1);
    if (hn::launch_x6_split(W, K, gam, N, K, 8, Wp, s) != 0) exit(1);
  };
  split();
  CK(hipStreamSynchronize(s));

  const int ids[] = {-1, 0, 1, 10, 11, 12, 13};      // -1: fp32 MFMA; 0 / 1: x6 variants; >= 10: ablations (wrong results)
  const int nvar = sizeof(ids) / sizeof(ids[0]);
  auto run = [&](int v) { return ids[v] < 0 ? hn::launch_gemm_nt(g1, 0, s) : hn::launch_gemm_nt_x6(g2, ids[v], s); };

  std::vector<float> h((size_t)M * N);
  std::vector<int> sm, sn;
  std::vector<double> ref;
  {
    std::mt19937 r2(7);
    for (int t = 0; t < 4000; ++t) {
      const int m = t < 64 ? (t < 32 ? t : M - 1 - (t - 32)) : (int)(r2() % M), n = t < 64 ? (t * 37) % N : (int)(r2() % N);
      double acc = 0.0, bsum = 0.0;
      // the staged operands as the device holds them: W * gamma rounded to fp32 (both kernels see that rounding), bias = W beta
      for (int k = 0; k < K; ++k) {
        const float wg = hW[(size_t)n * K + k] * hg[k];
        acc += (double)hA[(size_t)m * lda + k] * (double)wg;
        bsum += (double)hW[(size_t)n * K + k] * hb[k];
      }
      sm.push_back(m); sn.push_back(n); ref.push_back(acc + bsum);
    }
  }
  printf("shape M=%d N=%d K=%d dist=%d\n", M, N, K, dist);
  std::vector<float> hfirst;
  for (int v = 0; v < nvar; ++v) {
    CK(hipMemsetAsync(ids[v] < 0 ? C0 : C1, 0xff, (size_t)M * N * 4, s));
    if (run(v) != 0) return 1;
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), ids[v] < 0 ? C0 : C1, h.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, sq = 0.0, scale = 0.0;
    for (size_t t = 0; t < ref.size(); ++t) {
      const double d = fabs(ref[t] - (double)h[(size_t)sm[t] * N + sn[t]]);
      worst = std::max(worst, d); sq += d * d; scale = std::max(scale, fabs(ref[t]));
    }
    size_t nan = 0;
    double maxdiff = 0.0;
    if (v == 0) hfirst = h;
    else
      for (size_t i = 0; i < h.size(); ++i) {
        if (!(h[i] == h[i])) ++nan;
        else maxdiff = std::max(maxdiff, (double)fabsf(h[i] - hfirst[i]));
      }
    printf("variant %2d: vs fp64 max %.3e rms %.3e (|C|max %.3f)   vs fp32-MFMA everywhere: max %.3e, %zu NaN\n", ids[v], worst,
           sqrt(sq / ref.size()), scale, maxdiff, nan);
  }

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flops = 2.0 * M * N * K;
  for (int r = 0; r < rounds; ++r) {
    for (int v = 0; v < nvar; ++v) {
      run(v);
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) run(v);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("round %d variant %2d: %8.1f us  %7.1f TF/s (fp32-equivalent)\n", r, ids[v], ms * 1000 / iters, flops / (ms / iters * 1e-3) * 1e-12);
    }
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) split();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("round %d split (bag + weight): %8.1f us\n", r, ms * 1000 / iters);
  }

  // ---- the weight gradient G = dKV^T z: (M x K rows) ... here A2 = C0 (K2 = M rows x N columns: stands for dKV), B2 = A (the bag)
  {
    const int K2 = M, M2 = N, N2 = K;      // contraction over the bag's rows
    float *G0, *G1, *cs0, *cs1, *scr;
    const size_t scr_floats = (size_t)32 * ((size_t)M2 * 800 + M2) + (6u << 20);
    CK(hipMalloc(&G0, (size_t)M2 * N2 * 4)); CK(hipMalloc(&G1, (size_t)M2 * N2 * 4)); CK(hipMalloc(&cs0, M2 * 4)); CK(hipMalloc(&cs1, M2 * 4));
    CK(hipMalloc(&scr, scr_floats * 4));
    unsigned short *At, *Bt;
    CK(hipMalloc(&At, hn::gemm_tn_x6_image_bytes(K2, M2, 8))); CK(hipMalloc(&Bt, hn::gemm_tn_x6_image_bytes(K2, N2 + 1, 5)));
    auto tn32 = [&]() { return hn::launch_gemm_tn_glds(C0, N, A, lda, G0, N2, M2, N2, K2, 1.0f, 0, scr, scr_floats, cs0, 0, s); };
    auto splitA = [&]() { return hn::launch_x6_split_t(C0, N, K2, M2, 8, -1, At, s); };
    auto splitB = [&]() { return hn::launch_x6_split_t(A, lda, K2, N2, 5, N2, Bt, s); };
    auto tnx6 = [&]() { return hn::launch_gemm_tn_x6(At, Bt, K2, M2, N2, G1, N2, cs1, scr, scr_floats, s); };
    if (!hn::gemm_tn_x6_eligible(K2, M2, N2)) { printf("TN: shape not eligible\n"); return 0; }
    run(0);      // C0 = the fp32 product: the stand-in for dKV
    CK(hipStreamSynchronize(s));
    std::vector<float> hD((size_t)M * N), g0((size_t)M2 * N2), g1((size_t)M2 * N2), c0(M2), c1(M2);
    CK(hipMemcpy(hD.data(), C0, hD.size() * 4, hipMemcpyDeviceToHost));
    if (tn32() != 0 || splitA() != 0 || splitB() != 0 || tnx6() != 0) return 1;
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(g0.data(), G0, g0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g1.data(), G1, g1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c0.data(), cs0, M2 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), cs1, M2 * 4, hipMemcpyDeviceToHost));
    double w0 = 0, w1 = 0, q0 = 0, q1 = 0, sc = 0, cw0 = 0, cw1 = 0, csc = 0;
    std::mt19937 r3(11);
    const int NS = 600;
    for (int t = 0; t < NS; ++t) {
      const int i = t < 8 ? t * 131 % M2 : (int)(r3() % M2), j = t < 8 ? (t & 1 ? N2 - 1 - t : t) : (int)(r3() % N2);
      double acc = 0.0;
      for (int r = 0; r < K2; ++r) acc += (double)hD[(size_t)r * N + i] * (double)hA[(size_t)r * lda + j];
      const double d0 = fabs(acc - g0[(size_t)i * N2 + j]), d1 = fabs(acc - g1[(size_t)i * N2 + j]);
      w0 = std::max(w0, d0); w1 = std::max(w1, d1); q0 += d0 * d0; q1 += d1 * d1; sc = std::max(sc, fabs(acc));
    }
    for (int t = 0; t < 64; ++t) {
      const int i = (int)(r3() % M2);
      double acc = 0.0;
      for (int r = 0; r < K2; ++r) acc += (double)hD[(size_t)r * N + i];
      cw0 = std::max(cw0, fabs(acc - c0[i])); cw1 = std::max(cw1, fabs(acc - c1[i])); csc = std::max(csc, fabs(acc));
    }
    double md = 0; size_t nan = 0;
    for (size_t i = 0; i < g0.size(); ++i) { if (!(g1[i] == g1[i])) ++nan; else md = std::max(md, (double)fabsf(g0[i] - g1[i])); }
    printf("TN %d x %d over %d rows (|G|max %.3f): fp32-MFMA vs fp64 max %.3e rms %.3e | x6 max %.3e rms %.3e | x6 vs fp32 everywhere %.3e, %zu NaN\n",
           M2, N2, K2, sc, w0, sqrt(q0 / NS), w1, sqrt(q1 / NS), md, nan);
    printf("TN column sums (|cs|max %.3f): fp32-MFMA max err %.3e | x6 max err %.3e\n", csc, cw0, cw1);
    auto timeit = [&](const char *name, auto fn) {
      fn();
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) fn();
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  %-28s %8.1f us\n", name, ms * 1000 / iters);
    };
    for (int r = 0; r < rounds; ++r) {
      timeit("TN fp32-MFMA (gemm_tn_glds)", tn32);
      timeit("TN x6 (+ reduce)", tnx6);
      timeit("split_t dKV (32768 x 1024)", splitA);
      timeit("split_t bag (32768 x 773)", splitB);
    }
  }
  return 0;
}
