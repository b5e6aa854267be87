// A/B bench of the fp32 patch-bag K/V projection (cfg4: 32 768 x 773 -> 1024): the round-3 kernel (gemm_big_kernel behind
// hn::launch_gemm, affine prologue in the loader) against the LDS-DMA kernel of gemm_nt.hip, interleaved in one process on
// uniform random operands, each variant checked against the other and against an fp64 host reference on sampled elements.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -o gemm_f32_bench gemm_f32_bench.hip
#define HN_GEMM_NT_BENCH 1
#include "../../healnet_amd/csrc/gemm.hip"
#include "../../healnet_amd/csrc/gemm_nt.hip"
#include "../../healnet_amd/csrc/backward.hip"
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
  setenv("HN_NO_GLDS_GEMM", "1", 1);      // hn::launch_gemm_ex below = the round-3 route (gemm_tn_lds_kernel)
  const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 773;
  const int rounds = argc > 4 ? atoi(argv[4]) : 5, iters = 10;
  const int lda = (K + 3) / 4 * 4, ldws = hn::gemm_nt_ldws(K);
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> hA((size_t)M * lda), hW((size_t)N * K), hg(K), hb(K);
  for (auto &v : hA) v = U(rng);
  for (auto &v : hW) v = U(rng) * 0.05f;
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

  hn::GemmArgs g0{};
  g0.A = A; g0.lda = lda; g0.W = W; g0.ldw = K; g0.C = C0; g0.ldc = N; g0.M = M; g0.N = N; g0.K = K; g0.batch = 1;
  g0.pro = hn::PRO_AFFINE; g0.gamma = gam; g0.beta = bet; g0.alpha = 1.0f; g0.act = hn::ACT_NONE;
  hn::GemmNtArgs g1{};
  g1.A = A; g1.lda = lda; g1.W = Ws; g1.ldw = ldws; g1.bias = bs; g1.C = C1; g1.ldc = N; g1.M = M; g1.N = N; g1.K = K; g1.alpha = 1.0f;

  // -1: round-3 kernel; >= 0: gemm_nt variants (10..13: ablations of variant 0 -- wrong results by construction)
  const int ids[] = {-1, 0, 1, 2, 3, 4, 10, 11, 12, 13};
  const int nvar = sizeof(ids) / sizeof(ids[0]);
  auto run = [&](int v) {
    if (ids[v] < 0) return hn::launch_gemm(g0, s);
    return hn::launch_gemm_nt(g1, ids[v], s);
  };
  if (hn::launch_gemm_nt_stage(W, K, gam, bet, nullptr, N, K, Ws, bs, s) != 0) return 1;
  CK(hipStreamSynchronize(s));

  // ---- correctness: every variant against fp64 on sampled elements, and the new ones against the round-3 output everywhere
  std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
  if (run(0) != 0) return 1;
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
  double ref_scale = 0.0;
  for (size_t i = 0; i < h0.size(); i += 997) ref_scale = std::max(ref_scale, (double)fabsf(h0[i]));
  auto spot = [&](const std::vector<float> &h) {
    double worst = 0.0;
    std::mt19937 r2(7);
    for (int t = 0; t < 2000; ++t) {
      const int m = t < 64 ? (t < 32 ? t : M - 1 - (t - 32)) : (int)(r2() % M), n = t < 64 ? (t * 37) % N : (int)(r2() % N);
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += ((double)hA[(size_t)m * lda + k] * hg[k] + hb[k]) * hW[(size_t)n * K + k];
      worst = std::max(worst, fabs(acc - h[(size_t)m * N + n]));
    }
    return worst / ref_scale;
  };
  printf("shape M=%d N=%d K=%d  |C|max~%.3f\n", M, N, K, ref_scale);
  printf("variant 0 (gemm_big): fp64 spot rel err %.3e\n", spot(h0));
  int bad = 0;
  for (int v = 1; v < nvar; ++v) {
    CK(hipMemsetAsync(C1, 0xff, (size_t)M * N * 4, s));
    if (run(v) != 0) return 1;
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    size_t nan = 0;
    for (size_t i = 0; i < h1.size(); ++i) {
      if (!(h1[i] == h1[i])) { ++nan; continue; }
      worst = std::max(worst, (double)fabsf(h1[i] - h0[i]));
    }
    size_t flips = 0;
    if (ids[v] < 10) {
      std::vector<float> h2(h1.size());
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(C1, 0xff, (size_t)M * N * 4, s));
        run(0); run(v);                          // (another kernel in between: different L2 / clock state)
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h2.data(), C1, h2.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h1.size(); ++i) flips += memcmp(&h1[i], &h2[i], 4) != 0;
      }
      if (flips) { printf("variant %d: %zu elements differ between repeated runs (RACE)\n", v, flips); bad = 1; }
    }
    const double sp = spot(h1);
    printf("variant %d (gemm_nt %d): vs gemm_big max rel %.3e, NaN %zu, fp64 spot rel err %.3e\n", v, ids[v], worst / ref_scale, nan, sp);
    if (ids[v] < 10 && (nan || worst / ref_scale > 1e-4 || sp > 1e-5)) bad = 1;
  }

  // ---- timing: interleaved rounds
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<std::vector<float>> us(nvar);
  for (int w = 0; w < 30; ++w) run(w % nvar);     // settle the clocks
  for (int r = 0; r < rounds; ++r)
    for (int v = 0; v < nvar; ++v) {
      run(v);
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) run(v);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      us[v].push_back(ms * 1000.0f / iters);
    }
  const double flops = 2.0 * M * N * K;
  for (int v = 0; v < nvar; ++v) {
    std::sort(us[v].begin(), us[v].end());
    const float med = us[v][us[v].size() / 2], mn = us[v][0];
    printf("variant %d [id %d]: median %.1f us (%.1f TF/s, %.3f of 157.3)  min %.1f us\n", v, ids[v], med, flops / med * 1e-6, flops / med * 1e-6 / 157.3, mn);
  }
  // ================= TN: G = dKV^T z (cfg4: 1024 x 773 over 32 768 rows), round-3 route (launch_gemm_ex -> gemm_tn_lds_kernel) vs gemm_tn_glds
  {
    const int TM = N, TNn = K, TK = M;           // 1024 x 773, contraction over the 32 768 rows: A = C0 (M x N, from above), B = A (M x lda)
    float *G0, *G1, *cs0, *cs1, *scr;
    const size_t scr_floats = hn::reduce_scratch_floats((long)TM * TNn, TM);
    CK(hipMalloc(&G0, (size_t)TM * TNn * 4)); CK(hipMalloc(&G1, (size_t)TM * TNn * 4)); CK(hipMalloc(&cs0, TM * 4)); CK(hipMalloc(&cs1, TM * 4));
    CK(hipMalloc(&scr, scr_floats * 4));
    run(0);                                       // C0 = the projected K|V rows, used as "dKV"
    hn::GemmExArgs e{};
    e.A = C0; e.a_rs = 1; e.a_cs = N; e.B = A; e.b_rs = 1; e.b_cs = lda; e.C = G0; e.ldc = TNn; e.M = TM; e.N = TNn; e.K = TK; e.batch = 1;
    e.alpha = 1.0f; e.accumulate = 0; e.colsum = cs0; e.colsum_accumulate = 0;
    auto run_tn = [&](int v) {
      if (v == 0) return hn::launch_gemm_ex(e, s, scr);
      return hn::launch_gemm_tn_glds(C0, N, A, lda, G1, TNn, TM, TNn, TK, 1.0f, 0, scr, scr_floats, cs1, 0, s, v - 1);
    };
    if (run_tn(0) != 0) return 1;
    CK(hipMemsetAsync(G1, 0xff, (size_t)TM * TNn * 4, s));
    if (run_tn(1) != 0) return 1;
    CK(hipStreamSynchronize(s));
    std::vector<float> g0((size_t)TM * TNn), g1v((size_t)TM * TNn), c0(TM), c1(TM);
    CK(hipMemcpy(g0.data(), G0, g0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g1v.data(), G1, g1v.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c0.data(), cs0, TM * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), cs1, TM * 4, hipMemcpyDeviceToHost));
    double sc = 0, worst = 0, csc = 0, cworst = 0; size_t nan = 0;
    for (size_t i = 0; i < g0.size(); ++i) { sc = std::max(sc, (double)fabsf(g0[i])); if (!(g1v[i] == g1v[i])) ++nan; else worst = std::max(worst, (double)fabsf(g0[i] - g1v[i])); }
    for (int i = 0; i < TM; ++i) { csc = std::max(csc, (double)fabsf(c0[i])); cworst = std::max(cworst, (double)fabsf(c0[i] - c1[i])); }
    // fp64 spot check of the new kernel
    double sworst = 0;
    std::mt19937 r3(11);
    for (int t = 0; t < 200; ++t) {
      const int i = t < 8 ? t * 146 % TM : (int)(r3() % TM), j = t < 8 ? TNn - 1 - t : (int)(r3() % TNn);
      double acc = 0;
      for (int r = 0; r < TK; ++r) acc += (double)h0[(size_t)r * N + i] * hA[(size_t)r * lda + j];
      sworst = std::max(sworst, fabs(acc - g1v[(size_t)i * TNn + j]));
    }
    {
      std::vector<float> g2(g1v.size());
      size_t flips = 0;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(G1, 0xff, (size_t)TM * TNn * 4, s));
        run(rep % 3); run_tn(1);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(g2.data(), G1, g2.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < g2.size(); ++i) flips += memcmp(&g1v[i], &g2[i], 4) != 0;
      }
      printf("TN glds: %zu elements differ over 6 repeated runs%s\n", flips, flips ? " (RACE)" : "");
      if (flips) bad = 1;
    }
    printf("TN %d x %d over %d: glds vs round-3 max rel %.3e (NaN %zu), colsum rel %.3e, fp64 spot rel %.3e (scale %.1f)\n", TM, TNn, TK, worst / sc, nan,
           cworst / csc, sworst / sc, sc);
    if (nan || worst / sc > 1e-4 || cworst / csc > 1e-4 || sworst / sc > 1e-4) bad = 1;
    const int ntv = 7;      // 0 round 3; 1 glds (XCD owns a row slice); 2 glds, tile-fastest map; 3..6 ablations: no stores / loads / barriers / all
    std::vector<float> t_us[ntv];
    for (int r = 0; r < rounds; ++r)
      for (int v = 0; v < ntv; ++v) {
        run_tn(v);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run_tn(v);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t_us[v].push_back(ms * 1000.0f / iters);
      }
    for (int v = 0; v < ntv; ++v) {
      std::sort(t_us[v].begin(), t_us[v].end());
      const float med = t_us[v][t_us[v].size() / 2];
      printf("TN variant %d (%s, incl. its reduce): median %.1f us (%.1f TF/s, %.3f of 157.3)  min %.1f us\n", v, v ? "gemm_tn_glds" : "round 3", med,
             2.0 * TM * TNn * TK / med * 1e-6, 2.0 * TM * TNn * TK / med * 1e-6 / 157.3, t_us[v][0]);
    }
  }
  // ================= TN with a narrow A (G = dKV^T z of a one-head model: M2 = 32 / 64 / 128 / 256 rows x 773 over the 32 768 rows)
  for (int M2 : {32, 64, 128, 256, 512}) {
    if (M2 > N) continue;
    const int TNn = K, TK = M;
    float *G0, *G1, *cs0, *cs1, *scr;
    const size_t scr_floats = hn::reduce_scratch_floats((long)M2 * TNn, M2);
    CK(hipMalloc(&G0, (size_t)M2 * TNn * 4)); CK(hipMalloc(&G1, (size_t)M2 * TNn * 4)); CK(hipMalloc(&cs0, M2 * 4)); CK(hipMalloc(&cs1, M2 * 4));
    CK(hipMalloc(&scr, scr_floats * 4));
    // A = the first M2 columns of C0 taken as a dense (TK, M2) matrix with pitch M2: reinterpret the buffer (values are arbitrary finite numbers)
    hn::GemmExArgs e{};
    e.A = C0; e.a_rs = 1; e.a_cs = M2; e.B = A; e.b_rs = 1; e.b_cs = lda; e.C = G0; e.ldc = TNn; e.M = M2; e.N = TNn; e.K = TK; e.batch = 1;
    e.alpha = 1.0f; e.accumulate = 0; e.colsum = cs0; e.colsum_accumulate = 0;
    setenv("HN_NO_GLDS_GEMM", "1", 1);
    auto run2 = [&](int v) {
      if (v == 0) return hn::launch_gemm_ex(e, s, scr);
      return hn::launch_gemm_tn_glds(C0, M2, A, lda, G1, TNn, M2, TNn, TK, 1.0f, 0, scr, scr_floats, cs1, 0, s, v);
    };
    const int forced[4] = {0, 32, 64, 128};
    if (run2(0) != 0) return 1;
    CK(hipStreamSynchronize(s));
    std::vector<float> g0v((size_t)M2 * TNn), g1v((size_t)M2 * TNn), c0(M2), c1(M2), g2v((size_t)M2 * TNn);
    CK(hipMemcpy(g0v.data(), G0, g0v.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c0.data(), cs0, M2 * 4, hipMemcpyDeviceToHost));
    for (int fv = 1; fv < 4; ++fv) {
      if (forced[fv] < 128 && forced[fv] * 8 < M2) continue;      // (keep the tile counts sane)
      CK(hipMemsetAsync(G1, 0xff, (size_t)M2 * TNn * 4, s)); CK(hipMemsetAsync(cs1, 0xff, M2 * 4, s));
      if (run2(forced[fv]) != 0) return 1;
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(g1v.data(), G1, g1v.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), cs1, M2 * 4, hipMemcpyDeviceToHost));
      double sc = 0, worst = 0, csc = 0, cworst = 0; size_t nan = 0, flips = 0;
      for (size_t i = 0; i < g0v.size(); ++i) { sc = std::max(sc, (double)fabsf(g0v[i])); if (!(g1v[i] == g1v[i])) ++nan; else worst = std::max(worst, (double)fabsf(g0v[i] - g1v[i])); }
      for (int i = 0; i < M2; ++i) { csc = std::max(csc, (double)fabsf(c0[i])); if (!(c1[i] == c1[i])) ++nan; else cworst = std::max(cworst, (double)fabsf(c0[i] - c1[i])); }
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(G1, 0xff, (size_t)M2 * TNn * 4, s));
        run(rep % 3); run2(forced[fv]);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(g2v.data(), G1, g2v.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < g2v.size(); ++i) flips += memcmp(&g1v[i], &g2v[i], 4) != 0;
      }
      run(0);
      std::vector<float> tt;
      for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run2(forced[fv]);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tt.push_back(ms * 1000.0f / iters);
      }
      std::sort(tt.begin(), tt.end());
      std::vector<float> t2;
      hn::g_tn_bench_skip_reduce = true;
      for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run2(forced[fv]);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t2.push_back(ms * 1000.0f / iters);
      }
      hn::g_tn_bench_skip_reduce = false;
      std::sort(t2.begin(), t2.end());
      printf("  (without the reduce: %.1f us)  ", t2[t2.size() / 2]);
      printf("TN narrow M=%d row tile %d (incl. reduce): %.1f us  max rel %.2e colsum rel %.2e NaN %zu flips %zu%s\n", M2, forced[fv], tt[tt.size() / 2],
             worst / sc, cworst / csc, nan, flips, flips ? " (RACE)" : "");
      if (nan || flips || worst / sc > 1e-4 || cworst / csc > 1e-4) bad = 1;
    }
    {
      std::vector<float> tt;
      for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run2(0);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tt.push_back(ms * 1000.0f / iters);
      }
      std::sort(tt.begin(), tt.end());
      printf("TN narrow M=%d round-3 route: %.1f us\n", M2, tt[tt.size() / 2]);
    }
    unsetenv("HN_NO_GLDS_GEMM");
    CK(hipFree(G0)); CK(hipFree(G1)); CK(hipFree(cs0)); CK(hipFree(cs1)); CK(hipFree(scr));
  }
  // ================= narrow outputs (one head of a tuned shape: N = 32 / 64 / 128 columns over the same 32 768 x 773 context)
  for (int N2 : {32, 64, 128}) {
    float *W2, *Ws2, *bs2, *D0, *D1;
    CK(hipMalloc(&W2, (size_t)N2 * K * 4)); CK(hipMalloc(&Ws2, (size_t)N2 * ldws * 4 + 4096)); CK(hipMalloc(&bs2, 512 * 4));
    CK(hipMalloc(&D0, (size_t)M * N2 * 4)); CK(hipMalloc(&D1, (size_t)M * N2 * 4));
    CK(hipMemcpy(W2, hW.data(), (size_t)N2 * K * 4, hipMemcpyHostToDevice));
    hn::GemmArgs a0 = g0; a0.W = W2; a0.N = N2; a0.C = D0; a0.ldc = N2;
    hn::GemmNtArgs a1 = g1; a1.W = Ws2; a1.bias = bs2; a1.N = N2; a1.C = D1; a1.ldc = N2;
    const int var = N2 == 32 ? 20 : (N2 == 64 ? 21 : 22);
    if (hn::launch_gemm_nt_stage(W2, K, gam, bet, nullptr, N2, K, Ws2, bs2, s) != 0) return 1;
    if (hn::launch_gemm(a0, s) != 0 || hn::launch_gemm_nt(a1, var, s) != 0) return 1;
    CK(hipStreamSynchronize(s));
    std::vector<float> d0((size_t)M * N2), d1((size_t)M * N2);
    CK(hipMemcpy(d0.data(), D0, d0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d1.data(), D1, d1.size() * 4, hipMemcpyDeviceToHost));
    double w = 0, sc = 0;
    for (size_t i = 0; i < d0.size(); ++i) { sc = std::max(sc, (double)fabsf(d0[i])); w = std::max(w, (double)fabsf(d0[i] - d1[i])); if (!(d1[i] == d1[i])) w = 1e30; }
    if (w / sc > 1e-4) bad = 1;
    float t[2] = {0, 0};
    for (int v = 0; v < 2; ++v) {
      std::vector<float> tt;
      for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) { if (v == 0) hn::launch_gemm(a0, s); else hn::launch_gemm_nt(a1, var, s); }
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tt.push_back(ms * 1000.0f / iters);
      }
      std::sort(tt.begin(), tt.end()); t[v] = tt[tt.size() / 2];
    }
    const double bytes = (double)M * lda * 4 + (double)M * N2 * 4;
    printf("narrow N=%d: tall_narrow %.1f us (%.2f TB/s)  gemm_nt[%d] %.1f us (%.2f TB/s)  max rel diff %.2e\n", N2, t[0], bytes / t[0] * 1e-6, var, t[1],
           bytes / t[1] * 1e-6, w / sc);
  }
  return bad;
}
