// Stand-alone timing + spot check of the bf16 K/V projection (healnet_amd/csrc/gemm_bf16.hip), development tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/gemm_bf16_bench.hip -o tools/ubench/gemm_bf16_bench
//   tools/ubench/gemm_bf16_bench [M=32768] [N=1024] [K=773] [iters=50]
#define HN_GEMM_BF16_BENCH 1
#include "../../healnet_amd/csrc/gemm_bf16.hip"
#include <vector>
#include <cstring>
#include <cmath>
#include <cstdlib>

namespace hn {
int fail(int code, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n");
  return code;
}
void debug_after_launch(hipStream_t) {}
}  // namespace hn

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float bf16_round(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float y; memcpy(&y, &u, 4); return y;
}

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 773;
  const int iters = argc > 4 ? atoi(argv[4]) : 50;
  const long lda = (K + 3) / 4 * 4, ldc = N;
  std::vector<float> A((size_t)M * lda), W((size_t)N * K), gam(K), bet(K);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
  for (auto &v : A) v = rnd();
  for (auto &v : W) v = rnd() * 0.05f;
  for (auto &v : gam) v = 1.0f + 0.1f * rnd();
  for (auto &v : bet) v = 0.1f * rnd();
  float *dA, *dW, *dg, *db, *dC, *dstage; uint16_t *dAb;
  const int Kp = hn::gemm_bf16_pitch(K);
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dg, K * 4)); CK(hipMalloc(&db, K * 4));
  CK(hipMalloc(&dC, (size_t)M * ldc * 4)); CK(hipMalloc(&dstage, hn::gemm_bf16_stage_floats(N, K) * 4)); CK(hipMalloc(&dAb, (size_t)M * Kp * 2));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, gam.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bet.data(), K * 4, hipMemcpyHostToDevice));
  hn::GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = dA; g.lda = lda; g.W = dW; g.ldw = K; g.C = dC; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.batch = 1; g.alpha = 1.0f;
  g.pro = hn::PRO_AFFINE; g.gamma = dg; g.beta = db;
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  if (hn::launch_rows_to_bf16(dA, lda, M, K, dAb, s) != 0) return 1;
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < 20; ++i) if (hn::launch_rows_to_bf16(dA, lda, M, K, dAb, s) != 0) return 1;
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
  printf("rows_to_bf16: %.1f us\n", ms * 1e3 / 20);
  for (int i = 0; i < 5; ++i) if (hn::launch_gemm_bf16(g, dAb, dstage, s) != 0) return 1;
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) if (hn::launch_gemm_bf16(g, dAb, dstage, s) != 0) return 1;
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  printf("M=%d N=%d K=%d: stage + gemm %.1f us = %.0f TF/s (C write %.0f MB + A read %.0f MB -> %.2f TB/s)\n", M, N, K, us,
         2.0 * M * N * K / us * 1e-6, M * (double)N * 4e-6, M * (double)Kp * 2e-6, (M * (double)N * 4 + M * (double)Kp * 2) / us * 1e-6);
  // spot check against a host evaluation with the same roundings
  std::vector<float> C((size_t)M * ldc);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int t = 0; t < 400; ++t) {
    const int m = t < 8 ? M - 1 - t : rand() % M, n = rand() % N;
    double acc = 0, cb = 0;
    for (int k = 0; k < K; ++k) {
      acc += (double)bf16_round(A[(size_t)m * lda + k]) * (double)bf16_round(W[(size_t)n * K + k] * gam[k]);
      cb += (double)W[(size_t)n * K + k] * bet[k];
    }
    worst = fmax(worst, fabs(acc + cb - C[(size_t)m * ldc + n]));
  }
  printf("max |err| vs host (same roundings) over 400 samples: %.3g\n", worst);
  return (worst < 1e-3 || getenv("HN_BF16_ABL")) ? 0 : 2;
}
