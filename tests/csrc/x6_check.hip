// Test infrastructure (never loaded by the product): the patch-bag GEMMs of healnet_amd/csrc/gemm_x6.hip (fp32-exact products on the
// bf16 pipe from three-plane images) and of gemm_nt.hip (fp32 MFMA) behind two plain C entry points, so tests/test_gpu_x6.py can
// hold BOTH against an fp64 product of the same operands: the claim under test is that the split kernels' error is not larger than
// the fp32 MFMA's.  Built by __graft_entry__.build_test_helpers() from the product's own sources.
#include "../../healnet_amd/csrc/gemm_nt.hip"
#include "../../healnet_amd/csrc/gemm_x6.hip"

namespace hn {
void debug_after_launch(hipStream_t) {}
KernelTimerScope::KernelTimerScope(const char *, hipStream_t stream) : stop(nullptr), s(stream) {}
void set_error(const char *, ...) {}
int fail(int code, const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
  return code;
}
}  // namespace hn

#define X6C(x) do { int rc_ = (x); if (rc_ != 0) return rc_; } while (0)

// C = A W^T (+ 0): A (M, K) pitch lda, W (N, K) pitch K.  ws: scratch of x6_check_ws_bytes(M, N, K) bytes.  route 0: fp32 MFMA, 1: x6
extern "C" __attribute__((visibility("default"))) size_t x6_check_ws_bytes(long M, int N, int K) {
  const size_t nt = hn::x6_plane_bytes(M, K, 8) + hn::x6_plane_bytes(N, K, 8) + (size_t)N * hn::gemm_nt_ldws(K) * 4 + (size_t)N * 4 + 8192;
  const size_t tn = hn::gemm_tn_x6_image_bytes(M, N, 8) + hn::gemm_tn_x6_image_bytes(M, K + 1, 5) + (size_t)40 * ((size_t)N * (K + 160) + N) * 4 + (32u << 20);
  return nt > tn ? nt : tn;
}

extern "C" __attribute__((visibility("default"))) int x6_check_nt(const float *A, long lda, const float *W, long M, int N, int K, float *C, int route,
                                                                  void *ws, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  char *p = (char *)ws;
  float *Ws = (float *)p; p += ((size_t)N * hn::gemm_nt_ldws(K) * 4 + 255) / 256 * 256;
  float *bs = (float *)p; p += ((size_t)N * 4 + 255) / 256 * 256;
  X6C(hn::launch_gemm_nt_stage(W, K, nullptr, nullptr, nullptr, N, K, Ws, bs, s));
  if (route == 0) {
    hn::GemmNtArgs g{};
    g.A = A; g.lda = lda; g.W = Ws; g.ldw = hn::gemm_nt_ldws(K); g.bias = bs; g.C = C; g.ldc = N; g.M = (int)M; g.N = N; g.K = K; g.alpha = 1.0f;
    return hn::launch_gemm_nt(g, 0, s);
  }
  unsigned short *Ap = (unsigned short *)p; p += hn::x6_plane_bytes(M, K, 8);
  unsigned short *Wp = (unsigned short *)p;
  X6C(hn::launch_x6_split(A, lda, nullptr, M, K, 8, Ap, s));
  X6C(hn::launch_x6_split(Ws, hn::gemm_nt_ldws(K), nullptr, N, K, 8, Wp, s));
  hn::GemmX6Args g{};
  g.Ap = Ap; g.a_rt = hn::x6_row_tiles(M); g.Wp = Wp; g.w_rt = hn::x6_row_tiles(N); g.bias = bs; g.C = C; g.ldc = N; g.M = (int)M; g.N = N;
  g.KT = (K + 15) / 16; g.alpha = 1.0f;
  return hn::launch_gemm_nt_x6(g, 0, s);
}

// G (M, N) = A^T B, colsum (M) = column sums of A: A (R, M) pitch M, B (R, N) pitch ldb
extern "C" __attribute__((visibility("default"))) int x6_check_tn(const float *A, const float *B, long ldb, long R, int M, int N, float *G, float *colsum,
                                                                  int route, void *ws, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  char *p = (char *)ws;
  const size_t scr_floats = (size_t)40 * ((size_t)M * (N + 160) + M) + (6u << 20);
  float *scr = (float *)p; p += (scr_floats * 4 + 255) / 256 * 256;
  if (route == 0) return hn::launch_gemm_tn_glds(A, M, B, ldb, G, N, M, N, (int)R, 1.0f, 0, scr, scr_floats, colsum, 0, s);
  unsigned short *At = (unsigned short *)p; p += hn::gemm_tn_x6_image_bytes(R, M, 8);
  unsigned short *Bt = (unsigned short *)p;
  X6C(hn::launch_x6_split_t(A, M, R, M, 8, -1, At, s));
  X6C(hn::launch_x6_split_t(B, ldb, R, N, 5, N, Bt, s));
  return hn::launch_gemm_tn_x6(At, Bt, R, M, N, G, N, colsum, scr, scr_floats, s);
}
