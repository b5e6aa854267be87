// C-ABI entry points of libhealnet_hip.so and the host-side orchestration of the fusion forward
// (the launch schedule that replaces HealNet.forward :190-250 of the reference).
#include "common.h"
#include <mutex>

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>
#include <vector>

namespace hn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// The one process-wide registration of this library (include/healnet_hip.h hn_set_kernel_timers): a training step launches its
// forward from the caller's thread and its backward from the autograd engine's device thread, and one table has to see both, so
// it cannot be thread-local.  What keeps it safe beside concurrent callers:
//   * publication is a seqlock -- a launching thread reads (table, n) as a consistent pair or not at all, never a new table with
//     an old count while another thread re-arms it;
//   * an entry only ever brackets launches on ITS stream (hn_kernel_timer.stream; NULL = any stream): a second thread working on
//     another stream is neither timed nor does it consume event pairs;
//   * slots are claimed with an atomic increment.
static hn_kernel_timer *g_timers = nullptr;
static int g_ntimers = 0;
static unsigned g_timer_seq = 0;          // even: stable; odd: being rewritten
KernelTimerScope::KernelTimerScope(const char *kernel, hipStream_t stream) : stop(nullptr), s(stream) {
  hn_kernel_timer *tab;
  int n;
  for (;;) {
    const unsigned s0 = __atomic_load_n(&g_timer_seq, __ATOMIC_ACQUIRE);
    tab = __atomic_load_n(&g_timers, __ATOMIC_RELAXED);
    n = __atomic_load_n(&g_ntimers, __ATOMIC_RELAXED);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if ((s0 & 1u) == 0 && __atomic_load_n(&g_timer_seq, __ATOMIC_RELAXED) == s0) break;
  }
  if (tab == nullptr) return;
  for (int i = 0; i < n; ++i) {
    hn_kernel_timer &t = tab[i];
    if (t.kernel && strcmp(t.kernel, kernel) == 0 && (t.stream == nullptr || t.stream == (void *)stream)) {
      const int slot = __atomic_fetch_add(&t.n_recorded, 1, __ATOMIC_RELAXED);
      if (slot < t.n_events) {
        (void)hipEventRecord((hipEvent_t)t.ev_start[slot], stream);
        stop = (hipEvent_t)t.ev_stop[slot];
      }
      return;
    }
  }
}

static inline DropCfg drop_off() { return make_drop(0.0f, 0, 0, 0); }
static inline DropCfg drop_of(float p, const hn_rng &r, bool ff) {
  return p > 0.0f ? make_drop(p, r.seed, r.offset, ff ? (r.stream | DROP_SID_FF) : (r.stream & ~DROP_SID_FF), r.offset_dev) : drop_off();
}
// development switches, read ONCE per process (no getenv on the launch path)
static bool chain_disabled() { static const bool off = getenv("HN_NO_CHAIN") != nullptr; return off; }
static bool bchain_disabled() { static const bool off = getenv("HN_NO_BCHAIN") != nullptr; return off; }
static bool qfold_chain_disabled() { static const bool off = getenv("HN_NO_QFOLD_CHAIN") != nullptr; return off; }
static bool merge_chain_disabled() { static const bool off = tuning_env("HN_NO_MERGE_CHAIN") != nullptr; return off; }
// operands the latent chain reads with 16-byte loads: an unaligned one (a parameter that is a view at an odd float offset of a
// user-made flat buffer, a tape / trace slot) sends the block down the per-block launches instead of failing the forward
static inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }
static inline bool chain_ff_aligned(const hn_ff_params *f) { return al16(f->w1) && al16(f->w2) && al16(f->norm_w) && al16(f->norm_b); }
static inline bool chain_out_aligned(const hn_attn_params *a) { return al16(a->w_out); }
static inline bool chain_proj_aligned(const hn_attn_params *a) { return al16(a->w_q) && al16(a->w_kv) && al16(a->norm_w) && al16(a->norm_b); }
static inline int pad_head_dim(int dh) { return dh <= 16 ? 16 : dh <= 32 ? 32 : dh <= 64 ? 64 : dh <= 128 ? 128 : 0; }
static inline int round16(int v) { return (v + 15) / 16 * 16; }
static inline int up128(int v) { return (v + 127) / 128 * 128; }
// the latent chains work on 16-row tiles: every (b * l_c, .) buffer they touch is allocated for the row count rounded up to 16
static inline size_t rows16(size_t rows) { return (rows + 15) / 16 * 16; }
// Staged layout (include/healnet_hip.h): zero-padded images of a narrower block.  The softmax scale and the LayerNorm statistics
// come from the valid widths; a staged block's w_out rows have the padded pitch (w_q / w_kv only gain zero rows at the end).
static inline bool staged_attn(const hn_attn_params *p) { return p->query_dim_valid > 0; }
static inline int dh_valid(const hn_attn_params *p) { return p->dim_head_valid > 0 ? p->dim_head_valid : p->dim_head; }
static inline int wo_ld(const hn_attn_params *p) { const int inner = p->heads * p->dim_head; return staged_attn(p) ? up128(inner) : inner; }
// a LayerNorm over fewer columns than the operand has exists in the chain kernels, ln_fwd and the head only
static inline bool narrow_ln(const hn_attn_params *p) { return p->query_dim_valid > 0 && p->query_dim_valid < p->query_dim; }
static inline bool narrow_ln(const hn_ff_params *p) { return p->dim_valid > 0 && p->dim_valid < p->dim; }

// ------------------------------------------------------------------------------------------------
// attention block
// ------------------------------------------------------------------------------------------------
struct Bf16Context {              // bf16 images of one modality's normalised context (encode.hip)
  const uint16_t *zb, *zT;
  int Np, DV, ns;
};

struct AttnPlan {
  int heads, dh, inner, dhp, Lp, N, D, dp;
  bool rank_d, self_attn, ones, bf16core;
  const uint16_t *ctx16;          // explicit binding under core_precision = bf16 (inference): bf16 image of the context rows -> the K/V projection runs on bf16 MFMA
  int nsplit, chunk;
  int nq;                         // query tiles per wave the forward split was planned for (0: the kernel's default)
  int nsplit_bwd, chunk_bwd;      // token split of attn_bwd_dq_kernel (≈150 VGPRs: 3 waves per SIMD)
  float cscale;
  // workspace carve
  float *obuf, *q, *qf, *kv, *opart, *mpart, *lpart, *bound;
  float *wstage;                  // explicit cross binding: scratch of launch_gemm_bf16 (bf16 image of to_kv.weight + its beta row)
  size_t bytes;
};

static int plan_attn(const hn_attn_params *p, bool has_ctx, int ld_ctx, int b, int L, int N, int D, void *ws,
                     size_t ws_bytes, AttnPlan *pl, int bf16core = 0 /* 0: fp32 core, else the number of bf16 operand planes */) {
  HN_REQUIRE(p, HN_E_NULL, "attn: params NULL");
  HN_REQUIRE(p->heads > 0 && p->dim_head > 0 && p->query_dim > 0 && b > 0 && L > 0, HN_E_SHAPE,
             "attn: heads=%d dim_head=%d query_dim=%d b=%d L=%d", p->heads, p->dim_head, p->query_dim, b, L);
  pl->heads = p->heads;
  pl->dh = p->dim_head;
  pl->inner = p->heads * p->dim_head;
  pl->dhp = pad_head_dim(p->dim_head);
  HN_REQUIRE(pl->dhp != 0, HN_E_UNSUPPORTED, "attn: dim_head=%d > 128 is not supported", p->dim_head);
  pl->Lp = round16(L);
  pl->self_attn = !has_ctx;
  pl->N = has_ctx ? N : L;
  pl->D = has_ctx ? D : p->query_dim;
  HN_REQUIRE(pl->N > 0 && pl->D > 0, HN_E_SHAPE, "attn: N=%d D=%d", pl->N, pl->D);
  HN_REQUIRE(!has_ctx || ld_ctx >= D, HN_E_SHAPE, "attn: ld_ctx=%d < D=%d", ld_ctx, D);
  pl->rank_d = has_ctx && (ld_ctx == 16 || ld_ctx == 32) && D <= ld_ctx && ld_ctx <= pl->dhp;
  pl->ones = pl->rank_d && D <= ld_ctx - 1;
  pl->dp = pl->rank_d ? ld_ctx : pl->dhp;
  HN_REQUIRE(p->dim_head_valid >= 0 && p->dim_head_valid <= p->dim_head && p->query_dim_valid >= 0 && p->query_dim_valid <= p->query_dim,
             HN_E_SHAPE, "attn: staged widths dim_head_valid=%d query_dim_valid=%d", p->dim_head_valid, p->query_dim_valid);
  pl->cscale = 2.0f * (1.0f / sqrtf((float)dh_valid(p))) * 1.44269504088896340736f;  // (1/0.5) * dh^-1/2 * log2(e)
  pl->bf16core = bf16core != 0 && pl->ones && pl->N > 1;
  pl->nq = (pl->rank_d && !pl->bf16core && p->dropout == 0.0f) ? attn_core_nq_small_batch(pl->dp, b, p->heads, pl->Lp) : 0;
  attn_core_geometry(b, p->heads, pl->Lp, pl->N, pl->dp, &pl->nsplit, &pl->chunk, 0, pl->nq);
  // (the explicit dp = 64 binding runs its dQ on the LDS ring, attention_lds.hip: 240 VGPRs = two resident waves per SIMD)
  attn_core_geometry(b, p->heads, pl->Lp, pl->N, pl->dp, &pl->nsplit_bwd, &pl->chunk_bwd, (!pl->rank_d && pl->dp == 64 && pl->N >= 256) ? 2 : 3);
  if (pl->bf16core) {
    // the plain dp = 16 bf16 core holds 161 VGPRs = 3 resident waves per SIMD: size the split for 3 (a split sized for 4 runs
    // a second, mostly idle round).  The larger variants measured faster with the default split (cfg3: 7.6 vs 11.2 ms).
    static const int w16 = tuning_env("HN_BF16_WAVES16") ? atoi(tuning_env("HN_BF16_WAVES16")) : 3;      // development knob
    if (pl->dp == 16 && bf16core == 1) attn_core_geometry(b, p->heads, pl->Lp, pl->N, pl->dp, &pl->nsplit, &pl->chunk, w16);
    pl->chunk = (pl->chunk + 31) / 32 * 32;
    pl->nsplit = (pl->N + pl->chunk - 1) / pl->chunk;
  }

  Arena ar(ws, ws_bytes);
  const size_t rows = rows16((size_t)b * L);
  pl->obuf = ar.take<float>(rows * pl->inner);
  if (pl->rank_d) {
    pl->q = ar.take<float>(rows * pl->inner);
    pl->qf = ar.take<float>((size_t)b * p->heads * pl->Lp * (pl->bf16core ? 48 : pl->dp));   // bf16 core: up to 96 bf16 slots per row
    pl->kv = nullptr;
    pl->bound = ar.take<float>((size_t)b * p->heads * pl->Lp + 64);      // per-row score bounds + the fallback flag
  } else {
    pl->bound = nullptr;
    pl->q = ar.take<float>(rows * p->heads * pl->dhp);
    pl->qf = nullptr;
    pl->kv = ar.take<float>((size_t)b * pl->N * 2 * p->heads * pl->dhp);
  }
  pl->ctx16 = nullptr;
  {
    const size_t w16 = gemm_bf16_stage_floats(2 * pl->inner, pl->D), w32 = gemm_nt_stage_floats(2 * p->heads * pl->dhp, pl->D);
    pl->wstage = (has_ctx && !pl->rank_d && pl->N > 1) ? ar.take<float>(w16 > w32 ? w16 : w32) : nullptr;
  }
  const size_t prow = (size_t)b * p->heads * pl->nsplit * pl->Lp;
  pl->opart = ar.take<float>(prow * pl->dp);
  pl->mpart = ar.take<float>(prow);
  pl->lpart = ar.take<float>(prow);
  pl->bytes = ar.off;
  if (ws != nullptr && ar.overflow)
    return fail(HN_E_WORKSPACE, "attn: workspace %zu bytes < required %zu", ws_bytes, ar.off);
  return HN_OK;
}

// Hooks of the fused latent chain (chain.hip) into an attention block: the block's projections were already produced by the
// chain in front of it (q / kv given, *_done), and / or its out-projection + residual is left to the chain behind it
// (defer_out: the merged attention output stays in the block's O buffer, reported through o_out / ldo_out).
struct AttnExt {
  float *q, *kv;          // external projection buffers (NULL: the plan's own)
  bool q_done, kv_done;
  bool defer_out;
  const float *o_out; int ldo_out;
  // shared-context (rank-D) block in front of a chain that can merge the split partials itself (ChainArgs.head == 3): the
  // block stops behind its core and reports the partials instead of running merge_vproj_kernel
  bool allow_defer_merge;
  bool merge_deferred;
  const float *opart, *mpart, *lpart;
  int nsplit, Lp, dp;
  // training: the block's projections live in the tape (sized from THIS block's plan) -- produced there by the chain in front
  // (q / kv above point at the same slots) or by the block's own GEMMs -- so that the backward does not recompute them
  float *q_home, *kv_home;
  // inference, rank-D block: the chain in front has written the FOLDED query and its score bounds (ChainArgs.qf): neither the
  // query projection nor qfold runs
  float *qf, *qf_bound; bool qf_done;
  // one-token shortcut with defer_out: the block's output row per sample (b, query_dim), before the broadcast add (ChainArgs.y)
  const float *y_out;
};

static int check_ws(void *ws, size_t ws_bytes, size_t need, const char *who) {
  HN_REQUIRE(ws != nullptr, HN_E_WORKSPACE, "%s: workspace is NULL (need %zu bytes)", who, need);
  HN_REQUIRE(((uintptr_t)ws & 255) == 0, HN_E_WORKSPACE, "%s: workspace must be 256-byte aligned", who);
  HN_REQUIRE(ws_bytes >= need, HN_E_WORKSPACE, "%s: workspace %zu bytes < required %zu", who, ws_bytes, need);
  return HN_OK;
}

static GemmArgs gemm_defaults() {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.batch = 1;
  g.alpha = 1.0f;
  g.eps = 1e-5f;
  return g;
}

// Steps shared by hn_attn_fwd and hn_attn_probs: the scaled query operand of the attention core and
// (explicit path) the projected keys / values.
static bool drop_bound_disabled() {      // development switch: dropout on the shared-context binding through the general core
  static const bool off = getenv("HN_NO_DROP_BOUND") != nullptr;
  return off;
}
static float *saved_kv(const AttnPlan &pl, bool has_ctx, bool masked, int b, int L, float *saved) {
  if (!saved || !has_ctx || pl.rank_d || (pl.N == 1 && !masked)) return nullptr;
  return saved + align_up(rows16((size_t)b * L) * pl.inner, 64);
}

// K | V = affine(ctx) W_kv^T of an explicit cross binding into `kvbuf` (pitch 2 heads dhp; pad columns of a padded head width are
// zero): the bf16 product on the bf16 context image (inference, core_precision = bf16), the LDS-DMA fp32 product on the staged
// weight (patch bags), or the generic GEMM.  `wstage`: the plan's staging scratch.
static int project_ctx_kv(const hn_attn_params *p, const AttnPlan &pl, const float *ctx, int ld_ctx, int b, float *kvbuf, float *wstage,
                          const uint16_t *ctx16, hipStream_t s) {
  const int kvpitch = 2 * p->heads * pl.dhp;
  static const bool no_glds = tuning_env("HN_NO_GLDS_GEMM") != nullptr;      // development switch: gemm_big_kernel / gemm_tall_narrow
  // the LDS-DMA projection lays the padded head width out itself (pad columns = 0): no fill in front of it
  const bool kv_nt = !ctx16 && wstage && !no_glds &&
                     gemm_nt_eligible((long)b * pl.N, 2 * pl.inner, pl.D, ld_ctx, ctx, pl.dh, pl.dhp, kvpitch, kvbuf);
  int rc;
  if (pl.dhp != pl.dh && !kv_nt && (rc = launch_fill(kvbuf, 0.0f, (long)((size_t)b * pl.N * kvpitch), s)) != HN_OK) return rc;
  GemmArgs gk = gemm_defaults();
  gk.A = ctx; gk.lda = ld_ctx; gk.M = b * pl.N; gk.K = pl.D;
  if (p->ctx_gamma) { gk.pro = PRO_AFFINE; gk.gamma = p->ctx_gamma; gk.beta = p->ctx_beta; }
  gk.W = p->w_kv; gk.ldw = pl.D;
  gk.N = 2 * pl.inner;
  gk.C = kvbuf; gk.ldc = kvpitch; gk.col_group = pl.dh; gk.col_group_pitch = pl.dhp;
  if (ctx16 && wstage && gemm_bf16_eligible(gk)) return launch_gemm_bf16(gk, ctx16, wstage, s);
  if (wstage && !no_glds && gemm_nt_eligible(gk.M, gk.N, gk.K, gk.lda, gk.A, gk.col_group, gk.col_group_pitch, gk.ldc, gk.C)) {
    // patch-bag K/V projection: LayerNorm affine folded into the staged weight, operands by LDS-DMA (gemm_nt.hip); a padded head
    // width is laid out by the staging, so the product writes dense rows (pad columns = 0) and needs no fill in front
    const int np = gemm_nt_padded_cols(gk.N, gk.col_group, gk.col_group_pitch);
    float *ws_w = wstage, *ws_b = wstage + (size_t)np * gemm_nt_ldws(gk.K);
    if ((rc = launch_gemm_nt_stage(gk.W, gk.ldw, gk.pro == PRO_AFFINE ? gk.gamma : nullptr, gk.pro == PRO_AFFINE ? gk.beta : nullptr, nullptr,
                                   gk.N, gk.K, ws_w, ws_b, s, gk.col_group, gk.col_group_pitch)) != HN_OK) return rc;
    GemmNtArgs gn;
    gn.A = gk.A; gn.lda = gk.lda; gn.W = ws_w; gn.ldw = gemm_nt_ldws(gk.K); gn.bias = ws_b; gn.C = gk.C; gn.ldc = gk.ldc;
    gn.M = gk.M; gn.N = np; gn.K = gk.K; gn.alpha = 1.0f; gn.col_group = 0; gn.col_group_pitch = 0;
    gn.ntm = gn.ntn = 0;
    return launch_gemm_nt(gn, 0, s);
  }
  return launch_gemm(gk, s);
}

// A shared-context block can take its query FOLDED and packed from the chain in front (ChainArgs.qf) exactly when its core is the
// bounded packed one on 16-column rows.  ONE predicate for the producer (add_next_proj of the inference forward, which then projects
// 128 folded columns and never produces the plain Q) and the consumer (attn_prepare, which has no route back once that happened):
// the two copies of these conditions agreed, but nothing made them.
static bool qfold_core_ok(const hn_attn_params *p, const AttnPlan &pl, int pack_ks, int L) {
  return pl.rank_d && pl.ones && p->ctx_gamma != nullptr && pl.dp == 16 && pl.Lp == L && pack_ks == packed_steps(pl.D, pl.dp);
}

// kv_tape (explicit cross binding, training): the projected K / V live in the tape instead of the workspace; the forward
// writes them there, the backward (kv_ready) reads them back instead of re-running the K/V projection GEMM.
static int attn_prepare(const hn_attn_params *p, const AttnPlan &pl, const float *x_in, const float *ctx, int ld_ctx,
                        int b, int L, hipStream_t s, AttnCoreArgs *core, int pack_ks = 0, float *kv_tape = nullptr,
                        bool kv_ready = false, bool use_bound = false, int *ext_flag = nullptr, const AttnExt *ext = nullptr) {
  const int rows = b * L;
  // external projection buffers are used only when the chain in front has actually produced them: they are sized for the
  // chain's own products (ckv: latent self-attention only), not for this block's plan (an explicit cross block
  // with N >> l_c would overrun ckv; padded head dims would leave the external q's pad columns unwritten)
  const bool q_done = ext && ext->q && ext->q_done;
  float *qbuf = q_done ? ext->q : ((ext && ext->q_home) ? ext->q_home : pl.q);
  GemmArgs gq = gemm_defaults();
  gq.A = x_in; gq.lda = p->query_dim;
  gq.W = p->w_q; gq.ldw = p->query_dim;
  gq.M = rows; gq.N = pl.inner; gq.K = p->query_dim;
  if (p->norm_w) { gq.pro = PRO_LAYERNORM; gq.gamma = p->norm_w; gq.beta = p->norm_b; }
  // (a staged block's projections come from the latent chain, whose LayerNorm knows the valid width; the GEMM prologue does not)
  HN_REQUIRE(!narrow_ln(p) || (q_done && (ctx != nullptr || kv_ready || (ext && ext->kv && ext->kv_done))), HN_E_UNSUPPORTED,
             "attn: a staged block (query_dim_valid=%d of %d) takes its projections from the latent chain", p->query_dim_valid, p->query_dim);
  memset(core, 0, sizeof(*core));
  core->b = b; core->h = p->heads; core->Lq = L; core->Lp = pl.Lp; core->N = pl.N; core->dp = pl.dp;
  core->nsplit = pl.nsplit; core->chunk = pl.chunk; core->nq = pl.nq;
  core->Opart = pl.opart; core->Mpart = pl.mpart; core->Lpart = pl.lpart;
  int rc;
  if (pl.rank_d && ext && ext->qf_done) {
    HN_REQUIRE(qfold_core_ok(p, pl, pack_ks, L) && use_bound && ext_flag, HN_E_SHAPE,
               "attn: folded query from the chain needs the bounded packed core (dp=%d Lp=%d)", pl.dp, pl.Lp);
    core->bound = ext->qf_bound; core->bound_flag = ext_flag;
    core->qk_steps = pack_ks;
    core->Q = ext->qf; core->q_b = (long)p->heads * pl.Lp * pl.dp; core->q_h = (long)pl.Lp * pl.dp; core->ldq = pl.dp;
    core->Kp = ctx; core->k_b = (long)pl.N * ld_ctx; core->k_h = 0; core->ldk = ld_ctx;
    core->Vp = ctx; core->v_b = core->k_b; core->v_h = 0; core->ldv = ld_ctx;
    core->ones_col = 1;
  } else if (pl.rank_d) {
    gq.C = qbuf; gq.ldc = pl.inner;
    if (!q_done && (rc = launch_gemm(gq, s)) != HN_OK) return rc;
    // score bounds need |z|^2 <= D, i.e. a context that went through the LayerNorm of PreNorm.norm_context (ctx_gamma set)
    float *bound = (pl.ones && p->ctx_gamma && use_bound) ? pl.bound : nullptr;
    // the fallback flag: the caller's pre-zeroed one (hn_fusion_forward zeroes all of a forward's flags in one launch) or ours
    int *bflag = bound ? (ext_flag ? ext_flag : (int *)(pl.bound + (size_t)b * p->heads * pl.Lp)) : nullptr;
    if (bound && !ext_flag && (rc = launch_fill((float *)bflag, 0.0f, 1, s)) != HN_OK) return rc;
    if ((rc = launch_qfold(qbuf, pl.inner, p->w_kv, pl.D, p->ctx_gamma, pl.cscale, pl.qf, b, p->heads, L, pl.Lp, pl.dh,
                           pl.dp, s, pack_ks, bound, bflag)) != HN_OK) return rc;
    core->bound = bound; core->bound_flag = bflag;
    core->qk_steps = pack_ks;
    core->Q = pl.qf; core->q_b = (long)p->heads * pl.Lp * pl.dp; core->q_h = (long)pl.Lp * pl.dp; core->ldq = pl.dp;
    core->Kp = ctx; core->k_b = (long)pl.N * ld_ctx; core->k_h = 0; core->ldk = ld_ctx;
    core->Vp = ctx; core->v_b = core->k_b; core->v_h = 0; core->ldv = ld_ctx;
    core->ones_col = pl.ones ? 1 : 0;
  } else {
    const int qpitch = p->heads * pl.dhp, kvpitch = 2 * p->heads * pl.dhp;
    const bool kv_ext = ext && ext->kv && ext->kv_done && !ctx;      // the chain projects K/V for latent self-attention only
    HN_REQUIRE(!(ext && ext->kv_done) || kv_ext, HN_E_SHAPE, "attn: external K/V projections exist for latent self-attention only");
    float *kvbuf = kv_tape ? kv_tape : (kv_ext ? ext->kv : ((ext && ext->kv_home && !ctx) ? ext->kv_home : pl.kv));
    if (kv_ext) kv_ready = true;
    if (pl.dhp != pl.dh) {
      // (projections found ready with a padded head width were written by this block's own forward into its tape slot, pad
      // columns included: the chain only projects for dim_head in {16, 32, 64, 128})
      if (!q_done) { int rc_ = launch_fill(qbuf, 0.0f, (long)((size_t)rows * qpitch), s); if (rc_ != HN_OK) return rc_; }
      if (!kv_ready && !ctx) { int rc_ = launch_fill(kvbuf, 0.0f, (long)((size_t)b * pl.N * kvpitch), s); if (rc_ != HN_OK) return rc_; }
    }
    gq.C = qbuf; gq.ldc = qpitch; gq.alpha = pl.cscale;
    gq.col_group = pl.dh; gq.col_group_pitch = pl.dhp;
    // latent self-attention: the K/V projection reads the same LayerNorm-ed x as the query projection -> one launch for both
    const bool fused_kv = !ctx && !kv_ready && !q_done;
    if (fused_kv) {
      gq.W2 = p->w_kv; gq.C2 = kvbuf; gq.ldc2 = kvpitch; gq.N2 = 2 * pl.inner; gq.alpha2 = 1.0f;
      gq.col_group2 = pl.dh; gq.col_group_pitch2 = pl.dhp;
    }
    if (!q_done && (rc = launch_gemm(gq, s)) != HN_OK) return rc;
    if (!kv_ready && ctx) {
      if ((rc = project_ctx_kv(p, pl, ctx, ld_ctx, b, kvbuf, pl.wstage, pl.ctx16, s)) != HN_OK) return rc;
    } else if (!kv_ready && !fused_kv) {   // self-attention behind a chain that projected Q only: context = normalised x (healnet.py:404)
      GemmArgs gk = gemm_defaults();
      gk.A = x_in; gk.lda = p->query_dim; gk.M = rows; gk.K = p->query_dim;
      if (p->norm_w) { gk.pro = PRO_LAYERNORM; gk.gamma = p->norm_w; gk.beta = p->norm_b; }
      gk.W = p->w_kv; gk.ldw = pl.D;
      gk.N = 2 * pl.inner;
      gk.C = kvbuf; gk.ldc = kvpitch; gk.col_group = pl.dh; gk.col_group_pitch = pl.dhp;
      if ((rc = launch_gemm(gk, s)) != HN_OK) return rc;
    }
    core->Q = qbuf; core->q_b = (long)L * qpitch; core->q_h = pl.dhp; core->ldq = qpitch;
    core->Kp = kvbuf; core->k_b = (long)pl.N * kvpitch; core->k_h = pl.dhp; core->ldk = kvpitch;
    core->Vp = kvbuf + (long)p->heads * pl.dhp; core->v_b = core->k_b; core->v_h = pl.dhp; core->ldv = kvpitch;
  }
  return HN_OK;
}

static int attn_fwd_impl(const hn_attn_params *p, const float *x_in, float *x_out, int residual, const float *ctx,
                         int ld_ctx, int b, int L, int N, int D, const uint8_t *mask, float *stats, void *ws,
                         size_t ws_bytes, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, float *o_save = nullptr,
                         bool ctx_has_ones = false, int ctx_pack_ks = 0, const Bf16Context *bc = nullptr, int *bound_flag = nullptr,
                         AttnExt *ext = nullptr, const uint16_t *ctx16 = nullptr) {
  HN_REQUIRE(x_in && (x_out || (ext && ext->defer_out)), HN_E_NULL, "attn: x is NULL");
  HN_REQUIRE(p && p->w_q && p->w_kv && p->w_out, HN_E_NULL, "attn: weight pointer is NULL");
  AttnPlan pl;
  int rc = plan_attn(p, ctx != nullptr, ld_ctx, b, L, N, D, nullptr, 0, &pl, bc ? bc->ns : 0);
  if (rc != HN_OK) return rc;
  if ((rc = check_ws(ws, ws_bytes, pl.bytes, "attn")) != HN_OK) return rc;
  if ((rc = plan_attn(p, ctx != nullptr, ld_ctx, b, L, N, D, ws, ws_bytes, &pl, bc ? bc->ns : 0)) != HN_OK) return rc;
  pl.ctx16 = o_save == nullptr ? ctx16 : nullptr;      // (training keeps the fp32 projection: the backward differentiates THAT product)

  // ---- one-token context without a mask (tabular / omic modality): softmax over a single key is exactly 1, so the
  // block reduces to y = LeakyReLU(W_out (W_v c) + b_out) broadcast over the latent rows; Q and K are dead
  // (SURVEY.md Appendix A-7).  Two skinny GEMMs on b rows instead of the b*L-row pipeline.
  // dropout on the probabilities: training entry points only (o_save != NULL); needs the general path (explicit
  // denominator, no one-token shortcut)
  const bool dropping = o_save != nullptr && p->dropout > 0.0f;
  HN_REQUIRE(p->dropout >= 0.0f && p->dropout < 1.0f, HN_E_SHAPE, "attn: dropout=%g", (double)p->dropout);
  if (ctx != nullptr && pl.N == 1 && mask == nullptr && !dropping) {
    float *vbuf = pl.q, *ybuf = pl.obuf;                   // (b, inner), (b, query_dim)
    GemmArgs gv = gemm_defaults();
    gv.A = ctx; gv.lda = ld_ctx; gv.M = b; gv.K = pl.D;
    if (p->ctx_gamma) { gv.pro = PRO_AFFINE; gv.gamma = p->ctx_gamma; gv.beta = p->ctx_beta; }
    gv.W = p->w_kv + (long)pl.inner * pl.D; gv.ldw = pl.D; gv.N = pl.inner;
    if (o_save) vbuf = o_save;                              // training: V straight into its tape slot (no copy behind the product)
    gv.C = vbuf; gv.ldc = pl.inner;
    if ((rc = launch_gemm(gv, s)) != HN_OK) return rc;
    GemmArgs gy = gemm_defaults();
    gy.A = vbuf; gy.lda = pl.inner; gy.M = b; gy.K = pl.inner;
    gy.W = p->w_out; gy.ldw = wo_ld(p); gy.N = p->query_dim;
    gy.bias = p->b_out; gy.act = ACT_LEAKY;
    gy.C = ybuf; gy.ldc = p->query_dim;
    if ((rc = launch_gemm(gy, s)) != HN_OK) return rc;
    // (`stats` is not written on this path: p == 1 for every row, and hn_attn_probs / hn_attn_importance / the backward
    // special-case a one-token context without a mask instead of reading it)
    if (ext && ext->defer_out) { ext->y_out = ybuf; return HN_OK; }      // the chain behind the block adds the row (ChainArgs.head == 2)
    return launch_add_row_broadcast(ybuf, residual ? x_in : nullptr, x_out, b, L, p->query_dim, s);
  }

  if (pl.bf16core) {
    // ---- bf16-MFMA core on the bf16 context images (inference only): q = LN(x) W_q^T, fold with W_k, core, merge
    HN_REQUIRE(bc && bc->zb && bc->zT && bc->DV == pl.dp && o_save == nullptr, HN_E_SHAPE, "attn: bf16 context images missing");
    GemmArgs gq = gemm_defaults();
    gq.A = x_in; gq.lda = p->query_dim;
    gq.W = p->w_q; gq.ldw = p->query_dim;
    gq.M = b * L; gq.N = pl.inner; gq.K = p->query_dim;
    if (p->norm_w) { gq.pro = PRO_LAYERNORM; gq.gamma = p->norm_w; gq.beta = p->norm_b; }
    const bool q_ext = ext && ext->q && ext->q_done;
    float *qraw = q_ext ? ext->q : pl.q;
    gq.C = qraw; gq.ldc = pl.inner;
    if (!q_ext && (rc = launch_gemm(gq, s)) != HN_OK) return rc;
    uint16_t *qfb = (uint16_t *)pl.qf;
    float *bound = p->ctx_gamma ? pl.bound : nullptr;
    int *bflag = bound ? (bound_flag ? bound_flag : (int *)(pl.bound + (size_t)b * p->heads * pl.Lp)) : nullptr;
    if (bound && !bound_flag && (rc = launch_fill((float *)bflag, 0.0f, 1, s)) != HN_OK) return rc;
    if ((rc = launch_qfold_bf16(qraw, pl.inner, p->w_kv, pl.D, p->ctx_gamma, pl.cscale, qfb, b, p->heads, L, pl.Lp, pl.dh, bc->DV, bc->ns, s,
                                bound, bflag)) != HN_OK)
      return rc;
    AttnCoreBf16Args ca;
    memset(&ca, 0, sizeof(ca));
    ca.Qf = qfb; ca.zb = bc->zb; ca.zT = bc->zT; ca.mask = mask;
    ca.Opart = pl.opart; ca.Mpart = pl.mpart; ca.Lpart = pl.lpart;
    ca.b = b; ca.h = p->heads; ca.Lq = L; ca.Lp = pl.Lp; ca.N = pl.N; ca.Np = bc->Np; ca.DV = bc->DV;
    ca.nsplit = pl.nsplit; ca.chunk = pl.chunk; ca.ns = bc->ns;
    ca.bound = bound; ca.bound_flag = bflag;
    if (ev0) HN_HIP_CHECK(hipEventRecord(ev0, s));
    if ((rc = launch_attn_core_bf16(ca, s)) != HN_OK) return rc;
    if (ev1) HN_HIP_CHECK(hipEventRecord(ev1, s));
    if ((rc = launch_merge_vproj(pl.opart, pl.mpart, pl.lpart, pl.nsplit, b, p->heads, L, pl.Lp, pl.dp, pl.D, p->ctx_gamma,
                                 p->ctx_beta, p->w_kv + (long)pl.inner * pl.D, pl.dh, pl.obuf, pl.inner, stats, nullptr, s, 0)) != HN_OK)
      return rc;
    if (ext && ext->defer_out) { ext->o_out = pl.obuf; ext->ldo_out = pl.inner; return HN_OK; }
    GemmArgs go = gemm_defaults();
    go.A = pl.obuf; go.lda = pl.inner;
    go.W = p->w_out; go.ldw = wo_ld(p);
    go.M = b * L; go.N = p->query_dim; go.K = pl.inner;
    go.bias = p->b_out; go.act = ACT_LEAKY;
    if (residual) { go.R = x_in; go.ldr = p->query_dim; }
    go.C = x_out; go.ldc = p->query_dim;
    return launch_gemm(go, s);
  }

  // ---- explicit binding of a large patch bag under core_precision = bf16 (inference): K / V projected on bf16 MFMA straight into
  // the bf16 images of the explicit bf16 core (heads of 64); otherwise the projection alone (fp32 rows, attn_prepare)
  static const bool no_expl16 = tuning_env("HN_NO_BF16_EXPL_CORE") != nullptr;      // development switch
  if (pl.ctx16 && pl.wstage && ctx && !pl.rank_d && !dropping && !no_expl16 && pl.dh == 64 && pl.dhp == 64 && p->heads % 2 == 0 && !narrow_ln(p)) {
    GemmArgs gk = gemm_defaults();
    gk.A = ctx; gk.lda = ld_ctx; gk.M = b * pl.N; gk.K = pl.D;
    if (p->ctx_gamma) { gk.pro = PRO_AFFINE; gk.gamma = p->ctx_gamma; gk.beta = p->ctx_beta; }
    gk.W = p->w_kv; gk.ldw = pl.D;
    gk.N = 2 * pl.inner;
    gk.C = pl.kv; gk.ldc = 2 * pl.inner;      // (not written: the images below alias it)
    // the fp32 K|V rows of this plan (b N x 2 inner floats) hold K16 (a quarter), V16 (a quarter; token slots rounded up to 32 per
    // sample) and the query image: a bag too small for that (few tokens against l_c query rows) keeps the fp32 core behind the
    // bf16 projection (attn_prepare) instead of failing
    const int Np = (pl.N + 31) / 32 * 32;
    const bool images_fit = align_up((size_t)b * pl.N * pl.inner, 8) + (size_t)b * Np * pl.inner + (size_t)b * p->heads * pl.Lp * 64 <=
                            (size_t)b * pl.N * pl.inner * 4;
    if (images_fit && gemm_bf16_eligible(gk)) {
      uint16_t *K16 = (uint16_t *)pl.kv;
      uint16_t *V16 = K16 + align_up((size_t)b * pl.N * pl.inner, 8);
      uint16_t *Q16 = V16 + (size_t)b * Np * pl.inner;
      const bool q_done = ext && ext->q && ext->q_done;
      float *qbuf = q_done ? ext->q : pl.q;
      if (!q_done) {
        GemmArgs gq = gemm_defaults();
        gq.A = x_in; gq.lda = p->query_dim;
        gq.W = p->w_q; gq.ldw = p->query_dim;
        gq.M = b * L; gq.N = pl.inner; gq.K = p->query_dim;
        if (p->norm_w) { gq.pro = PRO_LAYERNORM; gq.gamma = p->norm_w; gq.beta = p->norm_b; }
        gq.C = qbuf; gq.ldc = pl.inner; gq.alpha = pl.cscale;
        if ((rc = launch_gemm(gq, s)) != HN_OK) return rc;
      }
      if ((rc = launch_q_rows_to_bf16(qbuf, pl.inner, b, p->heads, L, pl.Lp, Q16, s)) != HN_OK) return rc;
      if ((rc = launch_gemm_bf16(gk, pl.ctx16, pl.wstage, s, K16, V16, pl.N)) != HN_OK) return rc;
      AttnCoreBf16Args ca;
      memset(&ca, 0, sizeof(ca));
      ca.Qf = Q16; ca.zb = K16; ca.zT = V16; ca.mask = mask;
      ca.Opart = pl.opart; ca.Mpart = pl.mpart; ca.Lpart = pl.lpart;
      ca.b = b; ca.h = p->heads; ca.Lq = L; ca.Lp = pl.Lp; ca.N = pl.N; ca.Np = Np; ca.DV = 64;
      // splits of >= 512 tokens (never more splits than the plan's, which sized the partial buffers): the sweep over {plan, 512, 1024,
      // 2048} x {2, 4 query tiles per wave} at cfg4 / cfg5 picked 512 -- 1024 (0.78 / 5.64 ms against 0.80 / 5.87 with the fp32 core's split)
      ca.chunk = (pl.chunk + 31) / 32 * 32;
      if (ca.chunk < 512) ca.chunk = 512;
      static const int chunk_knob = tuning_env("HN_BF16_EXPL_CHUNK") ? atoi(tuning_env("HN_BF16_EXPL_CHUNK")) : 0;      // development knob: coarser splits
      if (chunk_knob > ca.chunk) ca.chunk = (chunk_knob + 31) / 32 * 32;
      ca.nsplit = (pl.N + ca.chunk - 1) / ca.chunk;
      ca.ns = 1; ca.expl = 1; ca.k_pitch = pl.inner * 2;
      if (ev0) HN_HIP_CHECK(hipEventRecord(ev0, s));
      if ((rc = launch_attn_core_bf16(ca, s)) != HN_OK) return rc;
      if (ev1) HN_HIP_CHECK(hipEventRecord(ev1, s));
      if ((rc = launch_merge_explicit(pl.opart, pl.mpart, pl.lpart, ca.nsplit, b, p->heads, L, pl.Lp, 64, pl.dh, pl.obuf, pl.inner, stats,
                                      s)) != HN_OK) return rc;
      if (ext && ext->defer_out) { ext->o_out = pl.obuf; ext->ldo_out = pl.inner; return HN_OK; }
      GemmArgs go = gemm_defaults();
      go.A = pl.obuf; go.lda = pl.inner;
      go.W = p->w_out; go.ldw = wo_ld(p);
      go.C = x_out; go.ldc = p->query_dim;
      go.bias = p->b_out;
      go.M = b * L; go.N = p->query_dim; go.K = pl.inner;
      go.act = ACT_LEAKY;
      if (residual) { go.R = x_in; go.ldr = p->query_dim; }
      return launch_gemm(go, s);
    }
  }

  AttnCoreArgs core;
  const int pack_ks = (pl.rank_d && pl.ones && ctx_has_ones && p->ctx_gamma) ? ctx_pack_ks : 0;
  if ((rc = attn_prepare(p, pl, x_in, ctx, ld_ctx, b, L, s, &core, pack_ks,
                         saved_kv(pl, ctx != nullptr, mask != nullptr || dropping, b, L, o_save), false, !dropping || (pl.rank_d && pl.ones),
                         bound_flag, ext)) != HN_OK) return rc;
  core.mask = mask;
  core.ones_in_mem = (ctx_has_ones && pl.ones) ? 1 : 0;
  core.drop = drop_off();
  const int srow = (dropping && pl.rank_d) ? 1 : 0;       // the thinned probabilities' row sum rides in column dp-1
  HN_REQUIRE(!srow || pl.ones, HN_E_UNSUPPORTED, "attn: dropout on the shared-context binding needs a free column (D <= dp - 1)");
  if (dropping) {
    // shared-context binding with a score bound (LayerNorm-ed context): the bounded softmax stays, the ones column (in the
    // context rows of the training layout, else injected in registers) is the row-sum channel; otherwise the general path
    const bool keep_bound = srow && core.bound != nullptr && (pl.dp == 16 || pl.dp == 32) && !drop_bound_disabled();
    HN_REQUIRE(keep_bound || !core.ones_in_mem, HN_E_SHAPE, "attn: a context laid out with the ones column needs the bounded dropout core");
    core.ones_col = keep_bound ? 1 : 0; core.drop = drop_of(p->dropout, p->rng, false); core.drop_rowsum = srow;
    if (!keep_bound) { core.bound = nullptr; core.bound_flag = nullptr; }
  }
  if (o_save && !pl.rank_d) pl.obuf = o_save;      // training, explicit binding: the merged O is produced straight in its tape slot
  const bool direct = !pl.rank_d && pl.nsplit == 1;
  if (direct) { core.Ofinal = pl.obuf; core.ldo = pl.inner; core.dh = pl.dh; core.stats = stats; }
  if (ev0) HN_HIP_CHECK(hipEventRecord(ev0, s));
  if ((rc = launch_attn_core(core, s)) != HN_OK) return rc;
  if (ev1) HN_HIP_CHECK(hipEventRecord(ev1, s));
  if (pl.rank_d && ext && ext->defer_out && ext->allow_defer_merge && !o_save && !dropping && pl.ones && pl.dp == 16 &&
      pl.nsplit <= CHAIN_MERGE_MAX_SPLITS && p->heads <= 8 && (pl.dh == 16 || pl.dh == 32 || pl.dh == 64) && L % 16 == 0) {
    ext->merge_deferred = true;
    ext->opart = pl.opart; ext->mpart = pl.mpart; ext->lpart = pl.lpart;
    ext->nsplit = pl.nsplit; ext->Lp = pl.Lp; ext->dp = pl.dp;
    return HN_OK;
  }
  if (pl.rank_d) {
    rc = launch_merge_vproj(pl.opart, pl.mpart, pl.lpart, pl.nsplit, b, p->heads, L, pl.Lp, pl.dp, pl.D, p->ctx_gamma,
                            p->ctx_beta, p->w_kv + (long)pl.inner * pl.D, pl.dh, pl.obuf, pl.inner, stats, o_save, s, pack_ks, srow);
  } else if (!direct) {
    rc = launch_merge_explicit(pl.opart, pl.mpart, pl.lpart, pl.nsplit, b, p->heads, L, pl.Lp, pl.dp, pl.dh, pl.obuf,
                               pl.inner, stats, s);
  }
  if (rc != HN_OK) return rc;
  if (ext && ext->defer_out) { ext->o_out = pl.obuf; ext->ldo_out = pl.inner; return HN_OK; }

  GemmArgs go = gemm_defaults();
  go.A = pl.obuf; go.lda = pl.inner;
  go.W = p->w_out; go.ldw = wo_ld(p);
  go.C = x_out; go.ldc = p->query_dim;
  go.bias = p->b_out;
  go.M = b * L; go.N = p->query_dim; go.K = pl.inner;
  go.act = ACT_LEAKY;
  if (residual) { go.R = x_in; go.ldr = p->query_dim; }
  return launch_gemm(go, s);
}

// o (b*L, inner) = sum_r w_r o_r / sum_r w_r,  w_r = 2^(M_r - M) l_r,  M = max_r M_r  per (sample, head, row); fixed order
template <int V>      // V columns per thread: 4 (dim_head % 4 == 0: 16-byte pieces) or 1
__global__ __launch_bounds__(256) void attn_merge_parts_kernel(const float *__restrict__ o_parts, const float *__restrict__ st_parts,
                                                               int n_parts, int b, int heads, int L, int dh, float *__restrict__ o,
                                                               float *__restrict__ st_out, long ostride, long sstride) {
  const int inner = heads * dh, pieces = inner / V;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)b * L * pieces) return;
  const long row = idx / pieces;
  const int c = (int)(idx - row * pieces) * V, h = c / dh;
  const int bi = (int)(row / L), q = (int)(row - (long)bi * L);
  const long srow = (((long)bi * heads + h) * L + q) * 2;
  float M = -3.0e38f;
  for (int r = 0; r < n_parts; ++r) M = fmaxf(M, st_parts[r * sstride + srow]);
  float wsum = 0.0f, acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.0f;
  for (int r = 0; r < n_parts; ++r) {
    const float w = exp2f(st_parts[r * sstride + srow] - M) * st_parts[r * sstride + srow + 1];
    if (!(w > 0.0f)) continue;        // a shard whose keys are all masked has l = 0 and a 0 / 0 output: weight zero, never read
                                      // (every shard dead: 0 / 0 = NaN, like the reference's softmax over a fully masked row)
    const float *src = &o_parts[r * ostride + row * inner + c];
    if (V == 4) {
      const f32x4 v = *(const f32x4 *)src;
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = fmaf(w, v[e], acc[e]);
    } else {
      acc[0] = fmaf(w, src[0], acc[0]);
    }
    wsum += w;
  }
  const float inv = 1.0f / wsum;
#pragma unroll
  for (int e = 0; e < V; ++e) o[row * inner + c + e] = acc[e] * inv;
  if (st_out && c == h * dh) { st_out[srow] = M; st_out[srow + 1] = wsum; }
}

// ------------------------------------------------------------------------------------------------
// attention block, backward
// ------------------------------------------------------------------------------------------------
// What the training forward keeps per attention block besides the softmax statistics:
//   explicit K/V binding : O (b*L, inner), the normalised attention output
//   rank-D binding       : P z (b*L, heads*dp), the normalised context average (O is recomputed from it)
//   one-token context    : V (b, inner)
static size_t attn_saved_floats(const AttnPlan &pl, bool has_ctx, bool masked, int b, int L) {
  if (has_ctx && pl.N == 1 && !masked) return (size_t)b * pl.inner;
  if (pl.rank_d) return (size_t)b * L * pl.heads * pl.dp;
  // explicit binding: O, and for a cross block also the projected K / V (288 GB of HBM: keeping 134 MB per WSI-bag block
  // at cfg4 is cheaper than re-running its 52 GF projection in the backward)
  return align_up(rows16((size_t)b * L) * pl.inner, 64) + (has_ctx ? (size_t)b * pl.N * 2 * pl.heads * pl.dhp : 0);
}

struct AttnBwdPlan {
  float *dpre, *dO, *xhat, *dxhat, *lns, *delta, *dOp, *dQpart, *dQ, *dKV, *G, *cs, *Abuf, *dA, *E, *T, *dT, *dyb, *dV, *red;
  void *fwd_ws; size_t fwd_bytes, bytes;
};

static int plan_attn_bwd(const hn_attn_params *p, const AttnPlan &pl, bool has_ctx, bool masked, int b, int L, void *ws,
                         size_t ws_bytes, AttnBwdPlan *bp) {
  Arena ar(ws, ws_bytes);
  const size_t rows = rows16((size_t)b * L), qd = p->query_dim, inner = pl.inner, h = p->heads;
  memset(bp, 0, sizeof(*bp));
  bp->fwd_bytes = pl.bytes;
  bp->fwd_ws = ar.take<char>(pl.bytes);
  bp->dpre = ar.take<float>(rows * qd);
  {
    const long kdim = pl.D > (int)qd ? pl.D : (long)qd;
    bp->red = ar.take<float>(reduce_scratch_floats(2L * inner * kdim, (int)(2 * inner > qd ? 2 * inner : qd)));
  }
  if (has_ctx && pl.N == 1 && !masked) {
    bp->dyb = ar.take<float>((size_t)b * qd);
    bp->dV = ar.take<float>((size_t)b * inner);
    bp->G = ar.take<float>(inner * pl.D);
    bp->cs = ar.take<float>(inner);
  } else {
    bp->dO = ar.take<float>(rows * inner);
    bp->xhat = ar.take<float>(rows * qd);
    bp->dxhat = ar.take<float>(rows * qd);
    bp->lns = ar.take<float>(ln_bwd_scratch_floats(rows, (int)qd));
    bp->delta = ar.take<float>((size_t)b * h * L);
    bp->dQpart = ar.take<float>((size_t)b * h * pl.nsplit_bwd * pl.Lp * pl.dp);
    bp->dQ = ar.take<float>(rows * inner);
    if (pl.rank_d) {
      const size_t hp = rows * h * pl.dp;
      bp->Abuf = ar.take<float>(hp);
      bp->dA = ar.take<float>(hp);
      bp->dOp = ar.take<float>(hp);
      bp->E = ar.take<float>(hp);
      bp->T = ar.take<float>(hp);
      bp->dT = ar.take<float>(hp);
    } else {
      bp->dOp = ar.take<float>(rows * h * pl.dhp);
      bp->dKV = ar.take<float>((size_t)b * pl.N * 2 * inner);
      if (has_ctx) {
        bp->G = ar.take<float>(2 * inner * pl.D);
        bp->cs = ar.take<float>(2 * inner);
      }
    }
  }
  bp->bytes = ar.off;
  if (ws != nullptr && ar.overflow) return fail(HN_E_WORKSPACE, "attn_bwd: workspace %zu bytes < required %zu", ws_bytes, ar.off);
  return HN_OK;
}

static GemmExArgs gex(const float *A, long a_rs, long a_cs, const float *B, long b_rs, long b_cs, float *C, long ldc, int M,
                      int N, int K, int accumulate) {
  GemmExArgs e;
  memset(&e, 0, sizeof(e));
  e.A = A; e.a_rs = a_rs; e.a_cs = a_cs; e.B = B; e.b_rs = b_rs; e.b_cs = b_cs; e.C = C; e.ldc = ldc;
  e.M = M; e.N = N; e.K = K; e.batch = 1; e.alpha = 1.0f; e.accumulate = accumulate;
  return e;
}

// Hooks of the fused latent backward (bchain.hip) into an attention block's backward: the chain behind the block (in backward
// order: in front of it) has already produced dpre = dy * LeakyReLU'(.) and dO = dpre W_out, and / or the chain in front of it
// will run the projection backward (dx_hat = dQ W_q + dKV W_kv, LayerNorm backward, residual) and the batched weight-gradient
// launch takes dW_q / dW_kv (and dW_out when the block's O is on the tape).
struct AttnBwdExt {
  const float *dpre, *dO;    // given (rows, query_dim) / (rows, inner): the LeakyReLU backward and the dO product are skipped
  bool skip_wout;            // dW_out / db_out are left to the caller's batched launch (O = the tape's, explicit bindings only)
  bool defer_proj;           // stop behind the core: no dW_q / dW_kv, no dx; dQ / dKV / xhat are reported instead
  const float *dQ, *dKV, *xhat;   // out (defer_proj): (rows, inner) scaled, (rows, 2 inner) or NULL (cross blocks), LN(x_in) (rows, query_dim)
  const float *O;            // out: the block's attention output (rows, inner) -- the tape's or the recomputed one
  const float *q_taped, *kv_taped;      // in: the forward's projections from the tape (NULL: recomputed here)
  const float *xhat_taped;              // in: LN(x_in) from the tape (NULL: recomputed here)
  float *dQ_home, *dKV_home;            // in (defer_proj): where dQ (rows, inner) / dKV of a latent block (rows, 2 inner) are produced instead of
                                        // the op workspace -- they outlive the next block's backward (batched weight-gradient products)
  bool dx_without_residual;  // context split: `residual` tells where the sign of the pre-activation comes from (x_out - x_in), but dx receives
                             // the gradient through the queries only -- the residual term is replicated and added once, after the sum over ranks
  bool skip_replicated;      // context split (hn_attn_bwd_cp): the gradients that do NOT pass through the core backward -- dW_out / db_out, and for a
                             // shared-context block dW_v and the value side of the context LayerNorm affine -- are computed from replicated
                             // quantities only; every rank but the owner leaves them out, so that the sum over ranks counts them once
};

static int attn_bwd_impl(const hn_attn_params *p, const float *x_in, const float *x_out, int residual, const float *ctx,
                         int ld_ctx, int b, int L, int N, int D, const uint8_t *mask, const float *stats, const float *saved,
                         const float *dy, float *dx, const hn_attn_grads *g, void *ws, size_t ws_bytes, hipStream_t s,
                         int ctx_pack_ks = 0, AttnBwdExt *ext = nullptr) {
  HN_REQUIRE(p && x_in && x_out && stats && saved && dy && dx && g, HN_E_NULL, "attn_bwd: NULL pointer");
  HN_REQUIRE(p->w_q && p->w_kv && p->w_out, HN_E_NULL, "attn_bwd: weight pointer is NULL");
  const bool has_ctx = ctx != nullptr;
  const bool dropping = p->dropout > 0.0f;           // the forward that produced `saved` thinned its probabilities
  const bool general = mask != nullptr || dropping;   // ... and therefore took the general (not the one-token) path
  AttnPlan pl;
  int rc = plan_attn(p, has_ctx, ld_ctx, b, L, N, D, nullptr, 0, &pl);
  if (rc != HN_OK) return rc;
  AttnBwdPlan bp;
  if ((rc = plan_attn_bwd(p, pl, has_ctx, general, b, L, nullptr, 0, &bp)) != HN_OK) return rc;
  if ((rc = check_ws(ws, ws_bytes, bp.bytes, "attn_bwd")) != HN_OK) return rc;
  if ((rc = plan_attn_bwd(p, pl, has_ctx, general, b, L, ws, ws_bytes, &bp)) != HN_OK) return rc;
  if ((rc = plan_attn(p, has_ctx, ld_ctx, b, L, N, D, bp.fwd_ws, bp.fwd_bytes, &pl)) != HN_OK) return rc;
  if (ext && ext->defer_proj) {      // (the chain in front runs the projection backward: its operands may have to outlive this workspace)
    if (ext->dQ_home) bp.dQ = ext->dQ_home;
    if (ext->dKV_home && !has_ctx) bp.dKV = ext->dKV_home;
  }

  const int rows = b * L, qd = p->query_dim, inner = pl.inner, h = p->heads, dh = pl.dh;
  const float two_scale = 2.0f / sqrtf((float)dh_valid(p));
  // dpre = dy * LeakyReLU'(pre); the sign of pre is the sign of y = x_out - x_in
  const float *dpre = bp.dpre;
  const bool one_token = has_ctx && pl.N == 1 && !general;
  const bool one_token_fused = one_token && !ext && onetoken_bwd_fused_ok(b, qd);      // (forms dpre itself, row by row)
  if (ext && ext->dpre) dpre = ext->dpre;
  else if (!one_token_fused && (rc = launch_leaky_bwd(dy, x_out, residual ? x_in : nullptr, bp.dpre, (long)rows * qd, s)) != HN_OK) return rc;

  if (one_token_fused) {
    // ---- one-token context, three launches (backward.hip "Backward of the one-token cross block")
    if ((rc = launch_onetoken_bwd(dy, x_out, residual ? x_in : nullptr, b, L, qd, p->w_out, wo_ld(p), inner, saved, ctx, ld_ctx, pl.D,
                                  p->w_kv + (long)inner * pl.D, p->ctx_gamma, p->ctx_beta, bp.dyb, bp.dV, g->w_out, g->b_out,
                                  g->w_kv ? g->w_kv + (long)inner * pl.D : nullptr, g->ctx_gamma, g->ctx_beta, bp.red, s)) != HN_OK) return rc;
    if (residual) { if (dx != dy) return launch_add_into(dy, dx, (long)rows * qd, 0, s); return HN_OK; }
    { int rc_ = launch_fill(dx, 0.0f, (long)((size_t)rows * qd), s); if (rc_ != HN_OK) return rc_; }
    return HN_OK;
  }
  if (one_token) {
    HN_REQUIRE(!ext, HN_E_UNSUPPORTED, "attn_bwd: the one-token shortcut takes no chain hooks");   // ---- one-token context: y_b = LeakyReLU(W_out V_b + b_out) for every row
    if ((rc = launch_segsum(bp.dpre, L, qd, b, bp.dyb, s)) != HN_OK) return rc;
    if (g->b_out && (rc = launch_colsum(bp.dyb, qd, b, qd, 1.0f, g->b_out, 1, s, bp.red)) != HN_OK) return rc;
    if (g->w_out) {   // dWo += dyb^T V
      GemmExArgs e = gex(bp.dyb, 1, qd, saved, 1, inner, g->w_out, wo_ld(p), qd, inner, b, 1);
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
    }
    {   // dV = dyb Wo
      GemmExArgs e = gex(bp.dyb, qd, 1, p->w_out, 1, wo_ld(p), bp.dV, inner, b, inner, qd, 0);
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
    }
    {   // G = dV^T z, cs = colsum(dV)  -> gradients of the V half of to_kv and of the context LayerNorm affine
      GemmExArgs e = gex(bp.dV, 1, inner, ctx, 1, ld_ctx, bp.G, pl.D, inner, pl.D, b, 0);
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
      if ((rc = launch_colsum(bp.dV, inner, b, inner, 1.0f, bp.cs, 0, s, bp.red)) != HN_OK) return rc;
      if ((rc = launch_kv_weight_grads(bp.G, bp.cs, p->w_kv + (long)inner * pl.D, p->ctx_gamma, p->ctx_beta, inner, pl.D,
                                       g->w_kv ? g->w_kv + (long)inner * pl.D : nullptr, g->ctx_gamma, g->ctx_beta, s, bp.red)) != HN_OK)
        return rc;
    }
    if (residual) { if (dx != dy) return launch_add_into(dy, dx, (long)rows * qd, 0, s); return HN_OK; }
    { int rc_ = launch_fill(dx, 0.0f, (long)((size_t)rows * qd), s); if (rc_ != HN_OK) return rc_; }
    return HN_OK;
  }

  // ---- output projection: dWo += dpre^T O, dbo += colsum(dpre), dO = dpre Wo
  const float *O = saved;
  if (pl.rank_d) {   // O = (P z * gamma + beta) W_v^T is recomputed from the saved P z
    if (dropping) rc = launch_srow_affine(saved, nullptr, p->ctx_gamma, p->ctx_beta, 0, h, pl.D, pl.dp, rows, bp.Abuf, s);
    else rc = launch_head_affine(saved, h * pl.dp, pl.dp, nullptr, 0, 0, p->ctx_gamma, p->ctx_beta, 1.0f, h, pl.D, pl.dp,
                                 h * pl.dp, rows, bp.Abuf, s);
    if (rc != HN_OK) return rc;
    GemmExArgs e = gex(bp.Abuf, (long)h * pl.dp, 1, p->w_kv + (long)inner * pl.D, pl.D, 1, pl.obuf, inner, rows, dh, pl.D, 0);
    e.batch = h; e.strideA = pl.dp; e.strideB = (long)dh * pl.D; e.strideC = dh;
    if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
    O = pl.obuf;
  }
  if (ext) ext->O = O;
  const bool skip_repl = ext && ext->skip_replicated;
  if ((ext && ext->skip_wout) || skip_repl) {
    // dW_out / db_out: the caller's batched weight-gradient launch (or, context split: the owner rank's)
  } else if (g->w_out) {      // dWo += dpre^T O, and db_out += colsum(dpre) from the same pass over dpre
    GemmExArgs e = gex(dpre, 1, qd, O, 1, inner, g->w_out, wo_ld(p), qd, inner, rows, 1);
    e.colsum = g->b_out; e.colsum_accumulate = 1;
    if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
  } else if (g->b_out && (rc = launch_colsum(dpre, qd, rows, qd, 1.0f, g->b_out, 1, s, bp.red)) != HN_OK) return rc;
  const float *dO = bp.dO;
  if (ext && ext->dO) dO = ext->dO;
  else {
    GemmExArgs e = gex(dpre, qd, 1, p->w_out, 1, wo_ld(p), bp.dO, inner, rows, inner, qd, 0);
    if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
  }

  // ---- recompute the operands of the core (scaled Q, and K/V or the folded queries)
  AttnCoreArgs core;
  float *kv_saved = saved_kv(pl, has_ctx, general, b, L, const_cast<float *>(saved));
  // packed shared context (the training forward's layout, train_context_layout): folded queries, dO' and dQ'' in slot order
  const int pack_ks = (pl.rank_d && pl.ones && p->ctx_gamma) ? ctx_pack_ks : 0;
  float *kv_from = kv_saved ? kv_saved : ((ext && ext->kv_taped && !has_ctx) ? const_cast<float *>(ext->kv_taped) : nullptr);
  AttnExt pe;
  memset(&pe, 0, sizeof(pe));
  if (ext && ext->q_taped) { pe.q = const_cast<float *>(ext->q_taped); pe.q_done = true; }
  const float *qraw = (ext && ext->q_taped) ? ext->q_taped : pl.q;      // rank-D binding: Q before the fold
  if ((rc = attn_prepare(p, pl, x_in, ctx, ld_ctx, b, L, s, &core, pack_ks, kv_from, kv_from != nullptr, false, nullptr,
                         (ext && ext->q_taped) ? &pe : nullptr)) != HN_OK) return rc;
  const float *xhat = x_in;
  if (p->norm_w && ext && ext->xhat_taped) xhat = ext->xhat_taped;
  else if (p->norm_w) {
    if ((rc = launch_ln_fwd(x_in, p->norm_w, p->norm_b, rows, qd, bp.xhat, s, p->query_dim_valid)) != HN_OK) return rc;
    xhat = bp.xhat;
  }
  AttnBwdArgs ba;
  memset(&ba, 0, sizeof(ba));
  ba.Q = core.Q; ba.q_b = core.q_b; ba.q_h = core.q_h; ba.ldq = core.ldq;
  ba.Kp = core.Kp; ba.k_b = core.k_b; ba.k_h = core.k_h; ba.ldk = core.ldk;
  ba.Vp = core.Vp; ba.v_b = core.v_b; ba.v_h = core.v_h; ba.ldv = core.ldv;
  ba.mask = mask; ba.stats = stats; ba.delta = bp.delta; ba.dQpart = bp.dQpart;
  ba.b = b; ba.h = h; ba.Lq = L; ba.Lp = pl.Lp; ba.N = pl.N; ba.dp = pl.dp; ba.nsplit = pl.nsplit_bwd; ba.chunk = pl.chunk_bwd;
  ba.drop = dropping ? drop_of(p->dropout, p->rng, false) : drop_off();
  const bool srow = dropping && pl.rank_d;
  ba.drop_rowsum = srow ? 1 : 0;

  if (pl.rank_d) {
    const int hp = h * pl.dp;
    const float *wv = p->w_kv + (long)inner * pl.D, *wk = p->w_kv;
    float *dwv = g->w_kv ? g->w_kv + (long)inner * pl.D : nullptr, *dwk = g->w_kv;
    // O_h = A_h W_v,h^T with A = P z * gamma + beta:  dW_v,h += dO_h^T A_h ;  dA_h = dO_h W_v,h
    if (dwv && !skip_repl) {
      GemmExArgs e = gex(dO, 1, inner, bp.Abuf, 1, hp, dwv, pl.D, dh, pl.D, rows, 1);
      e.batch = h; e.strideA = dh; e.strideB = pl.dp; e.strideC = (long)dh * pl.D;
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
    }
    { int rc_ = launch_fill(bp.dA, 0.0f, (long)((size_t)rows * hp), s); if (rc_ != HN_OK) return rc_; }
    {
      GemmExArgs e = gex(dO, inner, 1, wv, 1, pl.D, bp.dA, hp, rows, pl.D, dh, 0);
      e.batch = h; e.strideA = dh; e.strideB = (long)dh * pl.D; e.strideC = pl.dp;
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
    }
    if (p->ctx_gamma && !skip_repl) {   // dgamma += sum dA * (P z) ; dbeta += sum dA   (over rows and heads)
      if ((rc = launch_head_affine(bp.dA, hp, pl.dp, saved, hp, pl.dp, nullptr, nullptr, 1.0f, h, pl.D, pl.dp, hp, rows, bp.E, s)) != HN_OK) return rc;
      if (g->ctx_gamma && (rc = launch_colsum(bp.E, pl.dp, (long)rows * h, pl.D, 1.0f, g->ctx_gamma, 1, s, bp.red)) != HN_OK) return rc;
      if (g->ctx_beta) {
        const float *src = bp.dA;
        if (srow) {   // d beta_c = sum dA_c * s
          if ((rc = launch_srow_affine(saved, bp.dA, nullptr, nullptr, 1, h, pl.D, pl.dp, rows, bp.E, s)) != HN_OK) return rc;
          src = bp.E;
        }
        if ((rc = launch_colsum(src, pl.dp, (long)rows * h, pl.D, 1.0f, g->ctx_beta, 1, s, bp.red)) != HN_OK) return rc;
      }
    }
    // d(P z) = dA * gamma (+ the row-sum channel under dropout) ;  delta = rowsum(d(P z) * P z)
    if (srow) rc = launch_srow_affine(saved, bp.dA, p->ctx_gamma, p->ctx_beta, 2, h, pl.D, pl.dp, rows, bp.dOp, s);
    else rc = launch_head_affine(bp.dA, hp, pl.dp, nullptr, 0, 0, p->ctx_gamma, nullptr, 1.0f, h, pl.D, pl.dp, hp, rows, bp.dOp, s);
    if (rc != HN_OK) return rc;
    if ((rc = launch_rowdot_heads(bp.dOp, hp, pl.dp, saved, hp, pl.dp, h, L, srow ? pl.dp : pl.D, rows, bp.delta, s)) != HN_OK) return rc;
    ba.dO = bp.dOp; ba.do_b = (long)L * hp; ba.do_h = pl.dp; ba.lddo = hp;
    ba.qk_steps = pack_ks;
    if (pack_ks && (rc = launch_pack_fold(bp.dOp, hp, h, pl.D, pl.dp, pack_ks, 0, rows, s, srow ? 1 : 0)) != HN_OK) return rc;
    if ((rc = launch_attn_bwd_dq(ba, s)) != HN_OK) return rc;
    // dQacc (rows, h*dp) = sum over splits; folded-query chain  Qf = c * gamma * T,  T = Q_h W_k,h
    if ((rc = launch_dq_reduce(bp.dQpart, pl.nsplit_bwd, b, h, L, pl.Lp, pl.dp, pl.dp, 1.0f, bp.E, hp, pl.dp, s)) != HN_OK) return rc;
    if (pack_ks && (rc = launch_pack_fold(bp.E, hp, h, pl.D, pl.dp, pack_ks, 1, rows, s)) != HN_OK) return rc;
    { int rc_ = launch_fill(bp.T, 0.0f, (long)((size_t)rows * hp), s); if (rc_ != HN_OK) return rc_; }
    {   // T = Qraw_h W_k,h   (Qraw = x_hat W_q^T lives in pl.q after attn_prepare)
      GemmExArgs e = gex(qraw, inner, 1, wk, 1, pl.D, bp.T, hp, rows, pl.D, dh, 0);
      e.batch = h; e.strideA = dh; e.strideB = (long)dh * pl.D; e.strideC = pl.dp;
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
    }
    if (p->ctx_gamma && g->ctx_gamma) {   // dgamma += 2 scale * sum T * dQacc
      if ((rc = launch_head_affine(bp.T, hp, pl.dp, bp.E, hp, pl.dp, nullptr, nullptr, two_scale, h, pl.D, pl.dp, hp, rows, bp.dT, s)) != HN_OK) return rc;
      if ((rc = launch_colsum(bp.dT, pl.dp, (long)rows * h, pl.D, 1.0f, g->ctx_gamma, 1, s, bp.red)) != HN_OK) return rc;
    }
    // dT = 2 scale * gamma * dQacc
    if ((rc = launch_head_affine(bp.E, hp, pl.dp, nullptr, 0, 0, p->ctx_gamma, nullptr, two_scale, h, pl.D, pl.dp, hp, rows, bp.dT, s)) != HN_OK) return rc;
    if (dwk) {   // dW_k,h += Qraw_h^T dT_h
      GemmExArgs e = gex(qraw, 1, inner, bp.dT, 1, hp, dwk, pl.D, dh, pl.D, rows, 1);
      e.batch = h; e.strideA = dh; e.strideB = pl.dp; e.strideC = (long)dh * pl.D;
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
    }
    {   // dQ_h = dT_h W_k,h^T
      GemmExArgs e = gex(bp.dT, hp, 1, wk, pl.D, 1, bp.dQ, inner, rows, dh, pl.D, 0);
      e.batch = h; e.strideA = pl.dp; e.strideB = (long)dh * pl.D; e.strideC = dh;
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
    }
  } else {
    const int qp = h * pl.dhp;
    if ((rc = launch_rowdot_heads(dO, inner, dh, O, inner, dh, h, L, dh, rows, bp.delta, s)) != HN_OK) return rc;
    if (pl.dhp == dh) {          // no head padding: the core reads dO where it is (the padding copy is the identity)
      ba.dO = dO; ba.do_b = (long)L * inner; ba.do_h = dh; ba.lddo = inner;
    } else {
      if ((rc = launch_head_affine(dO, inner, dh, nullptr, 0, 0, nullptr, nullptr, 1.0f, h, dh, pl.dhp, qp, rows, bp.dOp, s)) != HN_OK) return rc;
      ba.dO = bp.dOp; ba.do_b = (long)L * qp; ba.do_h = pl.dhp; ba.lddo = qp;
    }
    ba.dKV = bp.dKV; ba.dk_scale = 0.69314718055994530942f;
    // one token split (the latent self-attention, short contexts): the dQ kernel writes the finished, scaled rows itself -- what
    // dq_reduce makes of a single partial, bit for bit -- and that launch is gone (HN_NO_DQ_DIRECT=1: partial + reduce)
    static const bool no_dq_direct = getenv("HN_NO_DQ_DIRECT") != nullptr;
    const bool dq_direct = !no_dq_direct && pl.nsplit_bwd == 1 && !attn_bwd_dq_lds_eligible(ba);
    if (dq_direct) { ba.dQfinal = bp.dQ; ba.dq_ld = inner; ba.dq_pitch = dh; ba.dq_width = dh; ba.dq_scale = two_scale; }
    int rc_pair = HN_OK;
    if (!has_ctx && launch_attn_bwd_self_pair(ba, dh, inner, s, &rc_pair)) {      // latent self-attention: both products in one launch
      if (rc_pair != HN_OK) return rc_pair;
      if (!dq_direct && (rc = launch_dq_reduce(bp.dQpart, pl.nsplit_bwd, b, h, L, pl.Lp, pl.dp, dh, two_scale, bp.dQ, inner, dh, s)) != HN_OK) return rc;
    } else {
      if ((rc = launch_attn_bwd_dq(ba, s)) != HN_OK) return rc;
      if (!dq_direct && (rc = launch_dq_reduce(bp.dQpart, pl.nsplit_bwd, b, h, L, pl.Lp, pl.dp, dh, two_scale, bp.dQ, inner, dh, s)) != HN_OK) return rc;
      if ((rc = launch_attn_bwd_dkv(ba, dh, inner, s)) != HN_OK) return rc;
    }
    const long krows = (long)b * pl.N;
    if (has_ctx) {   // gradients of to_kv and of the context LayerNorm affine from G = dKV^T z and colsum(dKV)
      GemmExArgs e = gex(bp.dKV, 1, 2 * inner, ctx, 1, ld_ctx, bp.G, pl.D, 2 * inner, pl.D, (int)krows, 0);
      e.colsum = bp.cs; e.colsum_accumulate = 0;          // colsum(dKV) rides on the same pass
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
      if ((rc = launch_kv_weight_grads(bp.G, bp.cs, p->w_kv, p->ctx_gamma, p->ctx_beta, 2 * inner, pl.D, g->w_kv, g->ctx_gamma,
                                       g->ctx_beta, s, bp.red)) != HN_OK) return rc;
    } else if (g->w_kv && !(ext && ext->defer_proj)) {   // self-attention: K, V come from x_hat
      GemmExArgs e = gex(bp.dKV, 1, 2 * inner, xhat, 1, qd, g->w_kv, qd, 2 * inner, qd, rows, 1);
      if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
    }
  }

  if (ext && ext->defer_proj) {      // the chain in front of this block runs the projection backward; its batched launch dW_q / dW_kv
    ext->dQ = bp.dQ;
    ext->dKV = has_ctx ? nullptr : bp.dKV;
    ext->xhat = xhat;
    return HN_OK;
  }
  // ---- query projection: dWq += dQ^T x_hat ; dx_hat = dQ Wq (+ dKV Wkv for self-attention)
  HN_REQUIRE(!narrow_ln(p), HN_E_UNSUPPORTED, "attn_bwd: a staged block's projection backward runs in the latent chain");
  if (g->w_q) {
    GemmExArgs e = gex(bp.dQ, 1, inner, xhat, 1, qd, g->w_q, qd, inner, qd, rows, 1);
    if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
  }
  float *dxh = p->norm_w ? bp.dxhat : dx;
  if (ext && ext->dx_without_residual) residual = 0;      // (from here on `residual` only decides whether dy is added into dx)
  const bool direct_acc = !p->norm_w && residual;     // no LayerNorm: dx = dy + dQ Wq directly
  if (direct_acc && dx != dy && (rc = launch_add_into(dy, dx, (long)rows * qd, 0, s)) != HN_OK) return rc;
  {
    GemmExArgs e = gex(bp.dQ, inner, 1, p->w_q, 1, qd, dxh, qd, rows, qd, inner, direct_acc ? 1 : 0);
    if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
  }
  if (!has_ctx) {
    GemmExArgs e = gex(bp.dKV, 2 * inner, 1, p->w_kv, 1, qd, dxh, qd, rows, qd, 2 * inner, 1);
    if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
  }
  if (p->norm_w) {
    if (residual) { if (dx != dy && (rc = launch_add_into(dy, dx, (long)rows * qd, 0, s)) != HN_OK) return rc; }
    return launch_ln_bwd(x_in, bp.dxhat, p->norm_w, rows, qd, dx, residual ? 1 : 0, g->norm_w, g->norm_b, bp.lns, s);
  }
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// feed-forward block
// ------------------------------------------------------------------------------------------------
static size_t ff_ws_bytes(const hn_ff_params *p, int rows) {     // hidden (rows, 4 dim) + the pre-dropout output (rows, dim)
  return align_up((size_t)rows * 5 * p->dim * sizeof(float), 256);
}

static int ff_fwd_impl(const hn_ff_params *p, const float *x_in, float *x_out, int residual, int rows, void *ws,
                       size_t ws_bytes, hipStream_t s, bool training = false) {
  HN_REQUIRE(p && x_in && x_out, HN_E_NULL, "ff: NULL pointer");
  HN_REQUIRE(p->w1 && p->b1 && p->w2 && p->b2, HN_E_NULL, "ff: weight pointer is NULL");
  HN_REQUIRE(p->dim > 0 && rows > 0, HN_E_SHAPE, "ff: dim=%d rows=%d", p->dim, rows);
  HN_REQUIRE(p->gate == HN_GATE_SELU || p->gate == HN_GATE_GELU, HN_E_UNSUPPORTED, "ff: gate=%d", p->gate);
  const int hid = 4 * p->dim;
  HN_REQUIRE(p->dropout >= 0.0f && p->dropout < 1.0f, HN_E_SHAPE, "ff: dropout=%g", (double)p->dropout);
  const bool dropping = training && p->dropout > 0.0f;
  HN_REQUIRE(!narrow_ln(p), HN_E_UNSUPPORTED, "ff: a staged block (dim_valid=%d of %d) runs in the latent chain", p->dim_valid, p->dim);
  int rc = check_ws(ws, ws_bytes, ff_ws_bytes(p, rows), "ff");
  if (rc != HN_OK) return rc;
  float *hidden = (float *)ws;
  GemmArgs g1 = gemm_defaults();
  g1.A = x_in; g1.lda = p->dim;
  g1.W = p->w1; g1.ldw = p->dim;
  g1.C = hidden; g1.ldc = hid;
  g1.bias = p->b1;
  g1.M = rows; g1.N = hid; g1.K = p->dim;
  g1.act = p->gate == HN_GATE_SELU ? ACT_GLU_SELU : ACT_GLU_GELU;
  g1.glu_offset = hid;
  if (p->norm_w) { g1.pro = PRO_LAYERNORM; g1.gamma = p->norm_w; g1.beta = p->norm_b; }
  if ((rc = launch_gemm(g1, s)) != HN_OK) return rc;
  GemmArgs g2 = gemm_defaults();
  g2.A = hidden; g2.lda = hid;
  g2.W = p->w2; g2.ldw = hid;
  g2.C = x_out; g2.ldc = p->dim;
  g2.bias = p->b2;
  g2.M = rows; g2.N = p->dim; g2.K = hid;
  if (dropping) {   // y = x + dropout(h W2^T + b2)   (nn.Dropout is the last module of the block, :347)
    float *pre = hidden + (size_t)rows * hid;
    g2.C = pre;
    if ((rc = launch_gemm(g2, s)) != HN_OK) return rc;
    return launch_dropout_apply(pre, residual ? x_in : nullptr, x_out, rows, p->dim, drop_of(p->dropout, p->rng, true), s);
  }
  if (residual) { g2.R = x_in; g2.ldr = p->dim; }
  return launch_gemm(g2, s);
}

// ------------------------------------------------------------------------------------------------
// feed-forward block, backward
// ------------------------------------------------------------------------------------------------
struct FFBwdPlan { float *u, *h, *dh, *xhat, *dxhat, *lns, *red, *dyd; size_t bytes; };

static void plan_ff_bwd(const hn_ff_params *p, int rows, void *ws, size_t ws_bytes, FFBwdPlan *pl) {
  Arena ar(ws, ws_bytes);
  const size_t hid = 4 * (size_t)p->dim;
  pl->u = ar.take<float>((size_t)rows * 2 * hid);
  pl->h = ar.take<float>((size_t)rows * hid);
  pl->dh = ar.take<float>((size_t)rows * hid);
  pl->xhat = ar.take<float>((size_t)rows * p->dim);
  pl->dxhat = ar.take<float>((size_t)rows * p->dim);
  pl->lns = ar.take<float>(ln_bwd_scratch_floats(rows, p->dim));
  pl->red = ar.take<float>(reduce_scratch_floats(8L * p->dim * p->dim, 8 * p->dim));
  pl->dyd = ar.take<float>((size_t)rows * p->dim);          // dropout: the gradient that enters the block proper
  pl->bytes = ar.off;
}

static int ff_bwd_impl(const hn_ff_params *p, const float *x_in, const float *dy, float *dx, int residual, int rows,
                       const hn_ff_grads *g, void *ws, size_t ws_bytes, hipStream_t s) {
  HN_REQUIRE(p && x_in && dy && dx && g, HN_E_NULL, "ff_bwd: NULL pointer");
  HN_REQUIRE(p->w1 && p->b1 && p->w2 && p->b2, HN_E_NULL, "ff_bwd: weight pointer is NULL");
  HN_REQUIRE(p->dim > 0 && rows > 0, HN_E_SHAPE, "ff_bwd: dim=%d rows=%d", p->dim, rows);
  HN_REQUIRE(!narrow_ln(p), HN_E_UNSUPPORTED, "ff_bwd: a staged block (dim_valid=%d of %d) runs in the latent chain", p->dim_valid, p->dim);
  FFBwdPlan pl;
  plan_ff_bwd(p, rows, nullptr, 0, &pl);
  int rc = check_ws(ws, ws_bytes, pl.bytes, "ff_bwd");
  if (rc != HN_OK) return rc;
  plan_ff_bwd(p, rows, ws, ws_bytes, &pl);
  const int d = p->dim, hid = 4 * d;
  // recompute the pre-activations u = [a | g] = LN(x) W1^T + b1 and the normalised operand
  GemmArgs g1 = gemm_defaults();
  g1.A = x_in; g1.lda = d; g1.W = p->w1; g1.ldw = d; g1.C = pl.u; g1.ldc = 2 * hid; g1.bias = p->b1;
  g1.M = rows; g1.N = 2 * hid; g1.K = d;
  if (p->norm_w) { g1.pro = PRO_LAYERNORM; g1.gamma = p->norm_w; g1.beta = p->norm_b; }
  if ((rc = launch_gemm(g1, s)) != HN_OK) return rc;
  const float *xhat = x_in;
  if (p->norm_w) {
    if ((rc = launch_ln_fwd(x_in, p->norm_w, p->norm_b, rows, d, pl.xhat, s)) != HN_OK) return rc;
    xhat = pl.xhat;
  }
  // dropout sits between the block and the residual add: the block proper sees dy * keepscale (the forward's mask)
  const float *dyf = dy;
  if (p->dropout > 0.0f) {
    if ((rc = launch_dropout_apply(dy, nullptr, pl.dyd, rows, d, drop_of(p->dropout, p->rng, true), s)) != HN_OK) return rc;
    dyf = pl.dyd;
  }
  // dh = dy W2          (W2 is (d, hid): B(j = k, c = n) = W2[n, k])
  GemmExArgs e = {};
  e.batch = 1; e.alpha = 1.0f;
  e.A = dyf; e.a_rs = d; e.a_cs = 1; e.B = p->w2; e.b_rs = 1; e.b_cs = hid; e.C = pl.dh; e.ldc = hid; e.M = rows; e.N = hid; e.K = d;
  if ((rc = launch_gemm_ex(e, s, pl.red)) != HN_OK) return rc;
  // h = a * act(g);  u <- du
  if ((rc = launch_glu_bwd(pl.u, pl.dh, pl.h, rows, hid, p->gate == HN_GATE_GELU, s)) != HN_OK) return rc;
  if (g->w2) {   // dW2 += dy^T h
    GemmExArgs w = {};
    w.batch = 1; w.alpha = 1.0f; w.accumulate = 1;
    w.A = dyf; w.a_rs = 1; w.a_cs = d; w.B = pl.h; w.b_rs = 1; w.b_cs = hid; w.C = g->w2; w.ldc = hid; w.M = d; w.N = hid; w.K = rows;
    w.colsum = g->b2; w.colsum_accumulate = 1;          // db2 += colsum(dy) from the same pass
    if ((rc = launch_gemm_ex(w, s, pl.red)) != HN_OK) return rc;
  } else if (g->b2 && (rc = launch_colsum(dyf, d, rows, d, 1.0f, g->b2, 1, s, pl.red)) != HN_OK) return rc;
  if (g->w1) {   // dW1 += du^T x_hat
    GemmExArgs w = {};
    w.batch = 1; w.alpha = 1.0f; w.accumulate = 1;
    w.A = pl.u; w.a_rs = 1; w.a_cs = 2 * hid; w.B = xhat; w.b_rs = 1; w.b_cs = d; w.C = g->w1; w.ldc = d; w.M = 2 * hid; w.N = d; w.K = rows;
    w.colsum = g->b1; w.colsum_accumulate = 1;          // db1 += colsum(du) from the same pass
    if ((rc = launch_gemm_ex(w, s, pl.red)) != HN_OK) return rc;
  } else if (g->b1 && (rc = launch_colsum(pl.u, 2 * hid, rows, 2 * hid, 1.0f, g->b1, 1, s, pl.red)) != HN_OK) return rc;
  // dx_hat = du W1      (W1 is (2 hid, d): B(j = k, c = n) = W1[n, k])
  GemmExArgs x = {};
  x.batch = 1; x.alpha = 1.0f;
  x.A = pl.u; x.a_rs = 2 * hid; x.a_cs = 1; x.B = p->w1; x.b_rs = 1; x.b_cs = d; x.M = rows; x.N = d; x.K = 2 * hid;
  if (p->norm_w) {
    x.C = pl.dxhat; x.ldc = d;
    if ((rc = launch_gemm_ex(x, s, pl.red)) != HN_OK) return rc;
    if (residual) { if (dx != dy && (rc = launch_add_into(dy, dx, (long)rows * d, 0, s)) != HN_OK) return rc; }
    return launch_ln_bwd(x_in, pl.dxhat, p->norm_w, rows, d, dx, residual ? 1 : 0, g->norm_w, g->norm_b, pl.lns, s);
  }
  if (residual) { if (dx != dy && (rc = launch_add_into(dy, dx, (long)rows * d, 0, s)) != HN_OK) return rc; }
  x.C = dx; x.ldc = d; x.accumulate = residual ? 1 : 0;
  return launch_gemm_ex(x, s, pl.red);
}

static int context_pitch(int D, int dim_head) {
  const int dhp = pad_head_dim(dim_head);
  int dp = D <= 15 ? 16 : (D <= 31 ? 32 : 0);   // leave column dp-1 free for the kernel's synthetic ones column
  if (dp != 0 && dhp != 0 && dp <= dhp) return dp;
  return (D + 3) / 4 * 4;
}

// ------------------------------------------------------------------------------------------------
// whole forward
// ------------------------------------------------------------------------------------------------
struct FusionPlan {
  float *x;
  float *z[16];
  int ldz[16], N[16], D[16];
  bool ones[16];   // z carries the synthetic ones column (rank-D pitch with a free last column)
  float *wvf[16];  // inference: folded value projections of all layers (depth, inner, 16) for the chain's merge head, or NULL
  float *wqf[16];  // inference: folded query projections of all layers (depth, 128, l_d) for the chain's Q stage (ChainArgs.qf), or NULL
  float *cbound;   // score bounds written by that stage, (b, heads, l_c)
  int pack[16];    // ... and uses the packed channel layout with this many QK^T k-steps (0 = natural)
  bool bf16[16];   // core_precision = bf16: z holds the bf16 images (zb, then zT) instead of the fp32 rows
  uint16_t *z16[16];   // core_precision = bf16, explicit binding of a large patch bag: bf16 image of the rows of z (pitch gemm_bf16_pitch(D)) for the K/V projections, or NULL
  int Np[16], ns[16];
  int *flags;      // one pre-zeroed fallback flag of the score-bound softmax per (layer, modality), then the chain cluster flags
  float *xchg;     // exchange buffer of the latent chain's cluster mode (small batches)
  float *tabv[16], *taby[16];   // one-token modalities (inference): V (depth, b, inner) and block outputs (depth, b, query_dim) of ALL layers
  bool tab_ahead[16];           // ... computed ahead of the layer loop in two batched launches
  void *op_ws;
  size_t op_ws_bytes, bytes;
  int dominant;   // modality with the most tokens among the present ones
  bool chain;     // the latent side runs on latent_chain_kernel (chain.hip): l_d = 128, l_c % 16 == 0
  float *cq, *ckv;   // ... which writes the NEXT attention block's projections here (outside op_ws: the block in front still owns it)
  float *lk, *lvt;   // layer chains (lchain.hip, inference): LAYER_KV_SLOTS K images (b, 8, 128, 64) and V^T images (b, 8, 64, 128), or NULL
};
constexpr int LAYER_KV_SLOTS = LSEG_MAX / 2;

static int plan_fusion(const hn_model *m, const hn_modality_input *in, int b, void *ws, size_t ws_bytes, FusionPlan *fp,
                       bool inference = false) {
  HN_REQUIRE(m && in, HN_E_NULL, "fusion: NULL model / inputs");
  HN_REQUIRE(m->n_modalities >= 1 && m->n_modalities <= 16, HN_E_UNSUPPORTED, "fusion: n_modalities=%d (1..16)", m->n_modalities);
  HN_REQUIRE(m->depth >= 1 && m->l_c >= 1 && m->l_d >= 1 && b >= 1, HN_E_SHAPE, "fusion: depth=%d l_c=%d l_d=%d b=%d",
             m->depth, m->l_c, m->l_d, b);
  HN_REQUIRE(m->self_per_cross_attn == 0 || m->self_per_cross_attn == 1, HN_E_UNSUPPORTED,
             "fusion: self_per_cross_attn=%d (the reference only runs 0 or 1, healnet.py:242)", m->self_per_cross_attn);
  Arena ar(ws, ws_bytes);
  fp->x = ar.take<float>(rows16((size_t)b * m->l_c) * m->l_d);
  size_t op_max = 0;
  fp->dominant = -1;
  long best = -1;
  for (int i = 0; i < m->n_modalities; ++i) {
    fp->z[i] = nullptr;
    fp->z16[i] = nullptr;
    if (in[i].data == nullptr) continue;
    const int axes = m->num_spatial_axes[i];
    HN_REQUIRE(axes >= 1 && axes <= HN_MAX_AXES, HN_E_UNSUPPORTED, "fusion: modality %d has %d spatial axes (1..%d)", i, axes,
               HN_MAX_AXES);
    long n = 1;
    for (int a = 0; a < axes; ++a) {
      HN_REQUIRE(in[i].spatial[a] > 0, HN_E_SHAPE, "fusion: modality %d spatial[%d]=%d", i, a, in[i].spatial[a]);
      n *= in[i].spatial[a];
    }
    HN_REQUIRE(n < (1L << 31) / 16, HN_E_UNSUPPORTED, "fusion: modality %d has too many tokens", i);
    fp->N[i] = (int)n;
    fp->D[i] = m->channel_dims[i] + (m->fourier_encode_data ? axes * (2 * m->num_freq_bands + 1) : 0);
    const hn_attn_params *ap = &m->cross_attn[i];
    fp->ldz[i] = context_pitch(fp->D[i], ap->dim_head);
    // Dropout on the probabilities of the shared-context (rank-D) binding keeps the thinned row sum in a spare column of the
    // context row; D == 16 / 32 exactly has none, so such a modality takes the explicit K/V binding (pitch D + 4) when any of
    // its blocks drops.
    if ((fp->ldz[i] == 16 || fp->ldz[i] == 32) && fp->D[i] == fp->ldz[i]) {
      bool drops = false;
      for (int layer = 0; layer < m->depth; ++layer) drops = drops || m->cross_attn[layer * m->n_modalities + i].dropout > 0.0f;
      if (drops) fp->ldz[i] += 4;
    }
    // ones column / packed channel order: only for the shared-context (rank-D) binding, where the core reads z itself; the
    // explicit binding projects z through to_kv and needs the natural layout (the pitch alone does not tell: D = 29 with
    // dim_head = 4 gets pitch 32 from the 4-float rounding)
    // (and never for a one-token context: its shortcut runs z through the value projection in natural order)
    fp->ones[i] = n > 1 && (fp->ldz[i] == 16 || fp->ldz[i] == 32) && fp->D[i] <= fp->ldz[i] - 1;
    for (int layer = 0; layer < m->depth && fp->ones[i]; ++layer) {
      AttnPlan pl;
      int rc = plan_attn(&m->cross_attn[layer * m->n_modalities + i], true, fp->ldz[i], b, m->l_c, (int)n, fp->D[i], nullptr, 0, &pl, 0);
      if (rc != HN_OK) return rc;
      fp->ones[i] = pl.rank_d && pl.ones;
    }
    fp->pack[i] = fp->ones[i] ? packed_steps(fp->D[i], fp->ldz[i]) : 0;
    // the chain behind a shared-context block can merge its split partials itself when the folded value projections of the
    // modality's layers are staged up front (one launch per forward): dp = 16, equal heads / dim_head over the layers
    fp->wvf[i] = nullptr;
    fp->wqf[i] = nullptr;
    if (inference && fp->ones[i] && fp->ldz[i] == 16 && m->depth <= HN_SKINNY_MAXZ) {
      const hn_attn_params &a0 = m->cross_attn[i];
      bool ok = a0.heads <= 8 && (a0.dim_head == 16 || a0.dim_head == 32 || a0.dim_head == 64) && a0.ctx_gamma != nullptr;
      for (int layer = 1; layer < m->depth && ok; ++layer) {
        const hn_attn_params &al = m->cross_attn[layer * m->n_modalities + i];
        ok = al.heads == a0.heads && al.dim_head == a0.dim_head && al.ctx_gamma != nullptr;
      }
      if (ok) fp->wvf[i] = ar.take<float>((size_t)m->depth * a0.heads * a0.dim_head * 16);
      // ... and the query side: eight heads of 16 packed slots fill the chain's 128-column Q stage exactly
      if (ok && a0.heads * 16 == 128 && m->l_d == 128 && m->l_c % 16 == 0 && a0.dim_head <= 128 && a0.query_dim == m->l_d && fp->pack[i] > 0)
        fp->wqf[i] = ar.take<float>((size_t)m->depth * 128 * m->l_d);
    }
    // One workspace size serves the inference forward (which may use the bf16 core) and the training forward / backward
    // (always fp32) of the same model: size for the larger of the two layouts.
    const bool want_bf16 = (m->core_precision == HN_CORE_BF16 || m->core_precision == HN_CORE_BF16X3) && fp->ones[i] && n > 1;
    const int ns = m->core_precision == HN_CORE_BF16X3 ? 2 : 1;
    fp->ns[i] = ns;
    HN_REQUIRE(in[i].dtype == HN_F32 || in[i].dtype == HN_BF16 || in[i].dtype == HN_U8, HN_E_UNSUPPORTED,
               "fusion: modality %d dtype=%d", i, in[i].dtype);
    fp->bf16[i] = want_bf16 && inference;
    fp->Np[i] = (int)((n + 31) / 32 * 32);
    if (fp->bf16[i]) { fp->pack[i] = 0; fp->wvf[i] = nullptr; fp->wqf[i] = nullptr; }
    size_t zbytes = (size_t)b * n * fp->ldz[i] * sizeof(float);
    if (want_bf16) {
      const size_t zb16 = (size_t)b * fp->Np[i] * (bf16_row_slots(fp->ldz[i], ns) + ns * fp->ldz[i]) * sizeof(uint16_t);
      if (zb16 > zbytes) zbytes = zb16;
    }
    fp->z[i] = (float *)ar.take<char>(zbytes);
    // (sized whether or not this call is the inference forward: one workspace size serves all entry points; used by inference only)
    fp->z16[i] = nullptr;
    if (m->core_precision == HN_CORE_BF16 && !fp->ones[i] && n > 1 && fp->ldz[i] % 4 == 0 &&
        gemm_bf16_shape_ok((long)b * n, 2 * ap->heads * ap->dim_head, fp->D[i])) {
      uint16_t *img = (uint16_t *)ar.take<char>((size_t)b * n * gemm_bf16_pitch(fp->D[i]) * sizeof(uint16_t));
      if (inference) fp->z16[i] = img;
    }
    // one-token context (tabular / omic): y_l = LeakyReLU(W_out,l (W_v,l c_hat_l) + b_out,l) does not depend on the latent array,
    // so the inference forward evaluates all layers' vectors up front (weight-streaming GEMV shapes only, equal heads / dims)
    fp->tab_ahead[i] = false;
    fp->tabv[i] = fp->taby[i] = nullptr;
    if (inference && n == 1 && m->depth <= HN_SKINNY_MAXZ && fp->D[i] >= 512 && b <= 512) {
      bool same = true;
      for (int layer = 1; layer < m->depth; ++layer) {
        const hn_attn_params &a0 = m->cross_attn[i], &al = m->cross_attn[layer * m->n_modalities + i];
        same = same && al.heads == a0.heads && al.dim_head == a0.dim_head && al.query_dim == a0.query_dim &&
               (al.ctx_gamma != nullptr) == (a0.ctx_gamma != nullptr);
      }
      const int inner0 = ap->heads * ap->dim_head;
      if (same && inner0 >= 512) {
        fp->tab_ahead[i] = true;
        fp->tabv[i] = ar.take<float>((size_t)m->depth * b * inner0);
        fp->taby[i] = ar.take<float>((size_t)m->depth * b * ap->query_dim);
      }
    }
    if (n > best) { best = n; fp->dominant = i; }
    for (int layer = 0; layer < m->depth; ++layer) {
      AttnPlan pl;
      int rc = plan_attn(&m->cross_attn[layer * m->n_modalities + i], true, fp->ldz[i], b, m->l_c, (int)n, fp->D[i], nullptr,
                         0, &pl, 0);
      if (rc != HN_OK) return rc;
      if (pl.bytes > op_max) op_max = pl.bytes;
      if (want_bf16) {
        if ((rc = plan_attn(&m->cross_attn[layer * m->n_modalities + i], true, fp->ldz[i], b, m->l_c, (int)n, fp->D[i], nullptr,
                            0, &pl, ns)) != HN_OK) return rc;
        if (pl.bytes > op_max) op_max = pl.bytes;
      }
    }
  }
  HN_REQUIRE(fp->dominant >= 0, HN_E_SHAPE, "fusion: every modality is missing");
  if (m->self_per_cross_attn > 0) {
    for (int layer = 0; layer < m->depth; ++layer) {
      AttnPlan pl;
      int rc = plan_attn(&m->self_attn[layer], false, 0, b, m->l_c, m->l_c, m->l_d, nullptr, 0, &pl);
      if (rc != HN_OK) return rc;
      if (pl.bytes > op_max) op_max = pl.bytes;
    }
  }
  const size_t ffb = align_up((size_t)b * m->l_c * 5 * m->l_d * sizeof(float), 256);
  if (ffb > op_max) op_max = ffb;
  fp->flags = ar.take<int>((size_t)m->depth * m->n_modalities + CHAIN_XCHG_FLAGS);
  fp->xchg = ar.take<float>(CHAIN_XCHG_FLOATS);
  {
    int max_inner = 0, max_inner_self = 0;
    for (int k = 0; k < m->depth * m->n_modalities; ++k) max_inner = max_inner > m->cross_attn[k].heads * pad_head_dim(m->cross_attn[k].dim_head) ? max_inner : m->cross_attn[k].heads * pad_head_dim(m->cross_attn[k].dim_head);
    if (m->self_per_cross_attn > 0)
      for (int k = 0; k < m->depth; ++k) max_inner_self = max_inner_self > m->self_attn[k].heads * pad_head_dim(m->self_attn[k].dim_head) ? max_inner_self : m->self_attn[k].heads * pad_head_dim(m->self_attn[k].dim_head);
    if (max_inner_self > max_inner) max_inner = max_inner_self;
    // Every workgroup of the chain streams ALL weights of the chain through its CU however few rows there are, while the 2-D tiled
    // per-block GEMMs shrink with the row count; until the loader lost its vector address arithmetic (chain.hip, v6) that made the
    // per-block launches faster below ~160 workgroups.  Measured at cfg2 (l_c = 128) since, chain vs per-block launches, ms per
    // forward: b = 1 0.851 / 0.848, 2: 0.820 / 0.822, 4: 0.908 / 0.915, 8: 1.193 / 1.212, 16: 1.740 / 1.830, 24: 2.435 / 2.605,
    // 32: 2.911 / 3.205 -- no crossover left, the chain is the route whenever its shapes apply (HN_NO_CHAIN=1: development switch).
    // both forwards (the training one keeps x_mid).  Host-owned trace / output buffers hold exactly b * l_c rows, so a row count
    // that is not a multiple of 16 needs the staged route (internal, row-padded buffers).
    fp->chain = m->l_d == 128 && (m->l_c % 16 == 0 || m->l_d_valid > 0);
    fp->cq = fp->ckv = nullptr;
    fp->cbound = nullptr;
    if (fp->chain) {
      fp->cbound = ar.take<float>((size_t)b * 8 * m->l_c);
      fp->cq = ar.take<float>(rows16((size_t)b * m->l_c) * max_inner);
      fp->ckv = ar.take<float>(rows16((size_t)b * m->l_c) * 2 * (max_inner_self > 0 ? max_inner_self : 1));
    }
    fp->lk = fp->lvt = nullptr;
    if (fp->chain && inference && m->l_c == 128 && m->self_per_cross_attn > 0 && max_inner_self == 512 && latent_layer_enabled()) {
      fp->lk = ar.take<float>((size_t)LAYER_KV_SLOTS * b * 8 * 128 * 64);
      fp->lvt = ar.take<float>((size_t)LAYER_KV_SLOTS * b * 8 * 128 * 64);
    }
  }
  fp->op_ws_bytes = op_max;
  fp->op_ws = ar.take<char>(op_max);
  fp->bytes = ar.off;
  if (ws != nullptr && ar.overflow) return fail(HN_E_WORKSPACE, "fusion: workspace %zu bytes < required %zu", ws_bytes, ar.off);
  return HN_OK;
}

// ------------------------------------------------------------------------------------------------
// training: block schedule, tape layout, backward workspace
// ------------------------------------------------------------------------------------------------
enum StepKind { STEP_CROSS_ATTN, STEP_CROSS_FF, STEP_SELF_ATTN, STEP_SELF_FF };
struct Step { int kind, layer, m; };

// the executed blocks in order, identical for the training forward and the backward (healnet.py:227-245)
static int build_schedule(const hn_model *m, const hn_modality_input *in, int skip_self_on_missing, Step *steps, int cap) {
  int n = 0;
  for (int layer = 0; layer < m->depth; ++layer)
    for (int i = 0; i < m->n_modalities; ++i) {
      const bool present = in[i].data != nullptr;
      if (!present && ((skip_self_on_missing >> i) & 1)) continue;   // bit i: the verbose=True `continue` quirk for modality i
      if (present) {
        if (n + 2 > cap) return -1;
        steps[n++] = {STEP_CROSS_ATTN, layer, i};
        steps[n++] = {STEP_CROSS_FF, layer, i};
      }
      if (m->self_per_cross_attn > 0) {
        if (n + 2 > cap) return -1;
        steps[n++] = {STEP_SELF_ATTN, layer, i};
        steps[n++] = {STEP_SELF_FF, layer, i};
      }
    }
  return n;
}

constexpr int kMaxSteps = 4096;

struct TapePlan {
  int nsteps;
  Step steps[kMaxSteps];
  size_t x_off[kMaxSteps + 1];       // float offsets of the latent array before step k (x_off[nsteps] = final)
  size_t stats_off[kMaxSteps], saved_off[kMaxSteps];
  size_t q_off[kMaxSteps], kv_off[kMaxSteps];      // projections of the attention block at step k kept for the backward (kNoSlot: none)
  // the normalised contexts z (K1's output) of the present modalities: written here by the training forward and read back by the
  // backward instead of a second encode (round 4: 102 MB per patch bag of cfg4 against a 43 us HBM pass per step -- the tape has
  // the room on a 288 GB part; HN_NO_Z_TAPE=1 keeps them in the workspace and re-encodes)
  size_t z_off[16];
  // LN(x) of an attention block's input, the operand of its dW_q / dW_kv products: written by the chain that projects for the block
  // (it holds the tile in LDS anyway) or by the block itself, instead of a LayerNorm launch in front of every block backward
  size_t xhat_off[kMaxSteps];
  size_t floats;
};
constexpr size_t kNoSlot = (size_t)-1;

static int plan_tape(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self, const FusionPlan &fp, TapePlan *tp) {
  tp->nsteps = build_schedule(m, in, skip_self, tp->steps, kMaxSteps);
  HN_REQUIRE(tp->nsteps >= 0, HN_E_UNSUPPORTED, "fusion: more than %d blocks", kMaxSteps);
  const size_t xn = rows16((size_t)b * m->l_c) * m->l_d;
  size_t off = 0;
  for (int k = 0; k <= tp->nsteps; ++k) { tp->x_off[k] = off; off += align_up(xn, 64); }
  for (int k = 0; k < tp->nsteps; ++k) {
    const Step &st = tp->steps[k];
    tp->stats_off[k] = tp->saved_off[k] = 0;
    tp->q_off[k] = tp->kv_off[k] = tp->xhat_off[k] = kNoSlot;
    if (st.kind == STEP_CROSS_ATTN || st.kind == STEP_SELF_ATTN) {
      const bool cross = st.kind == STEP_CROSS_ATTN;
      const hn_attn_params *ap = cross ? &m->cross_attn[st.layer * m->n_modalities + st.m] : &m->self_attn[st.layer];
      AttnPlan pl;
      int rc = plan_attn(ap, cross, cross ? fp.ldz[st.m] : 0, b, m->l_c, cross ? fp.N[st.m] : m->l_c, cross ? fp.D[st.m] : m->l_d,
                         nullptr, 0, &pl);
      if (rc != HN_OK) return rc;
      tp->stats_off[k] = off; off += align_up((size_t)b * ap->heads * m->l_c * 2, 64);
      tp->saved_off[k] = off; off += align_up(attn_saved_floats(pl, cross, cross && (masked || ap->dropout > 0.0f), b, m->l_c), 64);
      // the q (and, for the latent self-attention, k / v) projections: 25 MB per self block at cfg2 b = 32 against a 10-27 us
      // recompute launch in front of every attention core backward (not for the one-token shortcut, which has no q / k)
      const bool one_token = cross && pl.N == 1 && !masked && !(ap->dropout > 0.0f);
      if (!one_token) {
        const size_t rows = rows16((size_t)b * m->l_c);
        tp->q_off[k] = off; off += align_up(rows * (pl.rank_d ? pl.inner : pl.heads * pl.dhp), 64);
        if (!cross) { tp->kv_off[k] = off; off += align_up(rows * 2 * pl.heads * pl.dhp, 64); }
        static const bool no_xhat_tape = tuning_env("HN_NO_XHAT_TAPE") != nullptr;
        if (ap->norm_w && ap->query_dim == m->l_d && !no_xhat_tape) { tp->xhat_off[k] = off; off += align_up(rows * m->l_d, 64); }
      }
    }
  }
  static const bool no_z_tape = tuning_env("HN_NO_Z_TAPE") != nullptr;
  for (int i = 0; i < m->n_modalities; ++i) {
    tp->z_off[i] = kNoSlot;
    if (no_z_tape || !in[i].data || fp.N[i] <= 0) continue;
    tp->z_off[i] = off; off += align_up((size_t)b * fp.N[i] * fp.ldz[i], 64);
  }
  tp->floats = off;
  return HN_OK;
}

// Registers, in the transposed-weight cache (backward.hip), every weight the dX products of a backward pass read in NN form.
// `steps == nullptr`: the full schedule (an upper bound, for the workspace size).
static void register_transposes(const hn_model *m, const hn_modality_input *in, int b, int masked, const Step *steps, int nsteps) {
  transpose_cache_begin();
  const int M = m->n_modalities, d = m->l_d;
  // the NN route needs >= 256 rows (launch_gemm_ex); the fused latent backward (bchain.hip) reads the transposes at any row count
  if ((long)b * m->l_c < 256 && !latent_bchain_supported(b * m->l_c, m->l_d, 4 * m->l_d)) return;
  auto add_attn = [&](const hn_attn_params &ap, bool self) {
    const int inner = ap.heads * ap.dim_head, qd = ap.query_dim;
    // (staged blocks: the padded allocation -- w_out rows of pitch wo_ld, w_q / w_kv with zero rows up to a multiple of 128)
    const bool st = staged_attn(&ap);
    transpose_cache_add(ap.w_out, wo_ld(&ap), qd, wo_ld(&ap));      // dO = dpre W_out
    transpose_cache_add(ap.w_q, qd, st ? up128(inner) : inner, qd);           // dx_hat = dQ W_q
    if (self) transpose_cache_add(ap.w_kv, qd, st ? up128(2 * inner) : 2 * inner, qd);   // ... + dKV W_kv
  };
  auto add_ff = [&](const hn_ff_params &fp) {
    transpose_cache_add(fp.w2, 4 * fp.dim, fp.dim, 4 * fp.dim);   // dh = dy W2
    transpose_cache_add(fp.w1, fp.dim, 8 * fp.dim, fp.dim);       // dx_hat = du W1
  };
  auto one_token_shortcut = [&](const hn_attn_params &ap, int i) {
    long n = 1;
    for (int a = 0; a < m->num_spatial_axes[i]; ++a) n *= in[i].spatial[a];
    return n == 1 && !masked && !(ap.dropout > 0.0f);
  };
  if (steps == nullptr) {
    for (int layer = 0; layer < m->depth; ++layer) {
      for (int i = 0; i < M; ++i) {
        if (!in[i].data) continue;
        if (!one_token_shortcut(m->cross_attn[layer * M + i], i)) add_attn(m->cross_attn[layer * M + i], false);
        add_ff(m->cross_ff[layer * M + i]);
      }
      if (m->self_per_cross_attn > 0) { add_attn(m->self_attn[layer], true); add_ff(m->self_ff[layer]); }
    }
    return;
  }
  for (int k = 0; k < nsteps; ++k) {
    const Step &st = steps[k];
    if (st.kind == STEP_CROSS_ATTN) { if (!one_token_shortcut(m->cross_attn[st.layer * M + st.m], st.m)) add_attn(m->cross_attn[st.layer * M + st.m], false); }
    else if (st.kind == STEP_SELF_ATTN) add_attn(m->self_attn[st.layer], true);
    else if (st.kind == STEP_CROSS_FF) add_ff(m->cross_ff[st.layer * M + st.m]);
    else add_ff(m->self_ff[st.layer]);
  }
  (void)d;
}

// buffers of the fused latent backward (bchain.hip): what a chain hands to the batched weight-gradient launch and to the
// attention core backward in front of it
// Scratch of the fused latent backward.  What a chain leaves for its weight-gradient products (H, dU, Xhat, dYff, dPre, lnpart) and
// what the attention backward behind it leaves for the NEXT chain's (dQ, dKV of a latent block) exists BCHAIN_SETS times: the
// products of up to BCHAIN_SETS - 1 consecutive chains -- normally all chains of a layer -- wait in one batch and run as ONE
// gemm_tn_lds_multi launch + ONE reduce (round 5; a launch pair per chain until then: 12 x 27 us of 16-28 workgroups each at cfg4).
constexpr int BCHAIN_SETS = 8;
struct BChainSet { float *H, *dU, *Xhat, *dYff, *dPre, *lnpart, *dQ, *dKV; };
struct BChainBufs { BChainSet set[BCHAIN_SETS]; float *dO, *tn, *xchg; int *xflags; size_t tn_floats; bool ok; };
constexpr int BCHAIN_XFLAGS = 2 * 256 + 1;

static size_t bchain_tn_scratch_floats(int rows) {
  // upper bound over every product subset a chain can batch (dW1, dW2, dW_out, dW_q, dW_kv at inner = 512): fewer products means
  // fewer tiles and therefore MORE k-slices (up to GEMM_EX_SPLITS), so the bound is the split cap times all partial sizes
  (void)rows;
  const long MN[5][2] = {{1024, 128}, {128, 512}, {128, 512}, {512, 128}, {1024, 128}};
  size_t n = 0;
  for (int i = 0; i < 5; ++i) n += (size_t)GEMM_EX_SPLITS * (MN[i][0] * MN[i][1] + MN[i][0]) + 128;
  return n;
}

static int fusion_bwd_workspace(const hn_model *m, const hn_modality_input *in, int b, int masked, void *ws, size_t ws_bytes,
                                FusionPlan *fp, float **dX, float **head_scratch, void **op_ws, size_t *op_bytes, size_t *total,
                                float **tbuf = nullptr, size_t *tfloats = nullptr, BChainBufs *bb = nullptr) {
  // same z / x carve as the forward (x is unused), then the backward scratch
  int rc = plan_fusion(m, in, b, nullptr, 0, fp);
  if (rc != HN_OK) return rc;
  Arena ar(ws, ws_bytes);
  for (int i = 0; i < m->n_modalities; ++i) {
    fp->z[i] = nullptr;
    if (in[i].data) fp->z[i] = ar.take<float>((size_t)b * fp->N[i] * fp->ldz[i]);
  }
  *dX = ar.take<float>(rows16((size_t)b * m->l_c) * m->l_d);
  *head_scratch = ar.take<float>(head_bwd_scratch_floats(b, m->l_d, m->out_dims > 0 ? m->out_dims : 1));
  size_t need = 0;
  for (int layer = 0; layer < m->depth; ++layer) {
    for (int i = 0; i < m->n_modalities; ++i) {
      if (!in[i].data) continue;
      const hn_attn_params *ap = &m->cross_attn[layer * m->n_modalities + i];
      AttnPlan pl;
      if ((rc = plan_attn(ap, true, fp->ldz[i], b, m->l_c, fp->N[i], fp->D[i], nullptr, 0, &pl)) != HN_OK) return rc;
      AttnBwdPlan bp;
      if ((rc = plan_attn_bwd(ap, pl, true, masked != 0 || ap->dropout > 0.0f, b, m->l_c, nullptr, 0, &bp)) != HN_OK) return rc;
      if (bp.bytes > need) need = bp.bytes;
    }
    if (m->self_per_cross_attn > 0) {
      AttnPlan pl;
      if ((rc = plan_attn(&m->self_attn[layer], false, 0, b, m->l_c, m->l_c, m->l_d, nullptr, 0, &pl)) != HN_OK) return rc;
      AttnBwdPlan bp;
      if ((rc = plan_attn_bwd(&m->self_attn[layer], pl, false, false, b, m->l_c, nullptr, 0, &bp)) != HN_OK) return rc;
      if (bp.bytes > need) need = bp.bytes;
    }
  }
  hn_ff_params ffp;
  memset(&ffp, 0, sizeof(ffp));
  ffp.dim = m->l_d;
  FFBwdPlan fb;
  plan_ff_bwd(&ffp, b * m->l_c, nullptr, 0, &fb);
  if (fb.bytes > need) need = fb.bytes;
  *op_bytes = need;
  *op_ws = ar.take<char>(need);
  register_transposes(m, in, b, masked, nullptr, 0);          // upper bound of the transposed-weight cache
  const size_t tf = transpose_cache_floats();
  transpose_cache_end();
  float *tb = ar.take<float>(tf);
  if (tbuf) *tbuf = tb;
  if (tfloats) *tfloats = tf;
  BChainBufs cb;
  memset(&cb, 0, sizeof(cb));
  const size_t rows = rows16((size_t)b * m->l_c);
  cb.ok = latent_bchain_supported(b * m->l_c, m->l_d, 4 * m->l_d);
  if (cb.ok) {
    for (int j = 0; j < BCHAIN_SETS; ++j) {
      BChainSet &bs = cb.set[j];
      bs.H = ar.take<float>(rows * 512);
      bs.dU = ar.take<float>(rows * 1024);
      bs.Xhat = ar.take<float>(rows * 128);
      bs.dYff = ar.take<float>(rows * 128);
      bs.dPre = ar.take<float>(rows * 128);
      bs.lnpart = ar.take<float>((rows / 16) * 4 * 128);
      bs.dQ = ar.take<float>(rows * 512);             // latent blocks whose projection backward rides on a chain: inner <= 512
      bs.dKV = ar.take<float>(rows * 1024);
    }
    cb.dO = ar.take<float>(rows * 512);
    cb.tn_floats = 2 * bchain_tn_scratch_floats(rows);
    cb.tn = ar.take<float>(cb.tn_floats);
    cb.xchg = ar.take<float>((size_t)2 * 256 * 16 * 128);      // cluster mode: two exchanges x <= 256 workgroups x one partial tile
    cb.xflags = ar.take<int>(BCHAIN_XFLAGS);
  }
  if (bb) *bb = cb;
  *total = ar.off;
  if (ws != nullptr && ar.overflow) return fail(HN_E_WORKSPACE, "fusion_backward: workspace %zu bytes < required %zu", ws_bytes, ar.off);
  return HN_OK;
}

}  // namespace hn

using namespace hn;

extern "C" {

int hn_abi_version(void) { return HN_ABI_VERSION; }
int hn_cluster_status(int device, int acknowledge, hn_cluster_info *info) { return cluster_status(device, acknowledge, info); }
int hn_cluster_config(int device, int enable, int timeout_us) { return cluster_config(device, enable, timeout_us); }
int hn_set_kernel_timers(hn_kernel_timer *timers, int n) {
  if (n < 0 || (n > 0 && timers == nullptr)) return fail(HN_E_SHAPE, "hn_set_kernel_timers: n=%d", n);
  static std::mutex writers;                   // (two threads arming at once: one after the other)
  std::lock_guard<std::mutex> lock(writers);
  __atomic_fetch_add(&g_timer_seq, 1u, __ATOMIC_ACQ_REL);      // odd: readers retry
  __atomic_store_n(&g_timers, n > 0 ? timers : (hn_kernel_timer *)nullptr, __ATOMIC_RELAXED);
  __atomic_store_n(&g_ntimers, n > 0 ? n : 0, __ATOMIC_RELAXED);
  __atomic_fetch_add(&g_timer_seq, 1u, __ATOMIC_RELEASE);      // even again
  return HN_OK;
}
#ifndef HN_BUILD_ID
#define HN_BUILD_ID "unstamped"
#endif
// the marker is also what the host side scans the FILE for (no dlopen: a process that has the old library mapped must still be able
// to read the id of a freshly linked one)
static const char kBuildIdMarker[] = "HN_BUILD_ID=" HN_BUILD_ID;
const char *hn_build_id(void) { return kBuildIdMarker + 12; }
const char *hn_last_error_string(void) { return g_err; }

int hn_context_pitch(int D, int dim_head) { return context_pitch(D, dim_head); }

int hn_fourier_encode_concat(const float *data, int b, int n_axes, const int *spatial, int channels, int num_freq_bands,
                             float max_freq, int fourier, float *ctx, int ld_out, void *stream) {
  return launch_encode(data, HN_F32, b, n_axes, spatial, channels, num_freq_bands, max_freq, fourier, 0, 0.0f, ctx, ld_out,
                       (hipStream_t)stream);
}

int hn_encode_norm(const float *data, int b, int n_axes, const int *spatial, int channels, int num_freq_bands,
                   float max_freq, int fourier, float eps, float *z, int ld_out, void *stream) {
  return launch_encode(data, HN_F32, b, n_axes, spatial, channels, num_freq_bands, max_freq, fourier, 1, eps, z, ld_out,
                       (hipStream_t)stream);
}

int hn_encode_norm_slab(const float *data, int b, int n_axes, const int *spatial, int channels, int num_freq_bands,
                        float max_freq, int fourier, float eps, float *z, int ld_out, int axis0_begin, int axis0_total,
                        void *stream) {
  HN_REQUIRE(axis0_total > 0, HN_E_SHAPE, "encode_norm_slab: axis0_total=%d", axis0_total);
  return launch_encode(data, HN_F32, b, n_axes, spatial, channels, num_freq_bands, max_freq, fourier, 1, eps, z, ld_out,
                       (hipStream_t)stream, -1, 0, axis0_begin, axis0_total);
}

// ------------------------------------------------------------------------------------------------
// Context split over ranks (second axis): a rank attends to ITS tokens only and hands back the normalised
// output of its shard with the softmax statistics; hn_attn_merge_fwd folds the shards of all ranks (the split-KV merge, one
// level up) and finishes the block.
// ------------------------------------------------------------------------------------------------
int hn_attn_partial_fwd(const hn_attn_params *p, const float *x_in, const float *ctx, int ld_ctx, int b, int L, int N, int D,
                        const uint8_t *mask, float *o_part, float *stats, void *workspace, size_t workspace_bytes, void *stream) {
  HN_REQUIRE(p && x_in && ctx && o_part && stats, HN_E_NULL, "attn_partial_fwd: NULL pointer");
  HN_REQUIRE(p->dropout == 0.0f, HN_E_UNSUPPORTED, "attn_partial_fwd: inference only (dropout = %g)", (double)p->dropout);
  HN_REQUIRE(N >= 2 || mask != nullptr, HN_E_UNSUPPORTED, "attn_partial_fwd: a one-token shard has no statistics; give it to hn_attn_fwd whole");
  hipStream_t s = (hipStream_t)stream;
  AttnExt ext;
  memset(&ext, 0, sizeof(ext));
  ext.defer_out = true;
  int rc = attn_fwd_impl(p, x_in, nullptr, 0, ctx, ld_ctx, b, L, N, D, mask, stats, workspace, workspace_bytes, s, nullptr, nullptr,
                         nullptr, false, 0, nullptr, nullptr, &ext);
  if (rc != HN_OK) return rc;
  HN_REQUIRE(ext.o_out != nullptr && !ext.merge_deferred, HN_E_UNSUPPORTED, "attn_partial_fwd: the block did not report its output");
  const int inner = p->heads * p->dim_head;
  if (ext.ldo_out == inner) return launch_copy(o_part, ext.o_out, (long)((size_t)b * L * inner), s);
  HN_HIP_CHECK(hipMemcpy2DAsync(o_part, (size_t)inner * 4, ext.o_out, (size_t)ext.ldo_out * 4, (size_t)inner * 4, (size_t)b * L,
                                hipMemcpyDeviceToDevice, s));
  return HN_OK;
}

size_t hn_attn_merge_workspace_bytes(const hn_attn_params *p, int b, int L) {
  if (!p || b <= 0 || L <= 0) return 0;
  return align_up((size_t)b * L * p->heads * p->dim_head * sizeof(float), 256);
}

int hn_attn_merge_fwd(const hn_attn_params *p, const float *x_in, float *x_out, int residual, const float *o_parts,
                      const float *stats_parts, int n_parts, int b, int L, float *stats, void *workspace, size_t workspace_bytes,
                      void *stream) {
  HN_REQUIRE(p && x_in && x_out && o_parts && stats_parts && p->w_out, HN_E_NULL, "attn_merge_fwd: NULL pointer");
  HN_REQUIRE(n_parts >= 1 && b > 0 && L > 0, HN_E_SHAPE, "attn_merge_fwd: parts=%d b=%d L=%d", n_parts, b, L);
  int rc = check_ws(workspace, workspace_bytes, hn_attn_merge_workspace_bytes(p, b, L), "attn_merge_fwd");
  if (rc != HN_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int inner = p->heads * p->dim_head;
  float *obuf = (float *)workspace;
  const bool vec = p->dim_head % 4 == 0 && (((uintptr_t)o_parts | (uintptr_t)obuf) & 15) == 0;
  const long pieces = (long)b * L * (vec ? inner >> 2 : inner);
  if (vec)
    hipLaunchKernelGGL(attn_merge_parts_kernel<4>, dim3((unsigned)ceil_div_ll(pieces, 256)), dim3(256), 0, s, o_parts, stats_parts, n_parts, b,
                       p->heads, L, p->dim_head, obuf, stats, (long)b * L * inner, (long)b * p->heads * L * 2);
  else
    hipLaunchKernelGGL(attn_merge_parts_kernel<1>, dim3((unsigned)ceil_div_ll(pieces, 256)), dim3(256), 0, s, o_parts, stats_parts, n_parts, b,
                       p->heads, L, p->dim_head, obuf, stats, (long)b * L * inner, (long)b * p->heads * L * 2);
  HN_LAUNCH_CHECK("attn_merge_parts");
  GemmArgs go = gemm_defaults();
  go.A = obuf; go.lda = inner;
  go.W = p->w_out; go.ldw = wo_ld(p);
  go.C = x_out; go.ldc = p->query_dim;
  go.bias = p->b_out;
  go.M = b * L; go.N = p->query_dim; go.K = inner;
  go.act = ACT_LEAKY;
  if (residual) { go.R = x_in; go.ldr = p->query_dim; }
  return launch_gemm(go, s);
}

size_t hn_attn_workspace_bytes(const hn_attn_params *p, int has_ctx, int ld_ctx, int b, int L, int N, int D) {
  AttnPlan pl;
  if (plan_attn(p, has_ctx != 0, ld_ctx, b, L, N, D, nullptr, 0, &pl) != HN_OK) return 0;
  return pl.bytes;
}

int hn_attn_fwd(const hn_attn_params *p, const float *x_in, float *x_out, int residual, const float *ctx, int ld_ctx,
                int b, int L, int N, int D, const uint8_t *mask, float *stats, void *workspace, size_t workspace_bytes,
                void *stream) {
  return attn_fwd_impl(p, x_in, x_out, residual, ctx, ld_ctx, b, L, N, D, mask, stats, workspace, workspace_bytes,
                       (hipStream_t)stream, nullptr, nullptr);
}

int hn_attn_probs(const hn_attn_params *p, const float *x_in, const float *ctx, int ld_ctx, int b, int L, int N, int D,
                  const uint8_t *mask, const float *stats, float *probs, void *workspace, size_t workspace_bytes,
                  void *stream) {
  HN_REQUIRE(p && x_in && stats && probs, HN_E_NULL, "attn_probs: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  AttnPlan pl;
  int rc = plan_attn(p, ctx != nullptr, ld_ctx, b, L, N, D, nullptr, 0, &pl);
  if (rc != HN_OK) return rc;
  if ((rc = check_ws(workspace, workspace_bytes, pl.bytes, "attn_probs")) != HN_OK) return rc;
  if ((rc = plan_attn(p, ctx != nullptr, ld_ctx, b, L, N, D, workspace, workspace_bytes, &pl)) != HN_OK) return rc;
  if (ctx != nullptr && pl.N == 1 && mask == nullptr) return launch_fill(probs, 1.0f, (long)b * p->heads * L, s);
  AttnCoreArgs core;
  if ((rc = attn_prepare(p, pl, x_in, ctx, ld_ctx, b, L, s, &core)) != HN_OK) return rc;
  return launch_probs(core.Q, core.q_b, core.q_h, core.ldq, pl.rank_d ? pl.D : pl.dh, core.Kp, core.k_b, core.k_h, core.ldk,
                      mask, stats, probs, b, p->heads, L, pl.N, s);
}

int hn_attn_importance(const hn_attn_params *p, const float *x_in, const float *ctx, int ld_ctx, int b, int L, int N, int D,
                       const uint8_t *mask, const float *stats, float *importance, void *workspace, size_t workspace_bytes,
                       void *stream) {
  HN_REQUIRE(p && x_in && stats && importance, HN_E_NULL, "attn_importance: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  AttnPlan pl;
  int rc = plan_attn(p, ctx != nullptr, ld_ctx, b, L, N, D, nullptr, 0, &pl);
  if (rc != HN_OK) return rc;
  if ((rc = check_ws(workspace, workspace_bytes, pl.bytes, "attn_importance")) != HN_OK) return rc;
  if ((rc = plan_attn(p, ctx != nullptr, ld_ctx, b, L, N, D, workspace, workspace_bytes, &pl)) != HN_OK) return rc;
  if (ctx != nullptr && pl.N == 1 && mask == nullptr) return launch_fill(importance, 1.0f, (long)b * p->heads, s);
  AttnCoreArgs core;
  if ((rc = attn_prepare(p, pl, x_in, ctx, ld_ctx, b, L, s, &core)) != HN_OK) return rc;
  return launch_importance(core.Q, core.q_b, core.q_h, core.ldq, pl.rank_d ? pl.D : pl.dh, core.Kp, core.k_b, core.k_h,
                           core.ldk, mask, stats, importance, b, p->heads, L, pl.N, s);
}

size_t hn_attn_saved_floats(const hn_attn_params *p, int has_ctx, int ld_ctx, int b, int L, int N, int D, int masked) {
  AttnPlan pl;
  if (plan_attn(p, has_ctx != 0, ld_ctx, b, L, N, D, nullptr, 0, &pl) != HN_OK) return 0;
  return attn_saved_floats(pl, has_ctx != 0, masked != 0 || p->dropout > 0.0f, b, L);
}

int hn_attn_fwd_train(const hn_attn_params *p, const float *x_in, float *x_out, int residual, const float *ctx, int ld_ctx,
                      int b, int L, int N, int D, const uint8_t *mask, float *stats, float *saved, void *workspace,
                      size_t workspace_bytes, void *stream) {
  HN_REQUIRE(stats && saved, HN_E_NULL, "attn_fwd_train: stats and saved are required");
  return attn_fwd_impl(p, x_in, x_out, residual, ctx, ld_ctx, b, L, N, D, mask, stats, workspace, workspace_bytes,
                       (hipStream_t)stream, nullptr, nullptr, saved);
}

size_t hn_attn_bwd_workspace_bytes(const hn_attn_params *p, int has_ctx, int ld_ctx, int b, int L, int N, int D, int masked) {
  AttnPlan pl;
  if (plan_attn(p, has_ctx != 0, ld_ctx, b, L, N, D, nullptr, 0, &pl) != HN_OK) return 0;
  AttnBwdPlan bp;
  if (plan_attn_bwd(p, pl, has_ctx != 0, masked != 0 || p->dropout > 0.0f, b, L, nullptr, 0, &bp) != HN_OK) return 0;
  return bp.bytes;
}

int hn_attn_bwd(const hn_attn_params *p, const float *x_in, const float *x_out, int residual, const float *ctx, int ld_ctx, int b,
                int L, int N, int D, const uint8_t *mask, const float *stats, const float *saved, const float *dy, float *dx,
                const hn_attn_grads *grads, void *workspace, size_t workspace_bytes, void *stream) {
  return attn_bwd_impl(p, x_in, x_out, residual, ctx, ld_ctx, b, L, N, D, mask, stats, saved, dy, dx, grads, workspace,
                       workspace_bytes, (hipStream_t)stream);
}

// ---- training with the context split over ranks, block level (ABI v11; include/healnet_hip.h "Context split: training")
int hn_attn_bwd_cp(const hn_attn_params *p, const float *x_in, const float *x_out, const float *ctx, int ld_ctx, int b, int L, int N,
                   int D, const float *stats, const float *saved, const float *dy, float *dx, const hn_attn_grads *grads,
                   int replicated_owner, void *workspace, size_t workspace_bytes, void *stream) {
  HN_REQUIRE(p && ctx, HN_E_NULL, "attn_bwd_cp: a cross block with its rank's slab of the context");
  HN_REQUIRE(p->dropout == 0.0f && N >= 2, HN_E_UNSUPPORTED, "attn_bwd_cp: dropout=%g N=%d (no dropout, at least two tokens per rank)", (double)p->dropout, N);
  AttnBwdExt ext;
  memset(&ext, 0, sizeof(ext));
  ext.skip_replicated = replicated_owner == 0;
  ext.dx_without_residual = true;
  return attn_bwd_impl(p, x_in, x_out, 1, ctx, ld_ctx, b, L, N, D, nullptr, stats, saved, dy, dx, grads, workspace, workspace_bytes,
                       (hipStream_t)stream, 0, &ext);
}

int hn_attn_saved_part_width(const hn_attn_params *p, int ld_ctx, int b, int L, int N, int D) {
  AttnPlan pl;
  if (plan_attn(p, true, ld_ctx, b, L, N, D, nullptr, 0, &pl) != HN_OK) return 0;
  if (pl.N == 1) return 0;
  return pl.rank_d ? pl.dp : pl.dh;
}

int hn_attn_merge_parts(const float *o_parts, const float *stats_parts, int n_parts, long o_stride, long stats_stride, int b, int heads,
                        int L, int width, float *o, float *stats, void *stream) {
  HN_REQUIRE(o_parts && stats_parts && o && stats, HN_E_NULL, "attn_merge_parts: NULL pointer");
  HN_REQUIRE(n_parts >= 1 && b > 0 && heads > 0 && L > 0 && width > 0 && o_stride >= (long)b * L * heads * width &&
                 stats_stride >= (long)b * heads * L * 2, HN_E_SHAPE, "attn_merge_parts: parts=%d b=%d heads=%d L=%d width=%d", n_parts, b, heads, L, width);
  hipStream_t s = (hipStream_t)stream;
  const int inner = heads * width;
  const bool vec = width % 4 == 0 && (((uintptr_t)o_parts | (uintptr_t)o) & 15) == 0 && o_stride % 4 == 0;
  const long pieces = (long)b * L * (vec ? inner >> 2 : inner);
  if (vec)
    hipLaunchKernelGGL(attn_merge_parts_kernel<4>, dim3((unsigned)ceil_div_ll(pieces, 256)), dim3(256), 0, s, o_parts, stats_parts, n_parts, b,
                       heads, L, width, o, stats, o_stride, stats_stride);
  else
    hipLaunchKernelGGL(attn_merge_parts_kernel<1>, dim3((unsigned)ceil_div_ll(pieces, 256)), dim3(256), 0, s, o_parts, stats_parts, n_parts, b,
                       heads, L, width, o, stats, o_stride, stats_stride);
  HN_LAUNCH_CHECK("attn_merge_parts");
  return HN_OK;
}

int hn_attn_finish_fwd(const hn_attn_params *p, const float *x_in, float *x_out, int residual, int ld_ctx, int b, int L, int N, int D,
                       const float *saved, void *workspace, size_t workspace_bytes, void *stream) {
  HN_REQUIRE(p && x_in && x_out && saved && p->w_out && p->w_kv, HN_E_NULL, "attn_finish_fwd: NULL pointer");
  HN_REQUIRE(p->dropout == 0.0f && N >= 2, HN_E_UNSUPPORTED, "attn_finish_fwd: dropout=%g N=%d", (double)p->dropout, N);
  hipStream_t s = (hipStream_t)stream;
  AttnPlan pl;
  int rc = plan_attn(p, true, ld_ctx, b, L, N, D, nullptr, 0, &pl);
  if (rc != HN_OK) return rc;
  if ((rc = check_ws(workspace, workspace_bytes, pl.bytes, "attn_finish_fwd")) != HN_OK) return rc;
  if ((rc = plan_attn(p, true, ld_ctx, b, L, N, D, workspace, workspace_bytes, &pl)) != HN_OK) return rc;
  const int rows = b * L, inner = pl.inner, h = p->heads;
  const float *O = saved;
  if (pl.rank_d) {      // O = (P z * gamma + beta) W_v^T from the merged context average (as attn_bwd_impl recomputes it)
    float *A = pl.qf;
    if ((rc = launch_head_affine(saved, h * pl.dp, pl.dp, nullptr, 0, 0, p->ctx_gamma, p->ctx_beta, 1.0f, h, pl.D, pl.dp, h * pl.dp, rows, A, s)) != HN_OK) return rc;
    GemmExArgs e = gex(A, (long)h * pl.dp, 1, p->w_kv + (long)inner * pl.D, pl.D, 1, pl.obuf, inner, rows, pl.dh, pl.D, 0);
    e.batch = h; e.strideA = pl.dp; e.strideB = (long)pl.dh * pl.D; e.strideC = pl.dh;
    if ((rc = launch_gemm_ex(e, s, nullptr)) != HN_OK) return rc;
    O = pl.obuf;
  }
  GemmArgs go = gemm_defaults();
  go.A = O; go.lda = inner;
  go.W = p->w_out; go.ldw = wo_ld(p);
  go.C = x_out; go.ldc = p->query_dim;
  go.bias = p->b_out;
  go.M = rows; go.N = p->query_dim; go.K = inner;
  go.act = ACT_LEAKY;
  if (residual) { go.R = x_in; go.ldr = p->query_dim; }
  return launch_gemm(go, s);
}

size_t hn_ff_workspace_bytes(const hn_ff_params *p, int rows) {
  if (!p || p->dim <= 0 || rows <= 0) return 0;
  return ff_ws_bytes(p, rows);
}

// p->dropout > 0 applies the mask of p->rng (the caller passes 0 outside training, as nn.Dropout does in eval mode)
int hn_ff_fwd(const hn_ff_params *p, const float *x_in, float *x_out, int residual, int rows, void *workspace,
              size_t workspace_bytes, void *stream) {
  return ff_fwd_impl(p, x_in, x_out, residual, rows, workspace, workspace_bytes, (hipStream_t)stream, true);
}

int hn_fourier_encode(const float *x, float *out, long n, int num_bands, float max_freq, void *stream) {
  return launch_fourier_encode(x, out, n, num_bands, max_freq, (hipStream_t)stream);
}

int hn_glu_gate(const float *x, float *out, long rows, int hidden, int gate, void *stream) {
  HN_REQUIRE(gate == HN_GATE_SELU || gate == HN_GATE_GELU, HN_E_UNSUPPORTED, "glu_gate: gate=%d", gate);
  return launch_glu_gate(x, out, rows, hidden, gate == HN_GATE_GELU, (hipStream_t)stream);
}

int hn_temperature_softmax(const float *logits, float *probs, long rows, int n, float temperature, void *stream) {
  return launch_temperature_softmax(logits, probs, rows, n, temperature, (hipStream_t)stream);
}

int hn_dropout_mask(float p, hn_rng rng, int is_ff, long rows, int cols, uint8_t *mask, void *stream) {
  HN_REQUIRE(mask && rows > 0 && cols > 0 && p >= 0.0f && p < 1.0f, HN_E_SHAPE, "dropout_mask: p=%g rows=%ld cols=%d", (double)p, rows, cols);
  DropCfg d = drop_of(p, rng, is_ff != 0);
  if (d.thr == 0) return launch_fill_bytes(mask, 1, rows * (long)cols, (hipStream_t)stream);
  return launch_dropout_mask(mask, rows, cols, d, (hipStream_t)stream);
}

int hn_head_fwd(const float *x, int b, int L, int d, const float *norm_w, const float *norm_b, const float *w,
                const float *bias, int out_dims, float *logits, void *stream) {
  return launch_head(x, b, L, d, norm_w, norm_b, w, bias, out_dims, logits, (hipStream_t)stream);
}

size_t hn_ff_bwd_workspace_bytes(const hn_ff_params *p, int rows) {
  if (!p || p->dim <= 0 || rows <= 0) return 0;
  FFBwdPlan pl;
  plan_ff_bwd(p, rows, nullptr, 0, &pl);
  return pl.bytes;
}

int hn_ff_bwd(const hn_ff_params *p, const float *x_in, const float *dy, float *dx, int residual, int rows,
              const hn_ff_grads *grads, void *workspace, size_t workspace_bytes, void *stream) {
  return ff_bwd_impl(p, x_in, dy, dx, residual, rows, grads, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t hn_head_bwd_workspace_bytes(int b, int d, int out_dims) { return align_up(head_bwd_scratch_floats(b, d, out_dims) * sizeof(float), 256); }

int hn_head_bwd(const float *x, int b, int L, int d, const float *norm_w, const float *norm_b, const float *w, int out_dims,
                const float *dlogits, float *dx, float *d_norm_w, float *d_norm_b, float *d_w, float *d_bias, void *workspace,
                size_t workspace_bytes, void *stream) {
  int rc = check_ws(workspace, workspace_bytes, hn_head_bwd_workspace_bytes(b, d, out_dims), "head_bwd");
  if (rc != HN_OK) return rc;
  return launch_head_bwd(x, b, L, d, norm_w, norm_b, w, out_dims, dlogits, dx, d_norm_w, d_norm_b, d_w, d_bias,
                         (float *)workspace, (hipStream_t)stream);
}

int hn_surv_nll(const float *logits, const int64_t *y, const float *censorship, const float *class_weights, int b, int n_bins,
                float alpha, float eps, float grad_scale, float *loss, float *dlogits, float *hazards, float *survival,
                float *risk, void *stream) {
  return launch_surv_nll(logits, (const long long *)y, censorship, class_weights, b, n_bins, alpha, eps, grad_scale, loss, dlogits,
                         hazards, survival, risk, (hipStream_t)stream);
}

size_t hn_l1_adam_workspace_bytes(void) { return L1_ADAM_PARTIALS * sizeof(float); }

int hn_l1_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, long n, double l1, double grad_scale,
                    double lr, double beta1, double beta2, double eps, int step, float *reg_loss, void *workspace,
                    size_t workspace_bytes, void *stream) {
  { const int prc = cluster_poll("hn_l1_adam_step", (hipStream_t)stream); if (prc != HN_OK) return prc; }
  int rc = check_ws(workspace, workspace_bytes, hn_l1_adam_workspace_bytes(), "l1_adam");
  if (rc != HN_OK) return rc;
  return launch_l1_adam(params, grads, exp_avg, exp_avg_sq, n, l1, grad_scale, lr, beta1, beta2, eps, step, reg_loss,
                        (float *)workspace, (hipStream_t)stream);
}

static size_t impl_fusion_workspace_bytes(const hn_model *model, const hn_modality_input *inputs, int b) {
  FusionPlan fp;
  if (plan_fusion(model, inputs, b, nullptr, 0, &fp, true) != HN_OK) return 0;
  return fp.bytes;
}

static int impl_fusion_forward(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                      int return_embeddings, float *out, float **attn_stats, float **x_trace, void *workspace,
                      size_t workspace_bytes, void *stream, hn_profile *prof, const hn_context_split *cp = nullptr) {
  hipStream_t s = (hipStream_t)stream;
  HN_REQUIRE(out, HN_E_NULL, "fusion: out is NULL");
  // cp: the context of the modalities in cp->split_mask is split over cp->n_parts ranks (hn_fusion_forward_cp): in[i] is this
  // rank's slab, the cross blocks of such a modality exchange their (output, statistics) pairs through cp->exchange
  auto is_split = [&](int i) { return cp != nullptr && ((cp->split_mask >> i) & 1u) != 0; };
  FusionPlan fp;
  int rc = plan_fusion(m, in, b, nullptr, 0, &fp, true);
  if (rc != HN_OK) return rc;
  if ((rc = check_ws(workspace, workspace_bytes, fp.bytes, "fusion")) != HN_OK) return rc;
  if ((rc = plan_fusion(m, in, b, workspace, workspace_bytes, &fp, true)) != HN_OK) return rc;
  const int M = m->n_modalities, L = m->l_c, d = m->l_d;
  const size_t xbytes = (size_t)b * L * d * sizeof(float);
  if (prof) prof->n_recorded = 0;

  // K1 once per forward: the normalised context of every present modality (layer independent)
  // (measured and dropped, round 4: the long modalities' encode on a side stream beside the one-token prelude -- the two
  // HBM-bound kernels slow each other down (skinny GEMMs 10 -> 18 us) and the join costs what is left: -1 % at cfg2 b = 32)
  for (int i = 0; i < M; ++i) {
    if (!in[i].data) continue;
    HN_REQUIRE(!is_split(i) || (!fp.bf16[i] && fp.N[i] >= 2), HN_E_UNSUPPORTED,
               "fusion: a split modality needs the fp32 core and at least two tokens per rank (modality %d: N=%ld)", i, (long)fp.N[i]);
    if (fp.bf16[i]) {
      uint16_t *zb = (uint16_t *)fp.z[i], *zT = zb + (size_t)b * fp.Np[i] * bf16_row_slots(fp.ldz[i], fp.ns[i]);
      rc = launch_encode_bf16ctx(in[i].data, in[i].dtype, b, m->num_spatial_axes[i], in[i].spatial, m->channel_dims[i],
                                 m->num_freq_bands, m->max_freq, m->fourier_encode_data, 1e-5f, zb, zT, fp.Np[i], fp.ldz[i],
                                 fp.ns[i], s);
    } else {
      rc = launch_encode(in[i].data, in[i].dtype, b, m->num_spatial_axes[i], in[i].spatial, m->channel_dims[i], m->num_freq_bands,
                         m->max_freq, m->fourier_encode_data, 1, 1e-5f, fp.z[i], fp.ldz[i], s, fp.ones[i] ? fp.ldz[i] - 1 : -1,
                         fp.pack[i], is_split(i) ? cp->axis0_begin[i] : 0, is_split(i) ? cp->axis0_total[i] : 0);
      if (rc == HN_OK && fp.z16[i]) rc = launch_rows_to_bf16(fp.z[i], fp.ldz[i], (long)b * fp.N[i], fp.D[i], fp.z16[i], s);
    }
    if (rc != HN_OK) return rc;
  }
  // one-token modalities: all layers' block outputs in two batched launches (a key mask routes them through the general path)
  bool tab_ready[16];
  for (int i = 0; i < M; ++i) {
    tab_ready[i] = false;
    if (!in[i].data || !fp.tab_ahead[i] || mask != nullptr) continue;
    const hn_attn_params &a0 = m->cross_attn[i];
    const int inner = a0.heads * a0.dim_head;
    GemmSkinnyMulti gv, gy;
    memset(&gv, 0, sizeof(gv));
    memset(&gy, 0, sizeof(gy));
    gv.nz = gy.nz = m->depth;
    gv.lda = fp.ldz[i]; gv.ldw = fp.D[i]; gv.ldc = inner; gv.M = b; gv.N = inner; gv.K = fp.D[i];
    gv.pro = a0.ctx_gamma ? PRO_AFFINE : PRO_NONE; gv.act = ACT_NONE;
    gy.lda = inner; gy.ldw = inner; gy.ldc = a0.query_dim; gy.M = b; gy.N = a0.query_dim; gy.K = inner;
    gy.pro = PRO_NONE; gy.act = ACT_LEAKY;
    for (int layer = 0; layer < m->depth; ++layer) {
      const hn_attn_params &al = m->cross_attn[layer * M + i];
      HN_REQUIRE(al.w_kv && al.w_out, HN_E_NULL, "attn: weight pointer is NULL");
      gv.A[layer] = fp.z[i]; gv.W[layer] = al.w_kv + (long)inner * fp.D[i]; gv.gamma[layer] = al.ctx_gamma; gv.beta[layer] = al.ctx_beta;
      gv.C[layer] = fp.tabv[i] + (size_t)layer * b * inner;
      gy.A[layer] = gv.C[layer]; gy.W[layer] = al.w_out; gy.bias[layer] = al.b_out;
      gy.C[layer] = fp.taby[i] + (size_t)layer * b * a0.query_dim;
    }
    if ((rc = launch_gemm_skinny_multi(gv, s)) != HN_OK) return rc;
    if ((rc = launch_gemm_skinny_multi(gy, s)) != HN_OK) return rc;
    tab_ready[i] = true;
  }

  // The latent array moves through a chain of buffers instead of being updated in place: the block in front of an
  // attention block writes straight into that block's x_trace slot (the input hn_attn_probs re-reads later), so
  // keeping the trace costs no copy.  Without trace slots every block works in place on fp.x as before.
  static thread_local Step steps[kMaxSteps];
  const int nsteps = build_schedule(m, in, skip_self_on_missing, steps, kMaxSteps);
  HN_REQUIRE(nsteps >= 0, HN_E_UNSUPPORTED, "fusion: more than %d blocks", kMaxSteps);
  auto slot_of = [&](const Step &st) { return st.layer * (M + 1) + (st.kind == STEP_CROSS_ATTN ? st.m : M); };
  auto input_buffer = [&](int k) -> float * {      // where step k wants to find x
    if (k < nsteps && x_trace && (steps[k].kind == STEP_CROSS_ATTN || steps[k].kind == STEP_SELF_ATTN) && x_trace[slot_of(steps[k])])
      return x_trace[slot_of(steps[k])];
    return fp.x;
  };
  float *cur = input_buffer(0);
  bool broadcast_done = false;      // the latent broadcast (:225) + the flags rode on a vfold launch
  // folded value projections for the chains that merge the split partials of a shared-context block themselves (one launch per
  // modality and forward; only when the chain is the route)
  bool vmerge[16], qfolded[16];
  for (int i = 0; i < M; ++i) {
    vmerge[i] = false; qfolded[i] = false;
    if (!in[i].data || !fp.wvf[i] || !fp.chain || merge_chain_disabled() || chain_disabled()) continue;
    VfoldMulti vf;
    memset(&vf, 0, sizeof(vf));
    const hn_attn_params &a0 = m->cross_attn[i];
    vf.n = m->depth; vf.D = fp.D[i]; vf.heads = a0.heads; vf.dh = a0.dim_head; vf.pack_ks = fp.pack[i];
    vf.out = fp.wvf[i]; vf.out_stride = (long)a0.heads * a0.dim_head * 16;
    for (int layer = 0; layer < m->depth; ++layer) {
      const hn_attn_params &al = m->cross_attn[layer * M + i];
      HN_REQUIRE(al.w_kv, HN_E_NULL, "attn: weight pointer is NULL");
      vf.w_v[layer] = al.w_kv + (long)al.heads * al.dim_head * fp.D[i]; vf.gamma[layer] = al.ctx_gamma; vf.beta[layer] = al.ctx_beta;
    }
    qfolded[i] = fp.wqf[i] != nullptr && fp.cbound != nullptr && !qfold_chain_disabled();
    if (qfolded[i]) {
      AttnPlan p0;
      if ((rc = plan_attn(&a0, true, fp.ldz[i], b, L, fp.N[i], fp.D[i], nullptr, 0, &p0)) != HN_OK) return rc;
      vf.cscale = p0.cscale; vf.l_d = d; vf.qout = fp.wqf[i]; vf.qout_stride = (long)128 * d;
      for (int layer = 0; layer < m->depth; ++layer) {
        const hn_attn_params &al = m->cross_attn[layer * M + i];
        HN_REQUIRE(al.w_q, HN_E_NULL, "attn: weight pointer is NULL");
        vf.w_k[layer] = al.w_kv; vf.w_q[layer] = al.w_q;
      }
    }
    static const bool no_bc_role = tuning_env("HN_NO_VFOLD_BROADCAST") != nullptr;      // route switch (A/B)
    if (!broadcast_done && !no_bc_role && ((long)L * d) % 4 == 0 && al16(m->latents) && al16(cur)) {
      vf.bc_src = m->latents; vf.bc_dst = cur; vf.bc_per = (long)L * d; vf.bc_total = (long)L * d * b;
      vf.bc_zero = fp.flags; vf.bc_nzero = m->depth * M + CHAIN_XCHG_FLAGS;
      broadcast_done = true;
    }
    if ((rc = launch_vfold(vf, s)) != HN_OK) return rc;
    vmerge[i] = true;
  }

  if (!broadcast_done && (rc = launch_broadcast_rows(m->latents, cur, (long)L * d, b, s, fp.flags, m->depth * M + CHAIN_XCHG_FLAGS)) != HN_OK) return rc;   // :225 (+ the bound / cluster flags)
  int chain_seq = 0;
  const bool head = m->final_classifier_head && !return_embeddings;
  const bool use_chain = fp.chain && !chain_disabled();      // HN_NO_CHAIN: development switch, the unfused launch sequence

  float *stats_override = nullptr;      // (context split: the statistics of a split block go to the exchange buffer)
  auto run_attn = [&](const Step &st, const float *xin, float *xout, AttnExt *ext) -> int {
    const int layer = st.layer, i = st.m;
    if (st.kind == STEP_SELF_ATTN)                                                              // :241-245
      return attn_fwd_impl(&m->self_attn[layer], xin, xout, 1, nullptr, 0, b, L, L, d, nullptr,
                           attn_stats ? attn_stats[slot_of(st)] : nullptr, fp.op_ws, fp.op_ws_bytes, s, nullptr, nullptr, nullptr, false, 0,
                           nullptr, nullptr, ext);
    const hn_attn_params *ap = &m->cross_attn[layer * M + i];
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof && i == fp.dominant && prof->n_recorded < prof->n_events) {
      e0 = (hipEvent_t)prof->ev_start[prof->n_recorded];
      e1 = (hipEvent_t)prof->ev_stop[prof->n_recorded];
      prof->n_recorded++;
    }
    Bf16Context bc;
    bc.zb = (const uint16_t *)fp.z[i]; bc.zT = bc.zb + (size_t)b * fp.Np[i] * bf16_row_slots(fp.ldz[i], fp.ns[i]);
    bc.Np = fp.Np[i]; bc.DV = fp.ldz[i]; bc.ns = fp.ns[i];
    return attn_fwd_impl(ap, xin, xout, 1, fp.z[i], fp.ldz[i], b, L, fp.N[i], fp.D[i], mask,
                         stats_override ? stats_override : (attn_stats ? attn_stats[slot_of(st)] : nullptr), fp.op_ws, fp.op_ws_bytes, s, e0, e1, nullptr,
                         fp.ones[i], fp.pack[i], fp.bf16[i] ? &bc : nullptr, fp.flags + layer * M + i, ext,
                         fp.z16[i]);
  };
  auto ff_of = [&](const Step &st) { return st.kind == STEP_CROSS_FF ? &m->cross_ff[st.layer * M + st.m] : &m->self_ff[st.layer]; };
  auto is_attn = [](const Step &st) { return st.kind == STEP_CROSS_ATTN || st.kind == STEP_SELF_ATTN; };
  // one-token cross block whose output vectors were computed ahead of the layer loop
  auto is_tab = [&](const Step &st) { return st.kind == STEP_CROSS_ATTN && tab_ready[st.m]; };

  bool q_done = false, kv_done = false, qf_done = false;      // projections of the attention block at `k` already produced by the chain in front of it
  const bool staged = m->l_d_valid > 0;      // staged model: every LayerNorm of the latent side runs inside a chain (valid width)
  // projections of the attention block at step kn (if it is one that needs them) as the last stages of chain `ca`
  auto add_next_proj = [&](ChainArgs &ca, int kn) -> int {
    q_done = kv_done = qf_done = false;
    if (!(kn < nsteps && is_attn(steps[kn]) && !is_tab(steps[kn]))) return HN_OK;
    const Step &sn = steps[kn];
    const bool self = sn.kind == STEP_SELF_ATTN;
    const hn_attn_params *an = self ? &m->self_attn[sn.layer] : &m->cross_attn[sn.layer * M + sn.m];
    AttnPlan pn;
    int rc2 = plan_attn(an, !self, self ? 0 : fp.ldz[sn.m], b, L, self ? L : fp.N[sn.m], self ? d : fp.D[sn.m], nullptr, 0, &pn);
    if (rc2 != HN_OK) return rc2;
    const bool one_token = !self && fp.N[sn.m] == 1 && mask == nullptr;
    if (!one_token && pn.dh == pn.dhp && (pn.inner % 128 == 0 || staged_attn(an)) && pn.inner % 16 == 0 && pn.inner <= 512 &&
        an->query_dim == d && an->w_q && an->w_kv && chain_proj_aligned(an)) {
      ca.p_nw = an->norm_w; ca.p_nb = an->norm_b;
      ca.nq = up128(pn.inner); ca.q_cols = pn.inner; ca.wq = an->w_q; ca.Q = fp.cq; ca.ldq = pn.inner;
      ca.alpha_q = pn.rank_d ? 1.0f : pn.cscale;       // the rank-D binding scales in its query fold
      q_done = true;
      if (self) { ca.nkv = up128(2 * pn.inner); ca.kv_cols = 2 * pn.inner; ca.wkv = an->w_kv; ca.KV = fp.ckv; ca.ldkv = 2 * pn.inner; kv_done = true; }
      // shared-context block whose query fold was staged (vfold launch): the Q stage projects 128 instead of `inner` columns and
      // leaves the folded, packed query with its score bounds -- qfold's launch and the wider projection disappear
      // (the consumer runs with use_bound = true and this forward's pre-zeroed flag: inference, no dropout)
      if (!self && qfolded[sn.m] && qfold_core_ok(an, pn, fp.pack[sn.m], L) && !staged && mask == nullptr && !fp.bf16[sn.m] && fp.ones[sn.m]) {
        ca.nq = 128; ca.q_cols = 128; ca.wq = fp.wqf[sn.m] + (size_t)sn.layer * 128 * d; ca.Q = nullptr; ca.ldq = 0; ca.alpha_q = 1.0f;
        ca.qf = fp.cq; ca.qf_bound = fp.cbound; ca.qf_flag = fp.flags + sn.layer * M + sn.m; ca.qf_heads = an->heads; ca.qf_D = pn.D;
        qf_done = true;
      }
    }
    return HN_OK;
  };
  auto launch_chain = [&](ChainArgs &ca) -> int {
    ca.rows = b * L; ca.L = L; ca.dv = m->l_d_valid;
    ca.xchg = fp.xchg; ca.xflags = fp.flags + m->depth * M; ca.seq = ++chain_seq;
    return launch_latent_chain(ca, s);
  };
  // ---- layer chains (lchain.hip): the whole latent side between two shared-context cores as ONE launch, the latent self-attention
  // inside it.  All or nothing per forward: every step must be a one-token block computed ahead, a shared-context block whose merge
  // and query fold the chains take over, or a latent self-attention block of 8 heads x 64 -- each followed by its feed-forward block.
  {
    int dev = 0;
    HN_HIP_CHECK(hipGetDevice(&dev));
    static const bool force_small = getenv("HN_FORCE_SELF_IN_CHAIN") != nullptr;      // route switch (tests): also below the size gate
    bool layer_ok = use_chain && latent_layer_enabled() && cluster_enabled(dev) && !staged && mask == nullptr && cp == nullptr &&
                    L == 128 && d == 128 && fp.lk != nullptr &&
                    (b * 8 > 128 || force_small) && (b + 7) / 8 * 64 + 1 <= CHAIN_XCHG_FLAGS && nsteps >= 2 && nsteps % 2 == 0 && al16(cur);
    int max_seg = 0, max_blk = 0;
    for (int k = 0, nseg = 0, nself = 0, nblk = 0; layer_ok && k < nsteps; k += 2) {
      const Step &st = steps[k];
      layer_ok = is_attn(st) && !is_attn(steps[k + 1]);
      if (!layer_ok) break;
      const hn_ff_params *fq = ff_of(steps[k + 1]);
      layer_ok = fq->dim == d && fq->w1 && fq->b1 && fq->w2 && fq->b2 && chain_ff_aligned(fq) && al16(fq->b1) && al16(fq->b2) &&
                 (fq->norm_w == nullptr) == (fq->norm_b == nullptr);
      if (!layer_ok) break;
      ChainArgs scratch;
      memset(&scratch, 0, sizeof(scratch));
      if (st.kind == STEP_SELF_ATTN) {
        const hn_attn_params *an = &m->self_attn[st.layer];
        if ((rc = add_next_proj(scratch, k)) != HN_OK) return rc;
        nblk += latent_layer_segment_blocks(0, 1) - latent_layer_segment_blocks(0, 0) + latent_layer_segment_blocks(4, 0);      // projections + core on the segment in front, this one's out-projection
        layer_ok = nseg > 0 && nself < LAYER_KV_SLOTS && q_done && kv_done && an->heads == 8 && an->dim_head == 64 && scratch.nq == 512 && scratch.nkv == 1024 &&
                   an->w_out && an->b_out && chain_out_aligned(an) && al16(an->b_out) && (an->norm_w == nullptr) == (an->norm_b == nullptr);
        ++nself;
      } else if (is_tab(st)) {
        layer_ok = m->cross_attn[st.layer * M + st.m].query_dim == d;
        nblk += latent_layer_segment_blocks(2, 0);
      } else {
        const hn_attn_params *an = &m->cross_attn[st.layer * M + st.m];
        AttnPlan pn;
        if ((rc = plan_attn(an, true, fp.ldz[st.m], b, L, fp.N[st.m], fp.D[st.m], nullptr, 0, &pn)) != HN_OK) return rc;
        if ((rc = add_next_proj(scratch, k)) != HN_OK) return rc;
        layer_ok = qf_done && vmerge[st.m] && pn.rank_d && pn.ones && pn.dp == 16 && pn.nsplit <= CHAIN_MERGE_MAX_SPLITS && an->heads <= 8 &&
                   (pn.dh == 16 || pn.dh == 32 || pn.dh == 64) && pn.inner == 512 && an->heads * pn.dh == 512 && scratch.qf_heads == 8 &&
                   an->w_out && an->b_out && chain_out_aligned(an) && al16(an->b_out);
        nblk += latent_layer_segment_blocks(0, 2) - latent_layer_segment_blocks(0, 0);      // the query fold closes the launch in front
        if (nblk > max_blk) max_blk = nblk;
        nseg = 0; nself = 0;                 // (a launch boundary: the core runs between two layer chains)
        nblk = latent_layer_segment_blocks(3, 0);
      }
      ++nseg;
      if (nseg > max_seg) max_seg = nseg;
      if (nblk > max_blk) max_blk = nblk;
    }
    q_done = kv_done = qf_done = false;
    if (layer_ok && max_seg <= LSEG_MAX && max_blk + 6 <= LAYER_MAXBLK) {
      LayerChainArgs la;
      memset(&la, 0, sizeof(la));
      int nself = 0;
      auto begin_launch = [&]() {
        memset(&la, 0, sizeof(la));
        la.b = b; la.x_in = cur;
        la.kbuf = fp.lk; la.vtbuf = fp.lvt; la.kv_stride = (long)b * 8 * 128 * 64;
        la.xflags = fp.flags + m->depth * M; la.flag_count = CHAIN_XCHG_FLAGS;
        nself = 0;
      };
      // x after the segment of steps (k, k + 1) is the input of step k + 2: kept where hn_attn_probs re-reads it (its trace slot)
      // when there is one; otherwise it only leaves LDS at the end of a launch (in place: a workgroup reads and writes its own rows)
      float *nxt = cur;
      auto flush = [&]() -> int {
        if (la.nseg == 0) return HN_OK;
        la.seg[la.nseg - 1].x_out = nxt;
        cur = nxt;
        la.seq = chain_seq + 1;
        chain_seq += nself;
        return launch_latent_layer(la, s);
      };
      begin_launch();
      for (int k = 0; k < nsteps; k += 2) {
        const Step &st = steps[k];
        const hn_ff_params *fq = ff_of(steps[k + 1]);
        LSeg sg;
        memset(&sg, 0, sizeof(sg));
        if (st.kind == STEP_SELF_ATTN) {
          const hn_attn_params *an = &m->self_attn[st.layer];
          AttnPlan pn;
          if ((rc = plan_attn(an, false, 0, b, L, L, d, nullptr, 0, &pn)) != HN_OK) return rc;
          LSeg &pv = la.seg[la.nseg - 1];    // the segment in front projects for this block and runs its core
          pv.proj = 1; pv.kv_slot = nself++; pv.alpha_q = pn.cscale;
          pv.stats = attn_stats ? attn_stats[slot_of(st)] : nullptr;
          pv.p_nw = an->norm_w; pv.p_nb = an->norm_b; pv.wq = an->w_q; pv.wkv = an->w_kv;
          sg.head = 4; sg.w_out = an->w_out; sg.b_out = an->b_out;
        } else if (is_tab(st)) {
          const hn_attn_params *an = &m->cross_attn[st.layer * M + st.m];
          sg.head = 2; sg.y = fp.taby[st.m] + (size_t)st.layer * b * an->query_dim;
        } else {
          const hn_attn_params *an = &m->cross_attn[st.layer * M + st.m];
          ChainArgs ca;
          memset(&ca, 0, sizeof(ca));
          if ((rc = add_next_proj(ca, k)) != HN_OK) return rc;      // (sets the folded-query fields; qf_done)
          if (la.nseg > 0) {
            LSeg &pv = la.seg[la.nseg - 1];
            pv.proj = 2; pv.p_nw = ca.p_nw; pv.p_nb = ca.p_nb; pv.wq = ca.wq;
            la.qf = ca.qf; la.qf_bound = ca.qf_bound; la.qf_flag = ca.qf_flag; la.qf_D = ca.qf_D;
            if ((rc = flush()) != HN_OK) return rc;
          } else {                           // the forward starts with this block: its query fold is a chain of its own
            ca.x_in = cur;
            if ((rc = launch_chain(ca)) != HN_OK) return rc;
          }
          AttnExt ext = {fp.cq, fp.ckv, false, false, true, nullptr, 0, false, false, nullptr, nullptr, nullptr, 0, 0, 0};
          ext.qf = fp.cq; ext.qf_bound = fp.cbound; ext.qf_done = true; ext.allow_defer_merge = true;
          if ((rc = run_attn(st, cur, nullptr, &ext)) != HN_OK) return rc;
          HN_REQUIRE(ext.merge_deferred, HN_E_UNSUPPORTED, "fusion: the shared-context block of layer %d, modality %d did not leave its merge to the chain", st.layer, st.m);
          begin_launch();
          la.Opart = ext.opart; la.Mpart = ext.mpart; la.Lpart = ext.lpart; la.nsplit = ext.nsplit; la.Lp = ext.Lp;
          la.heads = an->heads; la.dh = an->dim_head;
          la.wvf = fp.wvf[st.m] + (size_t)st.layer * an->heads * an->dim_head * 16;
          la.stats3 = attn_stats ? attn_stats[slot_of(st)] : nullptr;
          sg.head = 3; sg.w_out = an->w_out; sg.b_out = an->b_out;
        }
        sg.gate = fq->gate; sg.f_nw = fq->norm_w; sg.f_nb = fq->norm_b; sg.w1 = fq->w1; sg.b1 = fq->b1; sg.w2 = fq->w2; sg.b2 = fq->b2;
        nxt = input_buffer(k + 2);
        if (nxt != cur) sg.x_out = nxt;      // a trace slot (or back to the plan's buffer behind one)
        la.seg[la.nseg++] = sg;
      }
      if ((rc = flush()) != HN_OK) return rc;
      q_done = kv_done = qf_done = false;
      if (head) return launch_head(cur, b, L, d, m->head_norm_w, m->head_norm_b, m->head_w, m->head_b, m->out_dims, out, s, m->l_d_valid);
      return launch_copy(out, cur, (long)((xbytes) / sizeof(float)), s);
    }
  }
  if (staged && use_chain && nsteps > 0 && is_attn(steps[0]) && !is_tab(steps[0])) {      // the first block's projections: a chain of their own
    ChainArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.x_in = cur;
    if ((rc = add_next_proj(ca, 0)) != HN_OK) return rc;
    if (q_done && (rc = launch_chain(ca)) != HN_OK) return rc;
  }
  for (int k = 0; k < nsteps;) {
    const Step &st = steps[k];
    if (!is_attn(st)) {                      // a feed-forward block not absorbed by a chain
      float *dst = input_buffer(k + 1);
      if (staged && use_chain && chain_ff_aligned(ff_of(st)) && al16(cur) && al16(dst)) {      // ... is a chain without a head
        const hn_ff_params *fpar = ff_of(st);
        ChainArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.x_in = cur; ca.x_out = dst; ca.head = 0;
        ca.has_ff = 1; ca.gate = fpar->gate; ca.f_nw = fpar->norm_w; ca.f_nb = fpar->norm_b;
        ca.w1 = fpar->w1; ca.b1 = fpar->b1; ca.w2 = fpar->w2; ca.b2 = fpar->b2;
        if ((rc = add_next_proj(ca, k + 1)) != HN_OK) return rc;
        if ((rc = launch_chain(ca)) != HN_OK) return rc;
        cur = dst; ++k;
        continue;
      }
      if ((rc = ff_fwd_impl(ff_of(st), cur, dst, 1, b * L, fp.op_ws, fp.op_ws_bytes, s)) != HN_OK) return rc;
      cur = dst; ++k;
      continue;
    }
    const hn_attn_params *ap = st.kind == STEP_SELF_ATTN ? &m->self_attn[st.layer] : &m->cross_attn[st.layer * M + st.m];
    const int inner = ap->heads * ap->dim_head;
    // The chain behind this block: its out-projection (or the one-token broadcast add), the feed-forward block that follows
    // (healnet.py:237 / :245) and the projections of the attention block after that.
    const bool ff_next = k + 1 < nsteps && !is_attn(steps[k + 1]);
    // (a model configured with dropout runs the same chain: nothing is dropped on this entry point)
    bool fuse = use_chain && ff_next && ff_of(steps[k + 1])->dim == d &&
                chain_ff_aligned(ff_of(steps[k + 1])) && al16(cur) && al16(input_buffer(k + 2));
    if (fuse && !is_tab(st)) fuse = ap->query_dim == d && (inner % 128 == 0 || staged_attn(ap)) && inner % 16 == 0 && inner <= 512 &&
                                    chain_out_aligned(ap) && !(st.kind == STEP_CROSS_ATTN && fp.N[st.m] == 1 && mask == nullptr);
    if (!fuse) {
      HN_REQUIRE(!(st.kind == STEP_CROSS_ATTN && is_split(st.m)), HN_E_UNSUPPORTED,
                 "fusion: the context split inside the fused forward needs the chain's shapes (modality %d): use the block-level entry points", st.m);
      float *dst = input_buffer(k + 1);
      if (is_tab(st)) rc = launch_add_row_broadcast(fp.taby[st.m] + (size_t)st.layer * b * ap->query_dim, cur, dst, b, L, ap->query_dim, s);
      else {
        AttnExt ext = {fp.cq, fp.ckv, q_done, kv_done, false, nullptr, 0, false, false, nullptr, nullptr, nullptr, 0, 0, 0};
        ext.qf = fp.cq; ext.qf_bound = fp.cbound; ext.qf_done = qf_done;
        rc = run_attn(st, cur, dst, (q_done || kv_done) ? &ext : nullptr);
      }
      if (rc != HN_OK) return rc;
      cur = dst; ++k; q_done = kv_done = qf_done = false;
      continue;
    }
    ChainArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.x_in = cur;
    if (is_tab(st)) {
      ca.head = 2; ca.y = fp.taby[st.m] + (size_t)st.layer * b * ap->query_dim;
    } else {
      AttnExt ext = {fp.cq, fp.ckv, q_done, kv_done, true, nullptr, 0, false, false, nullptr, nullptr, nullptr, 0, 0, 0};
      ext.qf = fp.cq; ext.qf_bound = fp.cbound; ext.qf_done = qf_done;
      const bool split = st.kind == STEP_CROSS_ATTN && is_split(st.m);
      ext.allow_defer_merge = st.kind == STEP_CROSS_ATTN && vmerge[st.m] && inner % 128 == 0 && !split;
      const long n_o = (long)b * L * inner, n_s = (long)b * ap->heads * L * 2;
      if (split) stats_override = cp->local + n_o;
      rc = run_attn(st, cur, nullptr, &ext);
      stats_override = nullptr;
      if (rc != HN_OK) return rc;
      if (split) {
        // this rank's (normalised output | statistics) -> one all-gather -> every rank folds all parts in rank order into the block's
        // O buffer, and the chain carries on from there (out-projection, feed-forward, next projections)
        HN_REQUIRE(!ext.merge_deferred && ext.o_out && ext.ldo_out == inner, HN_E_UNSUPPORTED, "fusion: split block did not report its output");
        if ((rc = launch_copy(cp->local, ext.o_out, n_o, s)) != HN_OK) return rc;
        HN_REQUIRE(cp->exchange(cp->user, (int)(n_o + n_s), stream) == 0, HN_E_HIP, "fusion_cp: the caller's exchange callback failed (layer %d, modality %d)",
                   st.layer, st.m);
        float *st_out = attn_stats ? attn_stats[slot_of(st)] : nullptr;
        const bool vec = ap->dim_head % 4 == 0 && (((uintptr_t)cp->parts | (uintptr_t)ext.o_out) & 15) == 0 && (n_o + n_s) % 4 == 0;
        const long pieces = (long)b * L * (vec ? inner >> 2 : inner);
        if (vec)
          hipLaunchKernelGGL(attn_merge_parts_kernel<4>, dim3((unsigned)ceil_div_ll(pieces, 256)), dim3(256), 0, s, cp->parts, cp->parts + n_o,
                             cp->n_parts, b, ap->heads, L, ap->dim_head, (float *)ext.o_out, st_out, n_o + n_s, n_o + n_s);
        else
          hipLaunchKernelGGL(attn_merge_parts_kernel<1>, dim3((unsigned)ceil_div_ll(pieces, 256)), dim3(256), 0, s, cp->parts, cp->parts + n_o,
                             cp->n_parts, b, ap->heads, L, ap->dim_head, (float *)ext.o_out, st_out, n_o + n_s, n_o + n_s);
        HN_LAUNCH_CHECK("attn_merge_parts");
      }
      ca.inner_o = up128(inner); ca.o_cols = inner; ca.w_out = ap->w_out; ca.b_out = ap->b_out;
      if (ext.merge_deferred) {              // the chain merges the core's split partials and applies the value projection itself
        ca.head = 3; ca.Opart = ext.opart; ca.Mpart = ext.mpart; ca.Lpart = ext.lpart;
        ca.nsplit = ext.nsplit; ca.Lp = ext.Lp; ca.dp = ext.dp; ca.heads = ap->heads; ca.dh = ap->dim_head;
        ca.wvf = fp.wvf[st.m] + (size_t)st.layer * ap->heads * ap->dim_head * 16;
        ca.stats = attn_stats ? attn_stats[slot_of(st)] : nullptr;
      } else {
        ca.head = 1; ca.O = ext.o_out; ca.ldo = ext.ldo_out;
      }
    }
    const hn_ff_params *fpar = ff_of(steps[k + 1]);
    HN_REQUIRE(fpar->w1 && fpar->b1 && fpar->w2 && fpar->b2, HN_E_NULL, "ff: weight pointer is NULL");
    ca.has_ff = 1; ca.gate = fpar->gate; ca.f_nw = fpar->norm_w; ca.f_nb = fpar->norm_b;
    ca.w1 = fpar->w1; ca.b1 = fpar->b1; ca.w2 = fpar->w2; ca.b2 = fpar->b2;
    // projections of the attention block after the feed-forward block
    if ((rc = add_next_proj(ca, k + 2)) != HN_OK) return rc;
    float *dst = input_buffer(k + 2);
    ca.x_out = dst;
    if ((rc = launch_chain(ca)) != HN_OK) return rc;
    cur = dst; k += 2;
  }
  if (head) return launch_head(cur, b, L, d, m->head_norm_w, m->head_norm_b, m->head_w, m->head_b, m->out_dims, out, s, m->l_d_valid);
  { int rc_ = launch_copy(out, cur, (long)((xbytes) / sizeof(float)), s); if (rc_ != HN_OK) return rc_; }
  return HN_OK;
}

static size_t impl_fusion_tape_bytes(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self_on_missing) {
  FusionPlan fp;
  if (plan_fusion(m, in, b, nullptr, 0, &fp) != HN_OK) return 0;
  static thread_local TapePlan tp;
  if (plan_tape(m, in, b, masked, skip_self_on_missing, fp, &tp) != HN_OK) return 0;
  return align_up(tp.floats * sizeof(float), 256);
}

}  // extern "C"

// Context layout of the training forward / backward: the ones column and the packed channel order of the inference forward.
// Round 4: under dropout on the probabilities too -- the ones column's accumulator is the thinned row sum the rank-D binding
// needs there anyway (the full denominator is summed on the VALU), and the packed order saves one QK^T k-step in the forward
// core and two of twelve in the dQ kernel.  (Dropping blocks need the bounded core for that: a LayerNorm-ed context.)
static void train_context_layout(const hn_model *m, const FusionPlan &fp, bool *ones, int *pack) {
  for (int i = 0; i < m->n_modalities; ++i) {
    bool dropping = false, affine = true;
    for (int layer = 0; layer < m->depth; ++layer) {
      const hn_attn_params &ap = m->cross_attn[layer * m->n_modalities + i];
      dropping = dropping || ap.dropout > 0.0f;
      affine = affine && ap.ctx_gamma != nullptr;
    }
    ones[i] = fp.z[i] != nullptr && fp.ones[i] && affine && (!dropping || !drop_bound_disabled());
    // (the backward's dq kernel has no variant whose last 16-column block contributes zero k-steps: D = 16 / 17 on a 32-column
    // row packs into exactly 4 steps -- that shape trains on the natural layout)
    pack[i] = (ones[i] && fp.pack[i] % 4 != 0) ? fp.pack[i] : 0;
  }
}

extern "C" {

static int impl_fusion_tape_layout(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self_on_missing,
                          size_t *stats_off, size_t *x_off) {
  HN_REQUIRE(stats_off && x_off, HN_E_NULL, "fusion_tape_layout: NULL output");
  FusionPlan fp;
  int rc = plan_fusion(m, in, b, nullptr, 0, &fp);
  if (rc != HN_OK) return rc;
  static thread_local TapePlan tp;
  if ((rc = plan_tape(m, in, b, masked, skip_self_on_missing, fp, &tp)) != HN_OK) return rc;
  const int M = m->n_modalities;
  for (int i = 0; i < m->depth * (M + 1); ++i) stats_off[i] = x_off[i] = (size_t)-1;
  for (int k = 0; k < tp.nsteps; ++k) {        // a slot executed twice (the self-attention of a layer) reports its last run
    const Step &st = tp.steps[k];
    if (st.kind != STEP_CROSS_ATTN && st.kind != STEP_SELF_ATTN) continue;
    const int slot = st.layer * (M + 1) + (st.kind == STEP_CROSS_ATTN ? st.m : M);
    stats_off[slot] = tp.stats_off[k];
    x_off[slot] = tp.x_off[k];
  }
  return HN_OK;
}

static int impl_fusion_forward_train(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                            int return_embeddings, float *out, float **attn_stats, float **x_trace, void *tape,
                            size_t tape_bytes, void *workspace, size_t workspace_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  HN_REQUIRE(out && tape, HN_E_NULL, "fusion_forward_train: out / tape is NULL");
  FusionPlan fp;
  int rc = plan_fusion(m, in, b, nullptr, 0, &fp);
  if (rc != HN_OK) return rc;
  if ((rc = check_ws(workspace, workspace_bytes, fp.bytes, "fusion_forward_train")) != HN_OK) return rc;
  if ((rc = plan_fusion(m, in, b, workspace, workspace_bytes, &fp)) != HN_OK) return rc;
  static thread_local TapePlan tp;
  if ((rc = plan_tape(m, in, b, mask != nullptr, skip_self_on_missing, fp, &tp)) != HN_OK) return rc;
  HN_REQUIRE(tape_bytes >= tp.floats * sizeof(float) && ((uintptr_t)tape & 255) == 0, HN_E_WORKSPACE,
             "fusion_forward_train: tape %zu bytes < required %zu (256-byte aligned)", tape_bytes, tp.floats * sizeof(float));
  float *T = (float *)tape;
  const int M = m->n_modalities, L = m->l_c, d = m->l_d;
  bool tones[16]; int tpack[16];
  train_context_layout(m, fp, tones, tpack);
  for (int i = 0; i < M; ++i) {
    if (!in[i].data) continue;
    if (tp.z_off[i] != kNoSlot && fp.z[i]) fp.z[i] = T + tp.z_off[i];      // the context lives on the tape: the backward reads it back
    if ((rc = launch_encode(in[i].data, in[i].dtype, b, m->num_spatial_axes[i], in[i].spatial, m->channel_dims[i], m->num_freq_bands,
                            m->max_freq, m->fourier_encode_data, 1, 1e-5f, fp.z[i], fp.ldz[i], s, tones[i] ? fp.ldz[i] - 1 : -1,
                            tpack[i])) != HN_OK)
      return rc;
  }
  // (the launch also zeroes the cluster flags of the latent chains: small batches run them as clusters, chain.hip)
  if ((rc = launch_broadcast_rows(m->latents, T + tp.x_off[0], (long)L * d, b, s, fp.flags + m->depth * M, CHAIN_XCHG_FLAGS)) != HN_OK) return rc;
  int chain_seq = 0;
  const bool use_chain = fp.chain && !chain_disabled();
  auto is_attn_t = [](const Step &q) { return q.kind == STEP_CROSS_ATTN || q.kind == STEP_SELF_ATTN; };
  auto trace_copies = [&](int k) -> int {       // optional copies for hn_attn_probs (same slots as hn_fusion_forward)
    const Step &st = tp.steps[k];
    const int slot = st.layer * (M + 1) + (st.kind == STEP_CROSS_ATTN ? st.m : M);
    const int heads = st.kind == STEP_CROSS_ATTN ? m->cross_attn[st.layer * M + st.m].heads : m->self_attn[st.layer].heads;
    if (attn_stats && attn_stats[slot])
      { int rc_ = launch_copy(attn_stats[slot], T + tp.stats_off[k], (long)((size_t)b * heads * L * 2), s); if (rc_ != HN_OK) return rc_; }
    if (x_trace && x_trace[slot])
      { int rc_ = launch_copy(x_trace[slot], T + tp.x_off[k], (long)((size_t)b * L * d), s); if (rc_ != HN_OK) return rc_; }
    return HN_OK;
  };
  bool q_done = false, kv_done = false;          // projections of the attention block at `k` already produced by the chain in front of it
  // the projections of the attention block at step k live in its tape slots: what the chain in front wrote there is found there
  // (q / kv), what the block projects itself goes there (q_home / kv_home)
  auto tape_homes = [&](int k, AttnExt *e) -> bool {
    bool any = false;
    if (tp.q_off[k] != kNoSlot) { e->q = e->q_home = T + tp.q_off[k]; any = true; }
    if (tp.kv_off[k] != kNoSlot) { e->kv = e->kv_home = T + tp.kv_off[k]; any = true; }
    return any;
  };
  const bool staged = m->l_d_valid > 0;          // staged model: every LayerNorm of the latent side runs inside a chain (valid width)
  // projections of the attention block at step kn (if it is one that needs them) as the last stages of chain `ca`, straight into
  // that block's tape slots
  auto add_next_proj = [&](ChainArgs &ca, int kn) -> int {
    q_done = kv_done = false;
    if (!(kn < tp.nsteps && is_attn_t(tp.steps[kn]))) return HN_OK;
    const Step &sn = tp.steps[kn];
    const bool nself = sn.kind == STEP_SELF_ATTN;
    const hn_attn_params *an = nself ? &m->self_attn[sn.layer] : &m->cross_attn[sn.layer * M + sn.m];
    AttnPlan pn;
    int rc2 = plan_attn(an, !nself, nself ? 0 : fp.ldz[sn.m], b, L, nself ? L : fp.N[sn.m], nself ? d : fp.D[sn.m], nullptr, 0, &pn);
    if (rc2 != HN_OK) return rc2;
    const bool one_token = !nself && fp.N[sn.m] == 1 && mask == nullptr && !(an->dropout > 0.0f);
    if (!one_token && pn.dh == pn.dhp && (pn.inner % 128 == 0 || staged_attn(an)) && pn.inner % 16 == 0 && pn.inner <= 512 &&
        an->query_dim == d && an->w_q && an->w_kv && chain_proj_aligned(an)) {
      ca.p_nw = an->norm_w; ca.p_nb = an->norm_b;
      ca.nq = up128(pn.inner); ca.q_cols = pn.inner; ca.wq = an->w_q; ca.ldq = pn.inner;
      ca.Q = tp.q_off[kn] != kNoSlot ? T + tp.q_off[kn] : fp.cq;      // straight into the next block's tape slot
      ca.alpha_q = pn.rank_d ? 1.0f : pn.cscale;
      ca.xhat_out = (tp.xhat_off[kn] != kNoSlot && an->norm_w) ? T + tp.xhat_off[kn] : nullptr;
      q_done = true;
      if (nself) {
        ca.nkv = up128(2 * pn.inner); ca.kv_cols = 2 * pn.inner; ca.wkv = an->w_kv; ca.ldkv = 2 * pn.inner; kv_done = true;
        ca.KV = tp.kv_off[kn] != kNoSlot ? T + tp.kv_off[kn] : fp.ckv;
      }
    }
    return HN_OK;
  };
  auto launch_chain = [&](ChainArgs &ca) -> int {
    ca.rows = b * L; ca.L = L; ca.dv = m->l_d_valid;
    ca.xchg = fp.xchg; ca.xflags = fp.flags + m->depth * M; ca.seq = ++chain_seq;
    return launch_latent_chain(ca, s);
  };
  if (staged && use_chain && tp.nsteps > 0 && is_attn_t(tp.steps[0])) {      // the first block's projections: a chain of their own
    ChainArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.x_in = T + tp.x_off[0];
    if ((rc = add_next_proj(ca, 0)) != HN_OK) return rc;
    if (q_done && (rc = launch_chain(ca)) != HN_OK) return rc;
  }
  for (int k = 0; k < tp.nsteps;) {
    const Step &st = tp.steps[k];
    const float *xin = T + tp.x_off[k];
    float *xout = T + tp.x_off[k + 1];
    // dropout: one generator state per forward (hn_model.rng), one stream id per executed block (its step index)
    const hn_rng rng = {m->rng.seed, m->rng.offset, (uint32_t)k, m->rng.offset_dev};
    if (staged && use_chain && !is_attn_t(st)) {       // a feed-forward block not absorbed by the chain of an attention block: a chain without a head
      const hn_ff_params &fq = st.kind == STEP_CROSS_FF ? m->cross_ff[st.layer * M + st.m] : m->self_ff[st.layer];
      if (chain_ff_aligned(&fq) && al16(xin) && al16(xout) && fq.w1 && fq.b1 && fq.w2 && fq.b2) {
        ChainArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.x_in = xin; ca.x_out = xout; ca.head = 0;
        ca.has_ff = 1; ca.gate = fq.gate; ca.f_nw = fq.norm_w; ca.f_nb = fq.norm_b;
        ca.w1 = fq.w1; ca.b1 = fq.b1; ca.w2 = fq.w2; ca.b2 = fq.b2;
        ca.ff_drop = drop_of(fq.dropout, rng, true);
        if ((rc = add_next_proj(ca, k + 1)) != HN_OK) return rc;
        if ((rc = launch_chain(ca)) != HN_OK) return rc;
        ++k;
        continue;
      }
    }
    // The latent chain behind an attention block, as in hn_fusion_forward (out-projection + residual, the feed-forward block,
    // the projections of the attention block after it), with the feed-forward block's input kept on the tape (x_mid): the
    // backward recomputes everything else of these blocks from the tape as before.  The feed-forward block's dropout is applied
    // inside the chain (same generator, same stream id as the per-block route).  Not behind the one-token shortcut.
    bool fuse = false, fuse_tab = false;
    if (use_chain && is_attn_t(st) && k + 1 < tp.nsteps && !is_attn_t(tp.steps[k + 1])) {
      const Step &sf = tp.steps[k + 1];
      const hn_ff_params &fq = sf.kind == STEP_CROSS_FF ? m->cross_ff[sf.layer * M + sf.m] : m->self_ff[sf.layer];
      const hn_attn_params &aq = st.kind == STEP_SELF_ATTN ? m->self_attn[st.layer] : m->cross_attn[st.layer * M + st.m];
      const int inner = aq.heads * aq.dim_head;
      fuse = fq.dim == d && fq.dropout >= 0.0f && fq.dropout < 1.0f && aq.query_dim == d && (inner % 128 == 0 || staged_attn(&aq)) &&
             inner % 16 == 0 && inner <= 512 &&
             chain_ff_aligned(&fq) && chain_out_aligned(&aq) && al16(xin) && al16(xout) && al16(T + tp.x_off[k + 2]) &&
             !(st.kind == STEP_CROSS_ATTN && fp.N[st.m] == 1 && mask == nullptr && !(aq.dropout > 0.0f));
      // the one-token shortcut (tabular / omic modality): its two skinny products run as before, the broadcast add of the block's
      // row, the feed-forward block and the next projections ride on ONE chain (head == 2) as in the inference forward
      // (round 4: add_row + FF1 + FF2 + LayerNorm + projection = 40 us of launches per layer at cfg4 b = 8 became a 22 us chain)
      static const bool no_tab_chain = tuning_env("HN_NO_TAB_CHAIN") != nullptr;
      fuse_tab = !fuse && !no_tab_chain && !staged && st.kind == STEP_CROSS_ATTN && fp.N[st.m] == 1 && mask == nullptr && !(aq.dropout > 0.0f) &&
                 fq.dim == d && fq.dropout >= 0.0f && fq.dropout < 1.0f && aq.query_dim == d && chain_ff_aligned(&fq) && al16(xin) &&
                 al16(xout) && al16(T + tp.x_off[k + 2]) && aq.w_out && aq.b_out;
      fuse = fuse || fuse_tab;
    }
    // LN(x) of the block's input for the backward's dW_q / dW_kv: the chain that projected for the block wrote it (q_done), else here
    if (is_attn_t(st) && tp.xhat_off[k] != kNoSlot && !q_done) {
      const hn_attn_params &aq = st.kind == STEP_SELF_ATTN ? m->self_attn[st.layer] : m->cross_attn[st.layer * M + st.m];
      if ((rc = launch_ln_fwd(xin, aq.norm_w, aq.norm_b, b * L, d, T + tp.xhat_off[k], s, aq.query_dim_valid)) != HN_OK) return rc;
    }
    if (fuse) {
      const bool self = st.kind == STEP_SELF_ATTN;
      hn_attn_params ap = self ? m->self_attn[st.layer] : m->cross_attn[st.layer * M + st.m];
      ap.rng = rng;
      AttnExt ext = {fp.cq, fp.ckv, q_done, kv_done, true, nullptr, 0, false, false, nullptr, nullptr, nullptr, 0, 0, 0};
      tape_homes(k, &ext);
      if (self)
        rc = attn_fwd_impl(&ap, xin, nullptr, 1, nullptr, 0, b, L, L, d, nullptr, T + tp.stats_off[k], fp.op_ws, fp.op_ws_bytes, s,
                           nullptr, nullptr, T + tp.saved_off[k], false, 0, nullptr, nullptr, &ext);
      else
        rc = attn_fwd_impl(&ap, xin, nullptr, 1, fp.z[st.m], fp.ldz[st.m], b, L, fp.N[st.m], fp.D[st.m], mask, T + tp.stats_off[k],
                           fp.op_ws, fp.op_ws_bytes, s, nullptr, nullptr, T + tp.saved_off[k], tones[st.m], tpack[st.m], nullptr, nullptr,
                           &ext);
      if (rc != HN_OK) return rc;
      HN_REQUIRE(fuse_tab ? ext.y_out != nullptr : ext.o_out != nullptr, HN_E_UNSUPPORTED,
                 "fusion_forward_train: attention block did not defer its out-projection");
      const Step &sf = tp.steps[k + 1];
      const hn_ff_params &fq = sf.kind == STEP_CROSS_FF ? m->cross_ff[sf.layer * M + sf.m] : m->self_ff[sf.layer];
      HN_REQUIRE(fq.w1 && fq.b1 && fq.w2 && fq.b2, HN_E_NULL, "ff: weight pointer is NULL");
      ChainArgs ca;
      memset(&ca, 0, sizeof(ca));
      ca.x_in = xin;
      if (fuse_tab) { ca.head = 2; ca.y = ext.y_out; }
      else {
        ca.head = 1; ca.O = ext.o_out; ca.ldo = ext.ldo_out; ca.inner_o = up128(ap.heads * ap.dim_head); ca.o_cols = ap.heads * ap.dim_head;
        ca.w_out = ap.w_out; ca.b_out = ap.b_out;
      }
      ca.has_ff = 1; ca.gate = fq.gate; ca.f_nw = fq.norm_w; ca.f_nb = fq.norm_b;
      ca.w1 = fq.w1; ca.b1 = fq.b1; ca.w2 = fq.w2; ca.b2 = fq.b2;
      {
        const hn_rng rng_ff = {m->rng.seed, m->rng.offset, (uint32_t)(k + 1), m->rng.offset_dev};     // the feed-forward block's stream id: its step index
        ca.ff_drop = drop_of(fq.dropout, rng_ff, true);
      }
      ca.x_mid = xout;
      ca.x_out = T + tp.x_off[k + 2];
      if ((rc = add_next_proj(ca, k + 2)) != HN_OK) return rc;
      if ((rc = launch_chain(ca)) != HN_OK) return rc;
      if ((rc = trace_copies(k)) != HN_OK) return rc;
      k += 2;
      continue;
    }
    AttnExt ext = {fp.cq, fp.ckv, q_done, kv_done, false, nullptr, 0, false, false, nullptr, nullptr, nullptr, 0, 0, 0};
    AttnExt *extp = (q_done || kv_done) ? &ext : nullptr;
    if (is_attn_t(st) && tape_homes(k, &ext)) extp = &ext;
    switch (st.kind) {
      case STEP_CROSS_ATTN: {
        hn_attn_params ap = m->cross_attn[st.layer * M + st.m];
        ap.rng = rng;
        rc = attn_fwd_impl(&ap, xin, xout, 1, fp.z[st.m], fp.ldz[st.m], b, L, fp.N[st.m], fp.D[st.m],
                           mask, T + tp.stats_off[k], fp.op_ws, fp.op_ws_bytes, s, nullptr, nullptr, T + tp.saved_off[k],
                           tones[st.m], tpack[st.m], nullptr, nullptr, extp);
        break;
      }
      case STEP_SELF_ATTN: {
        hn_attn_params ap = m->self_attn[st.layer];
        ap.rng = rng;
        rc = attn_fwd_impl(&ap, xin, xout, 1, nullptr, 0, b, L, L, d, nullptr, T + tp.stats_off[k], fp.op_ws,
                           fp.op_ws_bytes, s, nullptr, nullptr, T + tp.saved_off[k], false, 0, nullptr, nullptr, extp);
        break;
      }
      default: {
        hn_ff_params fpar = st.kind == STEP_CROSS_FF ? m->cross_ff[st.layer * M + st.m] : m->self_ff[st.layer];
        fpar.rng = rng;
        rc = ff_fwd_impl(&fpar, xin, xout, 1, b * L, fp.op_ws, fp.op_ws_bytes, s, true);
        break;
      }
    }
    if (rc != HN_OK) return rc;
    if (is_attn_t(st)) {
      if ((rc = trace_copies(k)) != HN_OK) return rc;
      q_done = kv_done = false;
    }
    ++k;
  }
  const float *xf = T + tp.x_off[tp.nsteps];
  if (m->final_classifier_head && !return_embeddings)
    return launch_head(xf, b, L, d, m->head_norm_w, m->head_norm_b, m->head_w, m->head_b, m->out_dims, out, s, m->l_d_valid);
  { int rc_ = launch_copy(out, xf, (long)((size_t)b * L * d), s); if (rc_ != HN_OK) return rc_; }
  return HN_OK;
}

static size_t impl_fusion_backward_workspace_bytes(const hn_model *m, const hn_modality_input *in, int b, int masked) {
  FusionPlan fp;
  float *dX, *hs;
  void *op;
  size_t opb, total;
  if (fusion_bwd_workspace(m, in, b, masked, nullptr, 0, &fp, &dX, &hs, &op, &opb, &total) != HN_OK) return 0;
  return total;
}

static int impl_fusion_backward(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                       int return_embeddings, const float *dout, const void *tape, const hn_model_grads *g, void *workspace,
                       size_t workspace_bytes, void *stream, const hn_grad_ready *ready) {
  hipStream_t s = (hipStream_t)stream;
  HN_REQUIRE(dout && tape && g, HN_E_NULL, "fusion_backward: NULL pointer");
  FusionPlan fp;
  float *dX, *hs;
  void *op;
  size_t opb, total;
  int rc = fusion_bwd_workspace(m, in, b, mask != nullptr, nullptr, 0, &fp, &dX, &hs, &op, &opb, &total);
  if (rc != HN_OK) return rc;
  if ((rc = check_ws(workspace, workspace_bytes, total, "fusion_backward")) != HN_OK) return rc;
  float *tbuf = nullptr;
  size_t tfloats = 0;
  BChainBufs cb;
  if ((rc = fusion_bwd_workspace(m, in, b, mask != nullptr, workspace, workspace_bytes, &fp, &dX, &hs, &op, &opb, &total, &tbuf, &tfloats, &cb)) != HN_OK) return rc;
  static thread_local TapePlan tp;
  if ((rc = plan_tape(m, in, b, mask != nullptr, skip_self_on_missing, fp, &tp)) != HN_OK) return rc;
  // every weight the dX products read transposed, in ONE batched launch per 16 instead of a launch in front of each product
  register_transposes(m, in, b, mask != nullptr, tp.steps, tp.nsteps);
  struct CacheGuard { ~CacheGuard() { transpose_cache_end(); } } cache_guard;      // the cache lives for this call only
  if ((rc = transpose_cache_run(tbuf, tfloats, s)) != HN_OK) return rc;
  const float *T = (const float *)tape;
  const int M = m->n_modalities, L = m->l_c, d = m->l_d;
  const size_t xn = (size_t)b * L * d;
  bool tones[16]; int tpack[16];
  train_context_layout(m, fp, tones, tpack);
  for (int i = 0; i < M; ++i) {     // the normalised contexts: from the tape, or recomputed (one HBM pass) when they were not kept
    if (!in[i].data) continue;
    if (tp.z_off[i] != kNoSlot && fp.z[i]) { fp.z[i] = const_cast<float *>(T) + tp.z_off[i]; continue; }
    if ((rc = launch_encode(in[i].data, in[i].dtype, b, m->num_spatial_axes[i], in[i].spatial, m->channel_dims[i], m->num_freq_bands,
                            m->max_freq, m->fourier_encode_data, 1, 1e-5f, fp.z[i], fp.ldz[i], s, tones[i] ? fp.ldz[i] - 1 : -1,
                            tpack[i])) != HN_OK)
      return rc;
  }
  const float *xf = T + tp.x_off[tp.nsteps];
  if (m->final_classifier_head && !return_embeddings) {
    if ((rc = launch_head_bwd(xf, b, L, d, m->head_norm_w, m->head_norm_b, m->head_w, m->out_dims, dout, dX, g->head_norm_w,
                              g->head_norm_b, g->head_w, g->head_b, hs, s, m->l_d_valid)) != HN_OK) return rc;
  } else {
    { int rc_ = launch_copy(dX, dout, (long)(xn), s); if (rc_ != HN_OK) return rc_; }
  }
  // grad_ready[depth]: the head's parameter gradients are final; grad_ready[l]: every block of layers >= l has run its
  // backward, so every gradient range only those layers accumulate into is final (layers finish in reverse order)
  auto signal = [&](int idx) -> int {
    if (!ready) return HN_OK;
    if (ready->events && ready->events[idx]) HN_HIP_CHECK(hipEventRecord((hipEvent_t)ready->events[idx], s));
    if (ready->notify) ready->notify(idx, ready->user);
    return HN_OK;
  };
  if ((rc = signal(m->depth)) != HN_OK) return rc;
  int next_layer_event = m->depth - 1;     // highest layer whose event has not been recorded yet
  static const hn_attn_grads no_attn = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  static const hn_ff_grads no_ff = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // ---- the fused latent backward (bchain.hip): a feed-forward block's backward runs in ONE launch together with the projection
  // backward of the attention block behind it (whose core backward has just run: `pend`) and the out-projection backward of the
  // attention block in front of it; the weight gradients of the chain follow in one batched launch + one reduce.
  const int rows = b * L;
  const bool use_bchain = cb.ok && fp.chain && !bchain_disabled() && !chain_disabled();
  int bchain_seq = 0;
  if (use_bchain && (rc = launch_fill((float *)cb.xflags, 0.0f, BCHAIN_XFLAGS, s)) != HN_OK) return rc;      // cluster flags (bchain.hip)
  // `durable`: dQ / dKV sit in a buffer set and LN(x) on the tape -- the products may wait for the batch's flush
  struct Pending { bool valid, durable; int layer; hn_attn_params ap; const hn_attn_grads *ag; const float *x_in, *dQ, *dKV, *xhat; } pend;
  memset(&pend, 0, sizeof(pend));
  int cur_set = 0;                     // buffer set of the chain that ran last (its dPre / dO feed the attention backward behind it)
  auto is_attn_b = [](const Step &q) { return q.kind == STEP_CROSS_ATTN || q.kind == STEP_SELF_ATTN; };
  auto attn_of = [&](const Step &q) { return q.kind == STEP_SELF_ATTN ? m->self_attn[q.layer] : m->cross_attn[q.layer * M + q.m]; };
  auto attn_grads_of = [&](const Step &q) -> const hn_attn_grads * {
    if (q.kind == STEP_SELF_ATTN) return g->self_attn ? &g->self_attn[q.layer] : &no_attn;
    return g->cross_attn ? &g->cross_attn[q.layer * M + q.m] : &no_attn;
  };
  auto ff_of_b = [&](const Step &q) { return q.kind == STEP_CROSS_FF ? m->cross_ff[q.layer * M + q.m] : m->self_ff[q.layer]; };
  auto ff_grads_of = [&](const Step &q) -> const hn_ff_grads * {
    if (q.kind == STEP_CROSS_FF) return g->cross_ff ? &g->cross_ff[q.layer * M + q.m] : &no_ff;
    return g->self_ff ? &g->self_ff[q.layer] : &no_ff;
  };
  auto one_token = [&](const Step &q, const hn_attn_params &ap) {
    return q.kind == STEP_CROSS_ATTN && fp.N[q.m] == 1 && mask == nullptr && !(ap.dropout > 0.0f);
  };
  // an attention block whose row-local backward (out-projection in front of the core, projections behind it) can ride on chains
  auto attn_chainable = [&](const Step &q, const hn_attn_params &ap) {
    const int inner = ap.heads * ap.dim_head;
    const bool sg = staged_attn(&ap);
    return use_bchain && is_attn_b(q) && !one_token(q, ap) && ap.query_dim == d && (inner % 128 == 0 || sg) && inner % 16 == 0 && inner <= 512 &&
           ap.w_q && ap.w_kv && ap.w_out && transpose_cache_lookup(ap.w_out, wo_ld(&ap), d, wo_ld(&ap)) &&
           transpose_cache_lookup(ap.w_q, d, sg ? up128(inner) : inner, d) &&
           (q.kind != STEP_SELF_ATTN || transpose_cache_lookup(ap.w_kv, d, sg ? up128(2 * inner) : 2 * inner, d)) && al16(ap.norm_w);
  };
  auto ff_chainable = [&](const hn_ff_params &f, const float *x) {
    return use_bchain && f.dim == d && f.dropout >= 0.0f && f.dropout < 1.0f && f.w1 && f.b1 && f.w2 && f.b2 && al16(f.w1) && al16(f.b1) && al16(f.norm_w) &&
           al16(f.norm_b) && (f.norm_w != nullptr || f.norm_b == nullptr) && al16(x) && transpose_cache_lookup(f.w2, 4 * d, d, 4 * d) &&
           transpose_cache_lookup(f.w1, d, 8 * d, d);
  };
  // ---- the weight-gradient products of the chains wait in `big` and run as ONE batched launch + ONE reduce per flush: at the end of
  // a layer (in front of its gradient-ready signal), when the buffer sets or the batch's capacity run out, at the end of the pass.
  // Chain number `chain_no` writes into buffer set chain_no % BCHAIN_SETS; the attention backward behind it leaves dQ / dKV for the
  // NEXT chain in set (chain_no + 1) % BCHAIN_SETS.
  static thread_local GemmTnMulti big;
  big.n = big.n_ln = 0;
  big.K = rows;
  int chain_no = 0, pending_chains = 0;
  bool flush_now = false;              // an operand of the batch lives in the op workspace (LN(x) not on the tape): no deferral
  auto flush_products = [&]() -> int {
    pending_chains = 0;
    flush_now = false;
    if (big.n == 0 && big.n_ln == 0) return HN_OK;
    int rc2 = launch_gemm_tn_multi(big, cb.tn, cb.tn_floats, s);
    big.n = big.n_ln = 0;
    return rc2;
  };
  // one chain launch; its weight-gradient products join the batch.  ff_k < 0: projection backward of `pend` only.
  auto run_bchain = [&](int ff_k, bool has_out, const AttnBwdExt *out_ext_in, const float **o_saved) -> int {
    (void)out_ext_in;
    int rc2 = HN_OK;
    // room for this chain's products (<= 5) and LayerNorm entries (<= 4), and a free buffer set for the attention backward behind it
    if (big.n + 5 > TN_MULTI_MAX || big.n_ln + 4 > TN_MULTI_LN_MAX || pending_chains >= BCHAIN_SETS - 2) {
      if ((rc2 = flush_products()) != HN_OK) return rc2;
    }
    const BChainSet &bs = cb.set[chain_no % BCHAIN_SETS];
    BChainArgs ca;
    memset(&ca, 0, sizeof(ca));
    GemmTnMulti &mm = big;
    const int n_before = mm.n, nln_before = mm.n_ln;
    auto add_product = [&](const float *A, long lda, int Mm, const float *B, long ldb, int Nn, float *C, long ldc, float *cs) {
      if (!C) {
        if (cs && rc2 == HN_OK) rc2 = launch_colsum(A, lda, rows, Mm, 1.0f, cs, 1, s);
        return;
      }
      TnProduct &pr = mm.p[mm.n++];
      pr.A = A; pr.lda = lda; pr.B = B; pr.ldb = ldb; pr.C = C; pr.ldc = ldc; pr.M = Mm; pr.N = Nn; pr.colsum = cs;
    };
    auto add_ln = [&](int slot, float *out) {
      if (!out) return;
      LnPartial &lp = mm.ln[mm.n_ln++];
      lp.part = bs.lnpart + (size_t)slot * 128; lp.nwg = (rows + 15) / 16; lp.width = 128; lp.stride = 4 * 128; lp.out = out;
    };
    ca.rows = rows; ca.L = L; ca.dy = dX; ca.dx_out = dX; ca.lnpart = bs.lnpart; ca.dv = m->l_d_valid;
    ca.xchg = cb.xchg; ca.xflags = cb.xflags; ca.seq = ++bchain_seq;
    if (pend.valid) {
      const int inner = pend.ap.heads * pend.ap.dim_head;
      const bool sg = staged_attn(&pend.ap);
      ca.has_p = 1; ca.dQ = pend.dQ; ca.lddq = inner; ca.nq = up128(inner); ca.q_cols = inner;
      ca.wqT = transpose_cache_lookup(pend.ap.w_q, d, sg ? up128(inner) : inner, d);
      if (pend.dKV) {
        ca.dKV = pend.dKV; ca.lddkv = 2 * inner; ca.nkv = up128(2 * inner); ca.kv_cols = 2 * inner;
        ca.wkvT = transpose_cache_lookup(pend.ap.w_kv, d, sg ? up128(2 * inner) : 2 * inner, d);
      }
      ca.p_x = pend.x_in; ca.p_nw = pend.ap.norm_w;
      add_product(pend.dQ, inner, inner, pend.xhat, d, d, pend.ag->w_q, d, nullptr);
      if (pend.dKV) add_product(pend.dKV, 2 * inner, 2 * inner, pend.xhat, d, d, pend.ag->w_kv, d, nullptr);
      if (pend.ap.norm_w) { add_ln(0, pend.ag->norm_w); add_ln(1, pend.ag->norm_b); }
      if (!pend.durable) flush_now = true;
    }
    if (ff_k >= 0) {
      const Step &sf = tp.steps[ff_k];
      const hn_ff_params f = ff_of_b(sf);
      const hn_ff_grads *fg = ff_grads_of(sf);
      ca.has_ff = 1; ca.gate = f.gate; ca.f_x = T + tp.x_off[ff_k];
      ca.f_nw = f.norm_w; ca.f_nb = f.norm_b; ca.w1 = f.w1; ca.b1 = f.b1;
      ca.w2T = transpose_cache_lookup(f.w2, 4 * d, d, 4 * d); ca.w1T = transpose_cache_lookup(f.w1, d, 8 * d, d);
      ca.H = bs.H; ca.dU = bs.dU; ca.Xhat = bs.Xhat; ca.dYff = bs.dYff;
      {
        const hn_rng rng_ff = {m->rng.seed, m->rng.offset, (uint32_t)ff_k, m->rng.offset_dev};      // the forward's generator state and stream id
        ca.ff_drop = drop_of(f.dropout, rng_ff, true);
      }
      add_product(bs.dU, 8 * d, 8 * d, bs.Xhat, d, d, fg->w1, d, fg->b1);
      add_product(bs.dYff, d, d, bs.H, 4 * d, 4 * d, fg->w2, 4 * d, fg->b2);
      if (f.norm_w) { add_ln(2, fg->norm_w); add_ln(3, fg->norm_b); }
      if (has_out) {
        const Step &sa = tp.steps[ff_k - 1];
        const hn_attn_params oa = attn_of(sa);
        const int inner = oa.heads * oa.dim_head;
        ca.has_out = 1; ca.inner_o = up128(inner); ca.o_cols = inner; ca.o_x = T + tp.x_off[ff_k - 1];
        ca.woT = transpose_cache_lookup(oa.w_out, wo_ld(&oa), d, wo_ld(&oa));
        ca.dPre = bs.dPre; ca.dO = cb.dO; ca.lddo = inner;
        if (o_saved && *o_saved) add_product(bs.dPre, d, d, *o_saved, inner, inner, attn_grads_of(sa)->w_out, wo_ld(&oa), attn_grads_of(sa)->b_out);
      }
    }
    if ((rc2 = (rc2 != HN_OK ? rc2 : launch_latent_bchain(ca, s))) != HN_OK) return rc2;
    pend.valid = false;
    cur_set = chain_no % BCHAIN_SETS;
    ++chain_no;
    if (mm.n > n_before || mm.n_ln > nln_before) ++pending_chains;
    static const bool no_batch = getenv("HN_NO_TN_BATCH") != nullptr;      // route switch (A/B): a launch pair per chain, as until round 5
    if (flush_now || no_batch) return flush_products();
    // (scratch: a batch never needs more than one chain's worst case -- the planner's split count shrinks as tiles are added, the
    // partials stay below max(32 x one chain's outputs, 768 tiles) -- and the buffer holds twice that; launch_gemm_tn_multi checks)
    return HN_OK;
  };
  // backward of an attention block with the chain hooks; `dpre` / `dO` non-NULL: its out-projection already ran in a chain
  auto run_attn = [&](int k, const float *dpre, const float *dO_in, bool skip_wout) -> int {
    const Step &st = tp.steps[k];
    hn_attn_params ap = attn_of(st);
    const hn_rng rng = {m->rng.seed, m->rng.offset, (uint32_t)k, m->rng.offset_dev};      // the forward's generator state and stream id
    ap.rng = rng;
    const bool defer = attn_chainable(st, ap) && al16(T + tp.x_off[k]);
    AttnBwdExt ext;
    memset(&ext, 0, sizeof(ext));
    ext.dpre = dpre; ext.dO = dO_in; ext.skip_wout = skip_wout; ext.defer_proj = defer;
    if (tp.q_off[k] != kNoSlot) ext.q_taped = T + tp.q_off[k];
    if (tp.kv_off[k] != kNoSlot) ext.kv_taped = T + tp.kv_off[k];
    if (tp.xhat_off[k] != kNoSlot) ext.xhat_taped = T + tp.xhat_off[k];
    AttnBwdExt *extp = (dpre || defer || ext.q_taped || ext.xhat_taped) ? &ext : nullptr;
    if (defer) {      // dQ / dKV for the NEXT chain's products: into that chain's buffer set
      const BChainSet &ns = cb.set[chain_no % BCHAIN_SETS];
      ext.dQ_home = ns.dQ; ext.dKV_home = ns.dKV;
    }
    const float *xin = T + tp.x_off[k], *xout = T + tp.x_off[k + 1];
    int rc2;
    if (st.kind == STEP_CROSS_ATTN)
      rc2 = attn_bwd_impl(&ap, xin, xout, 1, fp.z[st.m], fp.ldz[st.m], b, L, fp.N[st.m], fp.D[st.m], mask, T + tp.stats_off[k],
                          T + tp.saved_off[k], dX, dX, attn_grads_of(st), op, opb, s, tpack[st.m], extp);
    else
      rc2 = attn_bwd_impl(&ap, xin, xout, 1, nullptr, 0, b, L, L, d, nullptr, T + tp.stats_off[k], T + tp.saved_off[k], dX, dX,
                          attn_grads_of(st), op, opb, s, 0, extp);
    if (rc2 != HN_OK) return rc2;
    if (defer) {
      pend.valid = true; pend.layer = st.layer; pend.ap = ap; pend.ag = attn_grads_of(st); pend.x_in = xin;
      pend.dQ = ext.dQ; pend.dKV = ext.dKV; pend.xhat = ext.xhat;
      pend.durable = ext.xhat_taped != nullptr || !ap.norm_w;      // (no LayerNorm: xhat is the block's input on the tape)
    }
    return HN_OK;
  };
  for (int k = tp.nsteps - 1; k >= 0;) {
    const Step &st = tp.steps[k];
    const float *xin = T + tp.x_off[k];
    const hn_rng rng = {m->rng.seed, m->rng.offset, (uint32_t)k, m->rng.offset_dev};      // the forward's generator state and stream id
    if (!is_attn_b(st) && ff_chainable(ff_of_b(st), xin)) {
      bool has_out = false, o_on_tape = false;
      const float *o_saved = nullptr;
      if (k >= 1 && is_attn_b(tp.steps[k - 1])) {
        const Step &sa = tp.steps[k - 1];
        const hn_attn_params oa = attn_of(sa);
        has_out = attn_chainable(sa, oa) && al16(T + tp.x_off[k - 1]);
        if (has_out) {
          // explicit K/V bindings keep O on the tape: dW_out rides on the chain's batched launch; the shared-context (rank-D)
          // binding recomputes O inside its backward and keeps dW_out there
          AttnPlan pa;
          const bool self = sa.kind == STEP_SELF_ATTN;
          if ((rc = plan_attn(&oa, !self, self ? 0 : fp.ldz[sa.m], b, L, self ? L : fp.N[sa.m], self ? d : fp.D[sa.m], nullptr, 0, &pa)) != HN_OK) return rc;
          o_on_tape = !pa.rank_d;
          if (o_on_tape) o_saved = T + tp.saved_off[k - 1];
        }
      }
      if ((rc = run_bchain(k, has_out, nullptr, &o_saved)) != HN_OK) return rc;
      if (has_out) {
        if ((rc = run_attn(k - 1, cb.set[cur_set].dPre, cb.dO, o_on_tape)) != HN_OK) return rc;
        k -= 2;
      } else {
        k -= 1;
      }
    } else {
      if (pend.valid && (rc = run_bchain(-1, false, nullptr, nullptr)) != HN_OK) return rc;
      if (is_attn_b(st)) {
        rc = run_attn(k, nullptr, nullptr, false);
      } else {
        hn_ff_params fpar = ff_of_b(st);
        fpar.rng = rng;
        rc = ff_bwd_impl(&fpar, xin, dX, dX, 1, b * L, ff_grads_of(st), op, opb, s);
      }
      if (rc != HN_OK) return rc;
      k -= 1;
    }
    {
      int done_above = k >= 0 ? tp.steps[k].layer : -1;   // layers > done_above have no block left ...
      if (pend.valid && pend.layer > done_above) done_above = pend.layer;      // ... and no projection backward pending in a chain
      if (next_layer_event > done_above) {
        // a layer is complete: its chains' weight-gradient products run now, in one batch (and in front of its gradient-ready signal)
        if ((rc = flush_products()) != HN_OK) return rc;
        for (; next_layer_event > done_above; --next_layer_event)
          if ((rc = signal(next_layer_event)) != HN_OK) return rc;
      }
    }
  }
  if (pend.valid && (rc = run_bchain(-1, false, nullptr, nullptr)) != HN_OK) return rc;
  if ((rc = flush_products()) != HN_OK) return rc;
  for (; next_layer_event >= 0; --next_layer_event)
    if ((rc = signal(next_layer_event)) != HN_OK) return rc;
  if (g->latents) return launch_colsum(dX, (long)L * d, b, L * d, 1.0f, g->latents, 1, s);   // x0 = latents broadcast over the batch
  return HN_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// latent block = latent self-attention + feed-forward (healnet.py:241-245), SURVEY.md 8(b) hn_latent_block_fwd / _bwd
// ------------------------------------------------------------------------------------------------
struct LatentBlockPlan { float *q, *kv, *xmid; void *op; size_t op_bytes, bytes; bool chain; };

static int plan_latent_block(const hn_attn_params *ap, const hn_ff_params *fp, int b, int L, void *ws, size_t ws_bytes, LatentBlockPlan *lp) {
  HN_REQUIRE(ap && fp, HN_E_NULL, "latent_block: params NULL");
  HN_REQUIRE(fp->dim == ap->query_dim, HN_E_SHAPE, "latent_block: attention width %d != feed-forward width %d", ap->query_dim, fp->dim);
  AttnPlan pl;
  int rc = plan_attn(ap, false, 0, b, L, L, ap->query_dim, nullptr, 0, &pl);
  if (rc != HN_OK) return rc;
  Arena ar(ws, ws_bytes);
  const size_t rows = (size_t)b * L;
  lp->chain = latent_chain_supported((int)rows, ap->query_dim, 4 * fp->dim) && pl.dh == pl.dhp && pl.inner % 128 == 0 && pl.inner <= 512 &&
              fp->dropout == 0.0f && ap->dropout == 0.0f && !chain_disabled() && chain_ff_aligned(fp) && chain_out_aligned(ap) &&
              chain_proj_aligned(ap);
  lp->q = ar.take<float>(rows * pl.heads * pl.dhp);
  lp->kv = ar.take<float>(rows * 2 * pl.heads * pl.dhp);
  lp->xmid = ar.take<float>(rows * ap->query_dim);
  const size_t ffb = ff_ws_bytes(fp, (int)rows);
  lp->op_bytes = pl.bytes > ffb ? pl.bytes : ffb;
  lp->op = ar.take<char>(lp->op_bytes);
  lp->bytes = ar.off;
  if (ws != nullptr && ar.overflow) return fail(HN_E_WORKSPACE, "latent_block: workspace %zu bytes < required %zu", ws_bytes, ar.off);
  return HN_OK;
}

extern "C" {

size_t hn_latent_block_workspace_bytes(const hn_attn_params *attn, const hn_ff_params *ff, int b, int L) {
  LatentBlockPlan lp;
  if (plan_latent_block(attn, ff, b, L, nullptr, 0, &lp) != HN_OK) return 0;
  return lp.bytes;
}

int hn_latent_block_fwd(const hn_attn_params *attn, const hn_ff_params *ff, const float *x_in, float *x_out, int b, int L,
                        float *x_mid, float *stats, float *saved, void *workspace, size_t workspace_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  HN_REQUIRE(x_in && x_out, HN_E_NULL, "latent_block: x is NULL");
  LatentBlockPlan lp;
  int rc = plan_latent_block(attn, ff, b, L, nullptr, 0, &lp);
  if (rc != HN_OK) return rc;
  if ((rc = check_ws(workspace, workspace_bytes, lp.bytes, "latent_block")) != HN_OK) return rc;
  if ((rc = plan_latent_block(attn, ff, b, L, workspace, workspace_bytes, &lp)) != HN_OK) return rc;
  const int d = attn->query_dim;
  const bool training = saved != nullptr;        // the training form keeps x_mid / stats / saved for hn_latent_block_bwd
  HN_REQUIRE(!training || (x_mid && stats), HN_E_NULL, "latent_block: the training form needs x_mid and stats");
  const bool drops = training && (attn->dropout > 0.0f || ff->dropout > 0.0f);
  const bool aligned = al16(x_in) && al16(x_out) && al16(x_mid) && al16(saved);
  if (!lp.chain || drops || !aligned) {                      // unfused: the two blocks back to back (any shape; dropout)
    float *mid = x_mid ? x_mid : lp.xmid;
    if ((rc = attn_fwd_impl(attn, x_in, mid, 1, nullptr, 0, b, L, L, d, nullptr, stats, lp.op, lp.op_bytes, s, nullptr, nullptr, saved)) != HN_OK)
      return rc;
    return ff_fwd_impl(ff, mid, x_out, 1, b * L, lp.op, lp.op_bytes, s, training);
  }
  AttnPlan pl;
  if ((rc = plan_attn(attn, false, 0, b, L, L, d, nullptr, 0, &pl)) != HN_OK) return rc;
  HN_REQUIRE(attn->w_q && attn->w_kv && attn->w_out && attn->b_out, HN_E_NULL, "attn: weight pointer is NULL");
  ChainArgs c1;                                   // Q | KV = LN(x) W^T
  memset(&c1, 0, sizeof(c1));
  c1.rows = b * L; c1.L = L; c1.x_in = x_in;
  c1.p_nw = attn->norm_w; c1.p_nb = attn->norm_b;
  c1.nq = pl.inner; c1.wq = attn->w_q; c1.Q = lp.q; c1.ldq = pl.inner; c1.alpha_q = pl.cscale;
  c1.nkv = 2 * pl.inner; c1.wkv = attn->w_kv; c1.KV = lp.kv; c1.ldkv = 2 * pl.inner;
  if ((rc = launch_latent_chain(c1, s)) != HN_OK) return rc;
  AttnExt ext = {lp.q, lp.kv, true, true, true, nullptr, 0, false, false, nullptr, nullptr, nullptr, 0, 0, 0};
  // (training: O is produced in `saved`, the feed-forward block's input is written to x_mid by the second chain)
  if ((rc = attn_fwd_impl(attn, x_in, nullptr, 1, nullptr, 0, b, L, L, d, nullptr, stats, lp.op, lp.op_bytes, s, nullptr, nullptr, saved,
                          false, 0, nullptr, nullptr, &ext)) != HN_OK) return rc;
  ChainArgs c2;                                   // x_out = x1 + FF(LN x1), x1 = x + LeakyReLU(O W_out^T + b_out)
  memset(&c2, 0, sizeof(c2));
  c2.rows = b * L; c2.L = L; c2.x_in = x_in; c2.x_out = x_out; c2.x_mid = x_mid;
  c2.head = 1; c2.O = ext.o_out; c2.ldo = ext.ldo_out; c2.inner_o = pl.inner; c2.w_out = attn->w_out; c2.b_out = attn->b_out;
  HN_REQUIRE(ff->w1 && ff->b1 && ff->w2 && ff->b2, HN_E_NULL, "ff: weight pointer is NULL");
  c2.has_ff = 1; c2.gate = ff->gate; c2.f_nw = ff->norm_w; c2.f_nb = ff->norm_b; c2.w1 = ff->w1; c2.b1 = ff->b1; c2.w2 = ff->w2; c2.b2 = ff->b2;
  return launch_latent_chain(c2, s);
}

size_t hn_latent_block_bwd_workspace_bytes(const hn_attn_params *attn, const hn_ff_params *ff, int b, int L) {
  if (!attn || !ff) return 0;
  const size_t a = hn_attn_bwd_workspace_bytes(attn, 0, 0, b, L, L, attn->query_dim, 0), f = hn_ff_bwd_workspace_bytes(ff, b * L);
  if (a == 0 || f == 0) return 0;
  return (a > f ? a : f) + align_up((size_t)b * L * attn->query_dim * sizeof(float), 256);
}

int hn_latent_block_bwd(const hn_attn_params *attn, const hn_ff_params *ff, const float *x_in, const float *x_mid, int b, int L,
                        const float *stats, const float *saved, const float *dy, float *dx, const hn_attn_grads *attn_grads,
                        const hn_ff_grads *ff_grads, void *workspace, size_t workspace_bytes, void *stream) {
  HN_REQUIRE(attn && ff && x_in && x_mid && stats && saved && dy && dx && attn_grads && ff_grads, HN_E_NULL, "latent_block_bwd: NULL pointer");
  const size_t need = hn_latent_block_bwd_workspace_bytes(attn, ff, b, L);
  int rc = check_ws(workspace, workspace_bytes, need, "latent_block_bwd");
  if (rc != HN_OK) return rc;
  const size_t dmid_bytes = align_up((size_t)b * L * attn->query_dim * sizeof(float), 256);
  float *dmid = (float *)workspace;
  void *op = (char *)workspace + dmid_bytes;
  // x_out = x_mid + FF(LN x_mid);  x_mid = x_in + Attn(LN x_in): the two block backwards in reverse order
  if ((rc = ff_bwd_impl(ff, x_mid, dy, dmid, 1, b * L, ff_grads, op, workspace_bytes - dmid_bytes, (hipStream_t)stream)) != HN_OK) return rc;
  return attn_bwd_impl(attn, x_in, x_mid, 1, nullptr, 0, b, L, L, attn->query_dim, nullptr, stats, saved, dmid, dx, attn_grads, op,
                       workspace_bytes - dmid_bytes, (hipStream_t)stream);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Staged models (SURVEY.md 8 "next": the reference's tuned shapes, config/best_hyperparams.yml: l_d = 119 / 126 / 62 / 65, ONE
// cross head of 16 / 63 / 27 / 103, 25 / 17 / 17 / 16 latents, dropout on).  The latent chains (chain.hip / bchain.hip) are built
// for l_d = 128, head widths of 16 / 32 / 64 / 128 and 16-row tiles.  A model outside those shapes that FITS them after zero
// padding is run as its padded image: once per forward one table-driven launch (stage_kernel) copies every latent-side weight
// into a zero-padded shadow (l_d -> 128, dim_head -> 16 / 32 / 64 / 128 per head, projection rows -> a multiple of 128), the
// shadow model -- hn_model with the staged-layout fields set: LayerNorm statistics over the valid width, softmax scale of the
// valid head width -- runs the fast path on row-padded internal buffers, and the results are un-padded on the way out
// (embeddings, trace slots; in the backward the padded gradients are accumulated onto the real ones by the same kernel).
// Exact: every pad entry is zero and stays zero (zero weight rows / columns, gamma = beta = 0 beyond the valid width), the
// gradients of pad entries are never read.  A training forward keeps the staged weights on the tape for its backward.
// HN_NO_STAGING=1: development switch (the generic per-block route these shapes took before).
// ------------------------------------------------------------------------------------------------
namespace hn {
namespace {

static bool staging_disabled() { static const bool off = getenv("HN_NO_STAGING") != nullptr; return off; }

struct Stager {
  hn_model sm;                                   // the shadow descriptor
  std::vector<hn_attn_params> ca, sa;
  std::vector<hn_ff_params> cf, sf;
  hn_model_grads sg;                             // shadow gradients (same layout as the shadow weights)
  std::vector<hn_attn_grads> gca, gsa;
  std::vector<hn_ff_grads> gcf, gsf;
  std::vector<StagePiece> fwd, bwd;              // real weight -> shadow ; shadow gradient -> real gradient (accumulated)
  std::unordered_map<const float *, size_t> seen;  // tied weights share one shadow (and one shadow gradient)
  size_t floats;                                 // shadow region, in floats
  float *wbase, *gbase;
  int ld;                                        // the real latent width

  size_t take(size_t n) { const size_t o = floats; floats += align_up(n, 64); return o; }
  // one matrix: real (rows_r, cols_r) of pitch ld_r at `real` -> shadow rectangle (rows_s, cols_s) of pitch ld_s at slot offset
  // `at` of a shadow slot starting at `slot`; the gradient travels the other way.  greal == NULL: no gradient wanted.
  void piece(size_t slot, size_t at, const float *real, float *greal, int rows_r, int cols_r, int ld_r, int rows_s, int cols_s, int ld_s,
             bool want_grads) {
    if (wbase) fwd.push_back({real, wbase + slot + at, rows_r, cols_r, ld_r, rows_s, cols_s, ld_s});
    if (want_grads && greal && gbase) bwd.push_back({gbase + slot + at, greal, rows_r, cols_r, ld_s, rows_r, cols_r, ld_r});
  }
  // a whole parameter in `n` pieces; returns the shadow pointer (NULL for a NULL parameter) and the shadow gradient pointer
  struct Slot { size_t off; bool fresh; };
  Slot slot_for(const float *real, size_t n) {
    const size_t off = take(n);                  // the layout never depends on pointer values (size queries see fake ones)
    auto it = seen.find(real);
    if (it != seen.end()) return {it->second, false};
    seen[real] = off;
    return {off, true};
  }
  const float *wptr(const float *real, size_t off) const { return real ? (wbase ? wbase : (float *)256) + off : nullptr; }
  float *gptr(const float *greal, size_t off) const { return (greal && gbase) ? gbase + off : nullptr; }
  // a vector of the latent width (LayerNorm affine, biases, ...): (1, n_r) -> (1, n_s)
  void vec(const float *real, float *greal, int n_r, int n_s, const float **w_out, float **g_out, bool grads) {
    if (!real) { *w_out = nullptr; if (g_out) *g_out = nullptr; take(n_s); return; }
    const Slot sl = slot_for(real, n_s);
    if (sl.fresh) piece(sl.off, 0, real, greal, 1, n_r, n_r, 1, n_s, n_s, grads);
    *w_out = wptr(real, sl.off);
    if (g_out) *g_out = gptr(greal, sl.off);
  }

  void attn(const hn_attn_params &p, const hn_attn_grads *g, bool cross, int D, hn_attn_params *q, hn_attn_grads *qg, bool grads) {
    *q = p;
    const int H = p.heads, dh = p.dim_head, dhp = pad_head_dim(dh), inner_r = H * dh, inner_s = H * dhp, ip = up128(inner_s);
    q->dim_head = dhp; q->dim_head_valid = dh; q->query_dim = 128; q->query_dim_valid = ld;
    hn_attn_grads zero;
    memset(&zero, 0, sizeof(zero));
    const hn_attn_grads &gr = g ? *g : zero;
    if (qg) *qg = gr;                            // (the context-side entries stay the real ones)
    vec(p.norm_w, gr.norm_w, ld, 128, &q->norm_w, qg ? &qg->norm_w : nullptr, grads);
    vec(p.norm_b, gr.norm_b, ld, 128, &q->norm_b, qg ? &qg->norm_b : nullptr, grads);
    vec(p.b_out, gr.b_out, ld, 128, &q->b_out, qg ? &qg->b_out : nullptr, grads);
    {   // w_q (H dh, l_d) -> (ip, 128): head h at rows h dhp; the last head's rectangle runs to row ip
      const Slot sl = slot_for(p.w_q, (size_t)ip * 128);
      if (p.w_q && sl.fresh)
        for (int h = 0; h < H; ++h)
          piece(sl.off, (size_t)h * dhp * 128, p.w_q + (size_t)h * dh * ld, gr.w_q ? gr.w_q + (size_t)h * dh * ld : nullptr, dh, ld, ld,
                h == H - 1 ? dhp + ip - inner_s : dhp, 128, 128, grads);
      q->w_q = wptr(p.w_q, sl.off);
      if (qg) qg->w_q = gptr(gr.w_q, sl.off);
    }
    {   // w_out (l_d, H dh) -> (128, ip): head h at columns h dhp; the last head's rectangle runs to column ip
      const Slot sl = slot_for(p.w_out, (size_t)128 * ip);
      if (p.w_out && sl.fresh)
        for (int h = 0; h < H; ++h)
          piece(sl.off, (size_t)h * dhp, p.w_out + (size_t)h * dh, gr.w_out ? gr.w_out + (size_t)h * dh : nullptr, ld, dh, inner_r, 128,
                h == H - 1 ? dhp + ip - inner_s : dhp, ip, grads);
      q->w_out = wptr(p.w_out, sl.off);
      if (qg) qg->w_out = gptr(gr.w_out, sl.off);
    }
    if (cross) {   // to_kv reads the CONTEXT (D columns, unchanged): only the head rows move, and only when the head width is padded
      if (dh == dhp) { take((size_t)2 * inner_s * D); return; }
      const Slot sl = slot_for(p.w_kv, (size_t)2 * inner_s * D);
      if (p.w_kv && sl.fresh)
        for (int j = 0; j < 2 * H; ++j)
          piece(sl.off, (size_t)j * dhp * D, p.w_kv + (size_t)j * dh * D, gr.w_kv ? gr.w_kv + (size_t)j * dh * D : nullptr, dh, D, D, dhp, D, D, grads);
      q->w_kv = wptr(p.w_kv, sl.off);
      if (qg) qg->w_kv = gptr(gr.w_kv, sl.off);
    } else {       // latent self-attention: (2 H dh, l_d) -> (2 H dhp rounded up to 128, 128)
      const int kvp = up128(2 * inner_s);
      const Slot sl = slot_for(p.w_kv, (size_t)kvp * 128);
      if (p.w_kv && sl.fresh)
        for (int j = 0; j < 2 * H; ++j)
          piece(sl.off, (size_t)j * dhp * 128, p.w_kv + (size_t)j * dh * ld, gr.w_kv ? gr.w_kv + (size_t)j * dh * ld : nullptr, dh, ld, ld,
                j == 2 * H - 1 ? dhp + kvp - 2 * inner_s : dhp, 128, 128, grads);
      q->w_kv = wptr(p.w_kv, sl.off);
      if (qg) qg->w_kv = gptr(gr.w_kv, sl.off);
    }
  }

  void ff(const hn_ff_params &p, const hn_ff_grads *g, hn_ff_params *q, hn_ff_grads *qg, bool grads) {
    *q = p;
    q->dim = 128; q->dim_valid = ld;
    hn_ff_grads zero;
    memset(&zero, 0, sizeof(zero));
    const hn_ff_grads &gr = g ? *g : zero;
    if (qg) *qg = gr;
    const int hid = 4 * ld;
    vec(p.norm_w, gr.norm_w, ld, 128, &q->norm_w, qg ? &qg->norm_w : nullptr, grads);
    vec(p.norm_b, gr.norm_b, ld, 128, &q->norm_b, qg ? &qg->norm_b : nullptr, grads);
    vec(p.b2, gr.b2, ld, 128, &q->b2, qg ? &qg->b2 : nullptr, grads);
    {   // net.0.weight (8 l_d, l_d) -> (1024, 128): value rows at 0, gate rows at 512
      const Slot sl = slot_for(p.w1, (size_t)1024 * 128);
      if (p.w1 && sl.fresh) {
        piece(sl.off, 0, p.w1, gr.w1, hid, ld, ld, 512, 128, 128, grads);
        piece(sl.off, (size_t)512 * 128, p.w1 + (size_t)hid * ld, gr.w1 ? gr.w1 + (size_t)hid * ld : nullptr, hid, ld, ld, 512, 128, 128, grads);
      }
      q->w1 = wptr(p.w1, sl.off);
      if (qg) qg->w1 = gptr(gr.w1, sl.off);
    }
    {   // net.0.bias (8 l_d) -> (1024)
      const Slot sl = slot_for(p.b1, 1024);
      if (p.b1 && sl.fresh) {
        piece(sl.off, 0, p.b1, gr.b1, 1, hid, hid, 1, 512, 512, grads);
        piece(sl.off, 512, p.b1 + hid, gr.b1 ? gr.b1 + hid : nullptr, 1, hid, hid, 1, 512, 512, grads);
      }
      q->b1 = wptr(p.b1, sl.off);
      if (qg) qg->b1 = gptr(gr.b1, sl.off);
    }
    {   // net.2.weight (l_d, 4 l_d) -> (128, 512)
      const Slot sl = slot_for(p.w2, (size_t)128 * 512);
      if (p.w2 && sl.fresh) piece(sl.off, 0, p.w2, gr.w2, ld, hid, hid, 128, 512, 512, grads);
      q->w2 = wptr(p.w2, sl.off);
      if (qg) qg->w2 = gptr(gr.w2, sl.off);
    }
  }

  // wb / gb: where the shadow weights / shadow gradients live (NULL: layout only -- a size query); g: the caller's gradients
  void build(const hn_model *m, const hn_model_grads *g, float *wb, float *gb) {
    const int M = m->n_modalities, depth = m->depth;
    const bool grads = g != nullptr;
    ld = m->l_d; floats = 0; wbase = wb; gbase = gb;
    fwd.clear(); bwd.clear(); seen.clear();
    ca.assign((size_t)depth * M, hn_attn_params()); cf.assign((size_t)depth * M, hn_ff_params());
    sa.assign((size_t)depth, hn_attn_params()); sf.assign((size_t)depth, hn_ff_params());
    gca.assign((size_t)depth * M, hn_attn_grads()); gcf.assign((size_t)depth * M, hn_ff_grads());
    gsa.assign((size_t)depth, hn_attn_grads()); gsf.assign((size_t)depth, hn_ff_grads());
    sm = *m;
    sm.l_d = 128; sm.l_d_valid = ld;
    memset(&sg, 0, sizeof(sg));
    for (int k = 0; k < depth * M; ++k) {
      const int i = k % M;
      const int D = m->channel_dims[i] + (m->fourier_encode_data ? m->num_spatial_axes[i] * (2 * m->num_freq_bands + 1) : 0);
      attn(m->cross_attn[k], (g && g->cross_attn) ? &g->cross_attn[k] : nullptr, true, D, &ca[k], &gca[k], grads);
      ff(m->cross_ff[k], (g && g->cross_ff) ? &g->cross_ff[k] : nullptr, &cf[k], &gcf[k], grads);
    }
    if (m->self_per_cross_attn > 0)
      for (int k = 0; k < depth; ++k) {
        attn(m->self_attn[k], (g && g->self_attn) ? &g->self_attn[k] : nullptr, false, 0, &sa[k], &gsa[k], grads);
        ff(m->self_ff[k], (g && g->self_ff) ? &g->self_ff[k] : nullptr, &sf[k], &gsf[k], grads);
      }
    sm.cross_attn = ca.data(); sm.cross_ff = cf.data(); sm.self_attn = sa.data(); sm.self_ff = sf.data();
    {   // latents (l_c, l_d) -> (l_c, 128)
      const Slot sl = slot_for(m->latents, (size_t)m->l_c * 128);
      if (m->latents && sl.fresh) piece(sl.off, 0, m->latents, g ? g->latents : nullptr, m->l_c, ld, ld, m->l_c, 128, 128, grads);
      sm.latents = wptr(m->latents, sl.off);
      sg.latents = gptr(g ? g->latents : nullptr, sl.off);
    }
    vec(m->head_norm_w, g ? g->head_norm_w : nullptr, ld, 128, &sm.head_norm_w, &sg.head_norm_w, grads);
    vec(m->head_norm_b, g ? g->head_norm_b : nullptr, ld, 128, &sm.head_norm_b, &sg.head_norm_b, grads);
    {   // to_logits weight (out_dims, l_d) -> (out_dims, 128)
      const int od = m->out_dims > 0 ? m->out_dims : 1;
      const Slot sl = slot_for(m->head_w, (size_t)od * 128);
      if (m->head_w && sl.fresh) piece(sl.off, 0, m->head_w, g ? g->head_w : nullptr, od, ld, ld, od, 128, 128, grads);
      sm.head_w = wptr(m->head_w, sl.off);
      sg.head_w = gptr(g ? g->head_w : nullptr, sl.off);
    }
    sg.head_b = g ? g->head_b : nullptr;
    sg.cross_attn = gca.data(); sg.cross_ff = gcf.data(); sg.self_attn = gsa.data(); sg.self_ff = gsf.data();
  }

  int run(const std::vector<StagePiece> &pieces, int accumulate, hipStream_t s) const {
    StageTable t;
    for (size_t i = 0; i < pieces.size(); i += STAGE_MAX) {
      t.n = (int)((pieces.size() - i) < (size_t)STAGE_MAX ? pieces.size() - i : (size_t)STAGE_MAX);
      t.accumulate = accumulate;
      memcpy(t.p, pieces.data() + i, (size_t)t.n * sizeof(StagePiece));
      int rc = launch_stage(t, s);
      if (rc != HN_OK) return rc;
    }
    return HN_OK;
  }
};

// Does this model run as its padded image?  Yes when it is NOT already one of the chain's shapes but fits them after padding.
static bool stage_wanted(const hn_model *m) {
  if (!m || staging_disabled() || chain_disabled() || m->l_d_valid > 0) return false;
  if (m->l_d < 1 || m->l_d > 128 || m->l_c < 1 || m->depth < 1 || m->n_modalities < 1 || m->n_modalities > 16) return false;
  if (!m->cross_attn || !m->cross_ff || !m->channel_dims || !m->num_spatial_axes) return false;
  if (m->self_per_cross_attn < 0 || m->self_per_cross_attn > 1) return false;
  bool need = m->l_d != 128 || m->l_c % 16 != 0;
  auto fits = [&](const hn_attn_params &a, const hn_ff_params &f) {
    if (a.query_dim != m->l_d || f.dim != m->l_d || a.heads < 1 || a.dim_head < 1 || a.query_dim_valid > 0 || f.dim_valid > 0) return false;
    const int dhp = pad_head_dim(a.dim_head);
    if (dhp == 0 || a.heads * dhp > 512) return false;
    if (dhp != a.dim_head || (a.heads * dhp) % 128 != 0) need = true;
    return true;
  };
  for (int k = 0; k < m->depth * m->n_modalities; ++k)
    if (!fits(m->cross_attn[k], m->cross_ff[k])) return false;
  if (m->self_per_cross_attn > 0) {
    if (!m->self_attn || !m->self_ff) return false;
    for (int k = 0; k < m->depth; ++k)
      if (!fits(m->self_attn[k], m->self_ff[k])) return false;
  }
  return need;
}

static thread_local Stager g_stager;

}  // namespace
}  // namespace hn

extern "C" {

int hn_fusion_is_staged(const hn_model *m) { return stage_wanted(m) ? 1 : 0; }

size_t hn_fusion_workspace_bytes(const hn_model *m, const hn_modality_input *in, int b) {
  if (!stage_wanted(m)) return impl_fusion_workspace_bytes(m, in, b);
  Stager &st = g_stager;
  st.build(m, nullptr, nullptr, nullptr);
  const size_t inner = impl_fusion_workspace_bytes(&st.sm, in, b);
  if (inner == 0) return 0;
  const size_t xn = align_up(rows16((size_t)b * m->l_c) * 128 * sizeof(float), 256);
  const size_t n_slots = (size_t)m->depth * (m->n_modalities + 1);
  return align_up(st.floats * sizeof(float), 256) + (n_slots + 1) * xn + inner;      // shadow weights | trace slots | output | inner
}

size_t hn_context_split_floats(const hn_model *m, int b) {
  if (!m || b <= 0) return 0;
  size_t worst = 0;
  for (int i = 0; i < m->depth * m->n_modalities; ++i) {
    const hn_attn_params &a = m->cross_attn[i];
    const size_t f = (size_t)b * m->l_c * ((size_t)a.heads * a.dim_head + 2 * (size_t)a.heads);
    if (f > worst) worst = f;
  }
  return (worst + 63) / 64 * 64;
}

int hn_fusion_forward_cp(const hn_model *m, const hn_modality_input *in, int b, int return_embeddings, const hn_context_split *cp,
                         float *out, void *workspace, size_t workspace_bytes, void *stream) {
  { const int prc = cluster_poll("hn_fusion_forward_cp", (hipStream_t)stream); if (prc != HN_OK) return prc; }
  HN_REQUIRE(m && in && cp && out, HN_E_NULL, "fusion_cp: NULL pointer");
  HN_REQUIRE(cp->n_parts >= 1 && cp->local && cp->parts && cp->exchange, HN_E_NULL, "fusion_cp: exchange buffers / callback missing");
  HN_REQUIRE(m->n_modalities <= 16, HN_E_UNSUPPORTED, "fusion_cp: %d modalities", m->n_modalities);
  HN_REQUIRE(!stage_wanted(m), HN_E_UNSUPPORTED, "fusion_cp: staged models take the block-level entry points");
  for (int i = 0; i < m->n_modalities; ++i) HN_REQUIRE(in[i].data, HN_E_UNSUPPORTED, "fusion_cp: modality %d is missing", i);
  return impl_fusion_forward(m, in, b, nullptr, 0, return_embeddings, out, nullptr, nullptr, workspace, workspace_bytes, stream, nullptr, cp);
}

int hn_fusion_forward(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                      int return_embeddings, float *out, float **attn_stats, float **x_trace, void *workspace,
                      size_t workspace_bytes, void *stream, hn_profile *prof) {
  { const int prc = cluster_poll("hn_fusion_forward", (hipStream_t)stream); if (prc != HN_OK) return prc; }
  if (!stage_wanted(m))
    return impl_fusion_forward(m, in, b, mask, skip_self_on_missing, return_embeddings, out, attn_stats, x_trace, workspace, workspace_bytes,
                               stream, prof);
  hipStream_t s = (hipStream_t)stream;
  HN_REQUIRE(out && in, HN_E_NULL, "fusion: NULL pointer");
  Stager &st = g_stager;
  st.build(m, nullptr, nullptr, nullptr);
  const size_t wbytes = align_up(st.floats * sizeof(float), 256);
  const size_t xn = align_up(rows16((size_t)b * m->l_c) * 128 * sizeof(float), 256);
  const int n_slots = m->depth * (m->n_modalities + 1);
  const size_t head = wbytes + ((size_t)n_slots + 1) * xn;
  int rc = check_ws(workspace, workspace_bytes, head + 256, "fusion");
  if (rc != HN_OK) return rc;
  char *base = (char *)workspace;
  st.build(m, nullptr, (float *)base, nullptr);
  if ((rc = st.run(st.fwd, 0, s)) != HN_OK) return rc;
  const bool emb = return_embeddings || !m->final_classifier_head;
  float *out_pad = (float *)(base + wbytes + (size_t)n_slots * xn);
  static thread_local std::vector<float *> xt;
  xt.assign((size_t)n_slots, nullptr);
  if (x_trace)
    for (int i = 0; i < n_slots; ++i) {
      const int j = i % (m->n_modalities + 1);      // (slots of blocks that cannot run are left alone, as on the direct route)
      const bool live = j < m->n_modalities ? in[j].data != nullptr : m->self_per_cross_attn > 0;
      if (x_trace[i] && live) xt[i] = (float *)(base + wbytes + (size_t)i * xn);
    }
  rc = impl_fusion_forward(&st.sm, in, b, mask, skip_self_on_missing, return_embeddings, emb ? out_pad : out, attn_stats,
                           x_trace ? xt.data() : nullptr, base + head, workspace_bytes - head, stream, prof);
  if (rc != HN_OK) return rc;
  // un-pad what leaves: the embeddings and the trace slots, (b l_c, 128) -> (b l_c, l_d), one launch
  std::vector<StagePiece> outp;
  const int rows = b * m->l_c;
  if (emb) outp.push_back({out_pad, out, rows, m->l_d, 128, rows, m->l_d, m->l_d});
  if (x_trace)
    for (int i = 0; i < n_slots; ++i)
      if (x_trace[i] && xt[i]) outp.push_back({xt[i], x_trace[i], rows, m->l_d, 128, rows, m->l_d, m->l_d});
  return st.run(outp, 0, s);
}

size_t hn_fusion_tape_bytes(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self_on_missing) {
  if (!stage_wanted(m)) return impl_fusion_tape_bytes(m, in, b, masked, skip_self_on_missing);
  Stager &st = g_stager;
  st.build(m, nullptr, nullptr, nullptr);
  const size_t inner = impl_fusion_tape_bytes(&st.sm, in, b, masked, skip_self_on_missing);
  if (inner == 0) return 0;
  const size_t n_slots = (size_t)m->depth * (m->n_modalities + 1);
  // staged weights (the backward reads them back) | the shadow model's tape | the attention blocks' inputs un-padded (hn_attn_probs)
  return align_up(st.floats * sizeof(float), 256) + inner + n_slots * align_up((size_t)b * m->l_c * m->l_d * sizeof(float), 256);
}

int hn_fusion_tape_layout(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self_on_missing,
                          size_t *stats_off, size_t *x_off) {
  if (!stage_wanted(m)) return impl_fusion_tape_layout(m, in, b, masked, skip_self_on_missing, stats_off, x_off);
  Stager &st = g_stager;
  st.build(m, nullptr, nullptr, nullptr);
  int rc = impl_fusion_tape_layout(&st.sm, in, b, masked, skip_self_on_missing, stats_off, x_off);
  if (rc != HN_OK) return rc;
  const size_t inner = impl_fusion_tape_bytes(&st.sm, in, b, masked, skip_self_on_missing);
  HN_REQUIRE(inner != 0, HN_E_SHAPE, "fusion_tape_layout: tape size");
  const size_t wfloats = align_up(st.floats * sizeof(float), 256) / sizeof(float);
  const size_t xreal = wfloats + inner / sizeof(float), xstride = align_up((size_t)b * m->l_c * m->l_d * sizeof(float), 256) / sizeof(float);
  const int n_slots = m->depth * (m->n_modalities + 1);
  for (int i = 0; i < n_slots; ++i) {
    if (stats_off[i] == (size_t)-1) continue;
    stats_off[i] += wfloats;
    x_off[i] = xreal + (size_t)i * xstride;
  }
  return HN_OK;
}

int hn_fusion_forward_train(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                            int return_embeddings, float *out, float **attn_stats, float **x_trace, void *tape,
                            size_t tape_bytes, void *workspace, size_t workspace_bytes, void *stream) {
  { const int prc = cluster_poll("hn_fusion_forward_train", (hipStream_t)stream); if (prc != HN_OK) return prc; }
  if (!stage_wanted(m))
    return impl_fusion_forward_train(m, in, b, mask, skip_self_on_missing, return_embeddings, out, attn_stats, x_trace, tape, tape_bytes,
                                     workspace, workspace_bytes, stream);
  hipStream_t s = (hipStream_t)stream;
  HN_REQUIRE(out && tape && in, HN_E_NULL, "fusion_forward_train: NULL pointer");
  HN_REQUIRE(((uintptr_t)tape & 255) == 0, HN_E_WORKSPACE, "fusion_forward_train: tape must be 256-byte aligned");
  Stager &st = g_stager;
  st.build(m, nullptr, (float *)tape, nullptr);
  const size_t wbytes = align_up(st.floats * sizeof(float), 256);
  const size_t inner = impl_fusion_tape_bytes(&st.sm, in, b, mask != nullptr, skip_self_on_missing);
  HN_REQUIRE(inner != 0, HN_E_SHAPE, "fusion_forward_train: tape size");
  const int n_slots = m->depth * (m->n_modalities + 1);
  const size_t xstride = align_up((size_t)b * m->l_c * m->l_d * sizeof(float), 256);
  HN_REQUIRE(tape_bytes >= wbytes + inner + (size_t)n_slots * xstride, HN_E_WORKSPACE, "fusion_forward_train: tape %zu bytes < required %zu",
             tape_bytes, wbytes + inner + (size_t)n_slots * xstride);
  int rc = st.run(st.fwd, 0, s);
  if (rc != HN_OK) return rc;
  // (the output leaves through the padded slot at the head of the workspace, the inner call gets the rest)
  const bool emb = return_embeddings || !m->final_classifier_head;
  const size_t xn = align_up(rows16((size_t)b * m->l_c) * 128 * sizeof(float), 256);
  const size_t head = emb ? xn : 0;
  if ((rc = check_ws(workspace, workspace_bytes, head + 256, "fusion_forward_train")) != HN_OK) return rc;
  float *out_pad = (float *)workspace;
  char *itape = (char *)tape + wbytes;
  rc = impl_fusion_forward_train(&st.sm, in, b, mask, skip_self_on_missing, return_embeddings, emb ? out_pad : out, attn_stats, nullptr, itape,
                                 inner, (char *)workspace + head, workspace_bytes - head, stream);
  if (rc != HN_OK) return rc;
  static thread_local std::vector<size_t> so, xo;
  so.assign((size_t)n_slots, 0); xo.assign((size_t)n_slots, 0);
  if ((rc = impl_fusion_tape_layout(&st.sm, in, b, mask != nullptr, skip_self_on_missing, so.data(), xo.data())) != HN_OK) return rc;
  std::vector<StagePiece> outp;
  const int rows = b * m->l_c;
  if (emb) outp.push_back({out_pad, out, rows, m->l_d, 128, rows, m->l_d, m->l_d});
  for (int i = 0; i < n_slots; ++i) {
    if (xo[i] == (size_t)-1) continue;
    const float *src = (const float *)itape + xo[i];
    outp.push_back({src, (float *)(itape + inner + (size_t)i * xstride), rows, m->l_d, 128, rows, m->l_d, m->l_d});
    if (x_trace && x_trace[i]) outp.push_back({src, x_trace[i], rows, m->l_d, 128, rows, m->l_d, m->l_d});
  }
  return st.run(outp, 0, s);
}

size_t hn_fusion_backward_workspace_bytes(const hn_model *m, const hn_modality_input *in, int b, int masked) {
  if (!stage_wanted(m)) return impl_fusion_backward_workspace_bytes(m, in, b, masked);
  Stager &st = g_stager;
  st.build(m, nullptr, nullptr, nullptr);
  const size_t inner = impl_fusion_backward_workspace_bytes(&st.sm, in, b, masked);
  if (inner == 0) return 0;
  // shadow gradients | the padded output gradient | inner
  return align_up(st.floats * sizeof(float), 256) + align_up(rows16((size_t)b * m->l_c) * 128 * sizeof(float), 256) + inner;
}

int hn_fusion_backward(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                       int return_embeddings, const float *dout, const void *tape, const hn_model_grads *g, void *workspace,
                       size_t workspace_bytes, void *stream, const hn_grad_ready *ready) {
  { const int prc = cluster_poll("hn_fusion_backward", (hipStream_t)stream); if (prc != HN_OK) return prc; }
  if (!stage_wanted(m))
    return impl_fusion_backward(m, in, b, mask, skip_self_on_missing, return_embeddings, dout, tape, g, workspace, workspace_bytes, stream, ready);
  hipStream_t s = (hipStream_t)stream;
  HN_REQUIRE(dout && tape && g && in, HN_E_NULL, "fusion_backward: NULL pointer");
  Stager &st = g_stager;
  st.build(m, nullptr, nullptr, nullptr);
  const size_t wbytes = align_up(st.floats * sizeof(float), 256);
  const size_t xn = align_up(rows16((size_t)b * m->l_c) * 128 * sizeof(float), 256);
  const size_t head = wbytes + xn;
  int rc = check_ws(workspace, workspace_bytes, head + 256, "fusion_backward");
  if (rc != HN_OK) return rc;
  // the staged weights are the forward's (on the tape); the shadow gradients start at zero in the workspace
  st.build(m, g, (float *)const_cast<void *>(tape), (float *)workspace);
  if ((rc = launch_fill((float *)workspace, 0.0f, (long)(wbytes / sizeof(float)), s)) != HN_OK) return rc;
  const bool emb = return_embeddings || !m->final_classifier_head;
  const float *dout_in = dout;
  if (emb) {     // (b l_c, l_d) -> (b l_c, 128), zero pad columns
    float *dpad = (float *)((char *)workspace + wbytes);
    std::vector<StagePiece> pp;
    pp.push_back({dout, dpad, b * m->l_c, m->l_d, m->l_d, b * m->l_c, 128, 128});
    if ((rc = st.run(pp, 0, s)) != HN_OK) return rc;
    dout_in = dpad;
  }
  const size_t inner_tape = impl_fusion_tape_bytes(&st.sm, in, b, mask != nullptr, skip_self_on_missing);
  HN_REQUIRE(inner_tape != 0, HN_E_SHAPE, "fusion_backward: tape size");
  // gradient-readiness signals: the real gradients are complete only after the un-staging launch at the end
  rc = impl_fusion_backward(&st.sm, in, b, mask, skip_self_on_missing, return_embeddings, dout_in, (const char *)tape + wbytes, &st.sg,
                            (char *)workspace + head, workspace_bytes - head, stream, nullptr);
  if (rc != HN_OK) return rc;
  if ((rc = st.run(st.bwd, 1, s)) != HN_OK) return rc;
  if (ready)
    for (int idx = m->depth; idx >= 0; --idx) {
      if (ready->events && ready->events[idx]) HN_HIP_CHECK(hipEventRecord((hipEvent_t)ready->events[idx], s));
      if (ready->notify) ready->notify(idx, ready->user);
    }
  return HN_OK;
}

}  // extern "C"
