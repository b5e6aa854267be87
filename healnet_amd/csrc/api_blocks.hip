// C ABI of libhealnet_hip.so, block level: error string, kernel timers, the attention / feed-forward block implementations every schedule is
// built from, the per-op entry points and hn_latent_block_* (include/healnet_hip.h; replaces healnet/models/healnet.py:292-426 op by op).
#include "api_internal.h"

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


int plan_attn(const hn_attn_params *p, bool has_ctx, int ld_ctx, int b, int L, int N, int D, void *ws,
                     size_t ws_bytes, AttnPlan *pl, int bf16core) {
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
  pl->ctx3 = nullptr;
  {
    // (fp32 staging + the three-plane image of the staged weight behind it: gemm_x6.hip)
    const size_t w16 = gemm_bf16_stage_floats(2 * pl->inner, pl->D),
                 w32 = gemm_nt_stage_floats(2 * p->heads * pl->dhp, pl->D) + x6_plane_bytes(2 * p->heads * pl->dhp, pl->D, X6_ROW_TILE) / sizeof(float) + 64;
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

int check_ws(void *ws, size_t ws_bytes, size_t need, const char *who) {
  HN_REQUIRE(ws != nullptr, HN_E_WORKSPACE, "%s: workspace is NULL (need %zu bytes)", who, need);
  HN_REQUIRE(((uintptr_t)ws & 255) == 0, HN_E_WORKSPACE, "%s: workspace must be 256-byte aligned", who);
  HN_REQUIRE(ws_bytes >= need, HN_E_WORKSPACE, "%s: workspace %zu bytes < required %zu", who, ws_bytes, need);
  return HN_OK;
}

GemmArgs gemm_defaults() {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.batch = 1;
  g.alpha = 1.0f;
  g.eps = 1e-5f;
  return g;
}

float *saved_kv(const AttnPlan &pl, bool has_ctx, bool masked, int b, int L, float *saved) {
  if (!saved || !has_ctx || pl.rank_d || (pl.N == 1 && !masked)) return nullptr;
  return saved + align_up(rows16((size_t)b * L) * pl.inner, 64);
}

// K | V = affine(ctx) W_kv^T of an explicit cross binding into `kvbuf` (pitch 2 heads dhp; pad columns of a padded head width are
// zero): the bf16 product on the bf16 context image (inference, core_precision = bf16), the LDS-DMA fp32 product on the staged
// weight (patch bags), or the generic GEMM.  `wstage`: the plan's staging scratch.
int project_ctx_kv(const hn_attn_params *p, const AttnPlan &pl, const float *ctx, int ld_ctx, int b, float *kvbuf, float *wstage,
                          const uint16_t *ctx16, hipStream_t s, const uint16_t *ctx3) {
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
    if (ctx3 && gemm_nt_x6_eligible(gk.M, np, gk.K)) {
      // ... on the bf16 pipe from three-plane images (fp32-exact, gemm_x6.hip): the bag's image was built once behind its encode,
      // the staged weight's is built here
      unsigned short *wp = (unsigned short *)(wstage + align_up(gemm_nt_stage_floats(np, gk.K), 64));
      if ((rc = launch_x6_split(ws_w, gemm_nt_ldws(gk.K), nullptr, np, gk.K, X6_ROW_TILE, wp, s)) != HN_OK) return rc;
      GemmX6Args gx{};
      gx.Ap = ctx3; gx.a_rt = x6_row_tiles(gk.M); gx.Wp = wp; gx.w_rt = x6_row_tiles(np); gx.bias = ws_b; gx.C = gk.C; gx.ldc = gk.ldc;
      gx.M = gk.M; gx.N = np; gx.KT = (gk.K + 15) / 16; gx.alpha = 1.0f;
      return launch_gemm_nt_x6(gx, gemm_nt_x6_variant(gk.M), s);
    }
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
bool qfold_core_ok(const hn_attn_params *p, const AttnPlan &pl, int pack_ks, int L) {
  return pl.rank_d && pl.ones && p->ctx_gamma != nullptr && pl.dp == 16 && pl.Lp == L && pack_ks == packed_steps(pl.D, pl.dp);
}

// kv_tape (explicit cross binding, training): the projected K / V live in the tape instead of the workspace; the forward
// writes them there, the backward (kv_ready) reads them back instead of re-running the K/V projection GEMM.
int attn_prepare(const hn_attn_params *p, const AttnPlan &pl, const float *x_in, const float *ctx, int ld_ctx,
                        int b, int L, hipStream_t s, AttnCoreArgs *core, int pack_ks, float *kv_tape,
                        bool kv_ready, bool use_bound, int *ext_flag, const AttnExt *ext) {
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
      if ((rc = project_ctx_kv(p, pl, ctx, ld_ctx, b, kvbuf, pl.wstage, pl.ctx16, s, pl.ctx3)) != HN_OK) return rc;
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

int attn_fwd_impl(const hn_attn_params *p, const float *x_in, float *x_out, int residual, const float *ctx,
                         int ld_ctx, int b, int L, int N, int D, const uint8_t *mask, float *stats, void *ws,
                         size_t ws_bytes, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, float *o_save,
                         bool ctx_has_ones, int ctx_pack_ks, const Bf16Context *bc, int *bound_flag,
                         AttnExt *ext, const uint16_t *ctx16, const uint16_t *ctx3) {
  HN_REQUIRE(x_in && (x_out || (ext && ext->defer_out)), HN_E_NULL, "attn: x is NULL");
  HN_REQUIRE(p && p->w_q && p->w_kv && p->w_out, HN_E_NULL, "attn: weight pointer is NULL");
  AttnPlan pl;
  int rc = plan_attn(p, ctx != nullptr, ld_ctx, b, L, N, D, nullptr, 0, &pl, bc ? bc->ns : 0);
  if (rc != HN_OK) return rc;
  if ((rc = check_ws(ws, ws_bytes, pl.bytes, "attn")) != HN_OK) return rc;
  if ((rc = plan_attn(p, ctx != nullptr, ld_ctx, b, L, N, D, ws, ws_bytes, &pl, bc ? bc->ns : 0)) != HN_OK) return rc;
  pl.ctx16 = o_save == nullptr ? ctx16 : nullptr;      // (training keeps the fp32 projection: the backward differentiates THAT product)
  pl.ctx3 = ctx3;                                      // (fp32-exact: both forwards)

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

// ------------------------------------------------------------------------------------------------
// attention block, backward
// ------------------------------------------------------------------------------------------------
// What the training forward keeps per attention block besides the softmax statistics:
//   explicit K/V binding : O (b*L, inner), the normalised attention output
//   rank-D binding       : P z (b*L, heads*dp), the normalised context average (O is recomputed from it)
//   one-token context    : V (b, inner)
size_t attn_saved_floats(const AttnPlan &pl, bool has_ctx, bool masked, int b, int L) {
  if (has_ctx && pl.N == 1 && !masked) return (size_t)b * pl.inner;
  if (pl.rank_d) return (size_t)b * L * pl.heads * pl.dp;
  // explicit binding: O, and for a cross block also the projected K / V (288 GB of HBM: keeping 134 MB per WSI-bag block
  // at cfg4 is cheaper than re-running its 52 GF projection in the backward)
  return align_up(rows16((size_t)b * L) * pl.inner, 64) + (has_ctx ? (size_t)b * pl.N * 2 * pl.heads * pl.dhp : 0);
}

int plan_attn_bwd(const hn_attn_params *p, const AttnPlan &pl, bool has_ctx, bool masked, int b, int L, void *ws,
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
        if (gemm_tn_x6_eligible((long)b * pl.N, (int)(2 * inner), pl.D))
          bp->dkv3 = (uint16_t *)ar.take<char>(gemm_tn_x6_image_bytes((long)b * pl.N, (int)(2 * inner), 8));
      }
    }
  }
  bp->bytes = ar.off;
  if (ws != nullptr && ar.overflow) return fail(HN_E_WORKSPACE, "attn_bwd: workspace %zu bytes < required %zu", ws_bytes, ar.off);
  return HN_OK;
}

GemmExArgs gex(const float *A, long a_rs, long a_cs, const float *B, long b_rs, long b_cs, float *C, long ldc, int M,
                      int N, int K, int accumulate) {
  GemmExArgs e;
  memset(&e, 0, sizeof(e));
  e.A = A; e.a_rs = a_rs; e.a_cs = a_cs; e.B = B; e.b_rs = b_rs; e.b_cs = b_cs; e.C = C; e.ldc = ldc;
  e.M = M; e.N = N; e.K = K; e.batch = 1; e.alpha = 1.0f; e.accumulate = accumulate;
  return e;
}

int attn_bwd_impl(const hn_attn_params *p, const float *x_in, const float *x_out, int residual, const float *ctx,
                         int ld_ctx, int b, int L, int N, int D, const uint8_t *mask, const float *stats, const float *saved,
                         const float *dy, float *dx, const hn_attn_grads *g, void *ws, size_t ws_bytes, hipStream_t s,
                         int ctx_pack_ks, AttnBwdExt *ext) {
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
    bool dkv_planes = false;       // the dK/dV kernel wrote the transposed three-plane image itself (dKV is then not materialised)
    if (!has_ctx && launch_attn_bwd_self_pair(ba, dh, inner, s, &rc_pair)) {      // latent self-attention: both products in one launch
      if (rc_pair != HN_OK) return rc_pair;
      if (!dq_direct && (rc = launch_dq_reduce(bp.dQpart, pl.nsplit_bwd, b, h, L, pl.Lp, pl.dp, dh, two_scale, bp.dQ, inner, dh, s)) != HN_OK) return rc;
    } else {
      if ((rc = launch_attn_bwd_dq(ba, s)) != HN_OK) return rc;
      if (!dq_direct && (rc = launch_dq_reduce(bp.dQpart, pl.nsplit_bwd, b, h, L, pl.Lp, pl.dp, dh, two_scale, bp.dQ, inner, dh, s)) != HN_OK) return rc;
      if (has_ctx && bp.dkv3 && ext && ext->ctx3t) { ba.dkv3 = bp.dkv3; ba.dkv3_ct = x6_col_tiles(2 * inner, 8); }
      if ((rc = launch_attn_bwd_dkv(ba, dh, inner, s, &dkv_planes)) != HN_OK) return rc;
    }
    const long krows = (long)b * pl.N;
    if (has_ctx) {   // gradients of to_kv and of the context LayerNorm affine from G = dKV^T z and colsum(dKV)
      if (bp.dkv3 && ext && ext->ctx3t) {
        // ... fp32-exact on the bf16 pipe (gemm_x6.hip): the transposed three-plane image of dKV is built here, the context's (with
        // its ones column: colsum(dKV) is one more column of the product) once per backward by the caller
        const long kdim = pl.D > (int)p->query_dim ? pl.D : (long)p->query_dim;
        const size_t red_floats = reduce_scratch_floats(2L * inner * kdim, (int)(2 * inner > p->query_dim ? 2 * inner : p->query_dim));
        if (!dkv_planes && (rc = launch_x6_split_t(bp.dKV, 2 * inner, krows, 2 * inner, 8, -1, bp.dkv3, s)) != HN_OK) return rc;
        if ((rc = launch_gemm_tn_x6(bp.dkv3, ext->ctx3t, krows, 2 * inner, pl.D, bp.G, pl.D, bp.cs, bp.red, red_floats, s)) != HN_OK) return rc;
      } else {
        GemmExArgs e = gex(bp.dKV, 1, 2 * inner, ctx, 1, ld_ctx, bp.G, pl.D, 2 * inner, pl.D, (int)krows, 0);
        e.colsum = bp.cs; e.colsum_accumulate = 0;          // colsum(dKV) rides on the same pass
        if ((rc = launch_gemm_ex(e, s, bp.red)) != HN_OK) return rc;
      }
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
size_t ff_ws_bytes(const hn_ff_params *p, int rows) {
  return align_up((size_t)rows * 5 * p->dim * sizeof(float), 256);
}

int ff_fwd_impl(const hn_ff_params *p, const float *x_in, float *x_out, int residual, int rows, void *ws,
                       size_t ws_bytes, hipStream_t s, bool training) {
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

void plan_ff_bwd(const hn_ff_params *p, int rows, void *ws, size_t ws_bytes, FFBwdPlan *pl) {
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

int ff_bwd_impl(const hn_ff_params *p, const float *x_in, const float *dy, float *dx, int residual, int rows,
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

int context_pitch(int D, int dim_head) {
  const int dhp = pad_head_dim(dim_head);
  int dp = D <= 15 ? 16 : (D <= 31 ? 32 : 0);   // leave column dp-1 free for the kernel's synthetic ones column
  if (dp != 0 && dhp != 0 && dp <= dhp) return dp;
  return (D + 3) / 4 * 4;
}

// ------------------------------------------------------------------------------------------------
// latent block = latent self-attention + feed-forward (healnet.py:241-245), SURVEY.md 8(b) hn_latent_block_fwd / _bwd
// ------------------------------------------------------------------------------------------------
int plan_latent_block(const hn_attn_params *ap, const hn_ff_params *fp, int b, int L, void *ws, size_t ws_bytes, LatentBlockPlan *lp) {
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

}  // extern "C"

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
