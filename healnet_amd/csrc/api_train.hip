// Training: the executed-block schedule, the tape plan, the tape-recording forward (hn_fusion_forward_train) and the fused backward
// (hn_fusion_backward; serves healnet/main.py:425-467).
#include "api_internal.h"

namespace hn {

// the executed blocks in order, identical for the training forward and the backward (healnet.py:227-245)
int build_schedule(const hn_model *m, const hn_modality_input *in, int skip_self_on_missing, Step *steps, int cap) {
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

int plan_tape(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self, const FusionPlan &fp, TapePlan *tp) {
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
void register_transposes(const hn_model *m, const hn_modality_input *in, int b, int masked, const Step *steps, int nsteps) {
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

size_t bchain_tn_scratch_floats(int rows) {
  // upper bound over every product subset a chain can batch (dW1, dW2, dW_out, dW_q, dW_kv at inner = 512): fewer products means
  // fewer tiles and therefore MORE k-slices (up to GEMM_EX_SPLITS), so the bound is the split cap times all partial sizes
  (void)rows;
  const long MN[5][2] = {{1024, 128}, {128, 512}, {128, 512}, {512, 128}, {1024, 128}};
  size_t n = 0;
  for (int i = 0; i < 5; ++i) n += (size_t)GEMM_EX_SPLITS * (MN[i][0] * MN[i][1] + MN[i][0]) + 128;
  return n;
}

int fusion_bwd_workspace(const hn_model *m, const hn_modality_input *in, int b, int masked, void *ws, size_t ws_bytes,
                                FusionPlan *fp, float **dX, float **head_scratch, void **op_ws, size_t *op_bytes, size_t *total,
                                float **tbuf, size_t *tfloats, BChainBufs *bb) {
  // same z / x carve as the forward (x is unused), then the backward scratch
  int rc = plan_fusion(m, in, b, nullptr, 0, fp);
  if (rc != HN_OK) return rc;
  Arena ar(ws, ws_bytes);
  for (int i = 0; i < m->n_modalities; ++i) {
    fp->z[i] = nullptr;
    if (in[i].data) fp->z[i] = ar.take<float>((size_t)b * fp->N[i] * fp->ldz[i]);
    // (a large patch bag: the transposed three-plane image of its rows, gemm_x6.hip)
    fp->z3[i] = (in[i].data && fp->x6[i]) ? (uint16_t *)ar.take<char>(gemm_tn_x6_image_bytes((long)b * fp->N[i], fp->D[i] + 1, 5)) : nullptr;
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

size_t impl_fusion_tape_bytes(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self_on_missing) {
  FusionPlan fp;
  if (plan_fusion(m, in, b, nullptr, 0, &fp) != HN_OK) return 0;
  static thread_local TapePlan tp;
  if (plan_tape(m, in, b, masked, skip_self_on_missing, fp, &tp) != HN_OK) return 0;
  return align_up(tp.floats * sizeof(float), 256);
}

// Context layout of the training forward / backward: the ones column and the packed channel order of the inference forward.
// Round 4: under dropout on the probabilities too -- the ones column's accumulator is the thinned row sum the rank-D binding
// needs there anyway (the full denominator is summed on the VALU), and the packed order saves one QK^T k-step in the forward
// core and two of twelve in the dQ kernel.  (Dropping blocks need the bounded core for that: a LayerNorm-ed context.)
void train_context_layout(const hn_model *m, const FusionPlan &fp, bool *ones, int *pack) {
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

int impl_fusion_tape_layout(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self_on_missing,
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

int impl_fusion_forward_train(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
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
    if (fp.z3[i] && (rc = launch_x6_split(fp.z[i], fp.ldz[i], nullptr, (long)b * fp.N[i], fp.D[i], X6_ROW_TILE, fp.z3[i], s)) != HN_OK) return rc;
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
  // the one-token shortcut (tabular / omic modality) rides on a chain with head == 2 (below); whether the block at step k does
  auto tab_fusable = [&](int k) -> bool {
    static const bool no_tab_chain = tuning_env("HN_NO_TAB_CHAIN") != nullptr;
    const Step &st = tp.steps[k];
    if (!(use_chain && st.kind == STEP_CROSS_ATTN && k + 1 < tp.nsteps && !is_attn_t(tp.steps[k + 1]))) return false;
    const Step &sf = tp.steps[k + 1];
    const hn_ff_params &fq = sf.kind == STEP_CROSS_FF ? m->cross_ff[sf.layer * M + sf.m] : m->self_ff[sf.layer];
    const hn_attn_params &aq = m->cross_attn[st.layer * M + st.m];
    return !no_tab_chain && !staged && fp.N[st.m] == 1 && mask == nullptr && !(aq.dropout > 0.0f) && fq.dim == d && fq.dropout >= 0.0f &&
           fq.dropout < 1.0f && aq.query_dim == d && chain_ff_aligned(&fq) && al16(T + tp.x_off[k]) && al16(T + tp.x_off[k + 1]) &&
           al16(T + tp.x_off[k + 2]) && aq.w_out && aq.b_out;
  };
  // One-token modalities whose blocks ALL take that route: the block outputs y_l = LeakyReLU(W_out,l (W_v,l c_hat) + b_out,l) do not
  // depend on the latent array, so all layers' value rows (straight into their tape slots) and outputs are two batched launches ahead
  // of the layer loop, as in the inference forward (round 6: two skinny launches per layer until then, 6 x 5.6 us at cfg4)
  const float *tab_y[16];
  for (int i = 0; i < M; ++i) {
    tab_y[i] = nullptr;
    if (!in[i].data || !fp.tab_ahead[i] || !fp.taby[i] || mask != nullptr) continue;
    const hn_attn_params &a0 = m->cross_attn[i];
    const int inner = a0.heads * a0.dim_head;
    int slot[HN_SKINNY_MAXZ], found = 0;
    bool all = true;
    for (int k = 0; k < tp.nsteps; ++k) {
      const Step &st = tp.steps[k];
      if (st.kind != STEP_CROSS_ATTN || st.m != i) continue;
      all = all && tab_fusable(k) && st.layer < HN_SKINNY_MAXZ && tp.saved_off[k] != kNoSlot;
      if (st.layer < HN_SKINNY_MAXZ) slot[st.layer] = k;
      ++found;
    }
    if (!all || found != m->depth) continue;
    GemmSkinnyMulti gv, gy;
    memset(&gv, 0, sizeof(gv));
    memset(&gy, 0, sizeof(gy));
    gv.nz = gy.nz = m->depth;
    gv.lda = fp.ldz[i]; gv.ldw = fp.D[i]; gv.ldc = inner; gv.M = b; gv.N = inner; gv.K = fp.D[i];
    gv.pro = a0.ctx_gamma ? PRO_AFFINE : PRO_NONE; gv.act = ACT_NONE;
    gy.lda = inner; gy.ldw = inner; gy.ldc = a0.query_dim; gy.M = b; gy.N = a0.query_dim; gy.K = inner;
    gy.pro = PRO_NONE; gy.act = ACT_LEAKY;
    bool ok = true;
    for (int layer = 0; layer < m->depth; ++layer) {
      const hn_attn_params &al = m->cross_attn[layer * M + i];
      ok = ok && al.w_kv && al.w_out && !(al.dropout > 0.0f) && (al.ctx_gamma != nullptr) == (a0.ctx_gamma != nullptr);
      gv.A[layer] = fp.z[i]; gv.W[layer] = al.w_kv + (long)inner * fp.D[i]; gv.gamma[layer] = al.ctx_gamma; gv.beta[layer] = al.ctx_beta;
      gv.C[layer] = T + tp.saved_off[slot[layer]];                     // V of the block: what the backward reads
      gy.A[layer] = gv.C[layer]; gy.W[layer] = al.w_out; gy.bias[layer] = al.b_out;
      gy.C[layer] = fp.taby[i] + (size_t)layer * b * a0.query_dim;
    }
    if (!ok) continue;
    if ((rc = launch_gemm_skinny_multi(gv, s)) != HN_OK) return rc;
    if ((rc = launch_gemm_skinny_multi(gy, s)) != HN_OK) return rc;
    tab_y[i] = fp.taby[i];
  }
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
      fuse_tab = !fuse && tab_fusable(k);
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
      if (fuse_tab && tab_y[st.m] != nullptr)                    // computed ahead of the loop (V is on the tape already)
        ext.y_out = tab_y[st.m] + (size_t)st.layer * b * ap.query_dim;
      else if (self)
        rc = attn_fwd_impl(&ap, xin, nullptr, 1, nullptr, 0, b, L, L, d, nullptr, T + tp.stats_off[k], fp.op_ws, fp.op_ws_bytes, s,
                           nullptr, nullptr, T + tp.saved_off[k], false, 0, nullptr, nullptr, &ext);
      else
        rc = attn_fwd_impl(&ap, xin, nullptr, 1, fp.z[st.m], fp.ldz[st.m], b, L, fp.N[st.m], fp.D[st.m], mask, T + tp.stats_off[k],
                           fp.op_ws, fp.op_ws_bytes, s, nullptr, nullptr, T + tp.saved_off[k], tones[st.m], tpack[st.m], nullptr, nullptr,
                           &ext, nullptr, fp.z3[st.m]);
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
                           tones[st.m], tpack[st.m], nullptr, nullptr, extp, nullptr, fp.z3[st.m]);
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

size_t impl_fusion_backward_workspace_bytes(const hn_model *m, const hn_modality_input *in, int b, int masked) {
  FusionPlan fp;
  float *dX, *hs;
  void *op;
  size_t opb, total;
  if (fusion_bwd_workspace(m, in, b, masked, nullptr, 0, &fp, &dX, &hs, &op, &opb, &total) != HN_OK) return 0;
  return total;
}

int impl_fusion_backward(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
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
    if (tp.z_off[i] != kNoSlot && fp.z[i]) fp.z[i] = const_cast<float *>(T) + tp.z_off[i];
    else if ((rc = launch_encode(in[i].data, in[i].dtype, b, m->num_spatial_axes[i], in[i].spatial, m->channel_dims[i], m->num_freq_bands,
                                 m->max_freq, m->fourier_encode_data, 1, 1e-5f, fp.z[i], fp.ldz[i], s, tones[i] ? fp.ldz[i] - 1 : -1,
                                 tpack[i])) != HN_OK)
      return rc;
    // a large patch bag: the transposed three-plane image of its rows for the weight-gradient products of all its layers (gemm_x6.hip)
    if (fp.z3[i] && gemm_tn_x6_eligible((long)b * fp.N[i], 2 * m->cross_attn[i].heads * m->cross_attn[i].dim_head, fp.D[i])) {
      if ((rc = launch_x6_split_t(fp.z[i], fp.ldz[i], (long)b * fp.N[i], fp.D[i], 5, fp.D[i], fp.z3[i], s)) != HN_OK) return rc;
    } else fp.z3[i] = nullptr;
  }
  const float *xf = T + tp.x_off[tp.nsteps];
  const bool use_bchain = cb.ok && fp.chain && !bchain_disabled() && !chain_disabled();
  bool flags_cleared = false;                // the cluster flags of the backward chains (bchain.hip): cleared by the head's launch when there is one
  if (m->final_classifier_head && !return_embeddings) {
    if ((rc = launch_head_bwd(xf, b, L, d, m->head_norm_w, m->head_norm_b, m->head_w, m->out_dims, dout, dX, g->head_norm_w,
                              g->head_norm_b, g->head_w, g->head_b, hs, s, m->l_d_valid, use_bchain ? cb.xflags : nullptr, BCHAIN_XFLAGS)) != HN_OK) return rc;
    flags_cleared = use_bchain;
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
  int bchain_seq = 0;
  if (use_bchain && !flags_cleared && (rc = launch_fill((float *)cb.xflags, 0.0f, BCHAIN_XFLAGS, s)) != HN_OK) return rc;      // cluster flags (bchain.hip)
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
    if (st.kind == STEP_CROSS_ATTN) ext.ctx3t = fp.z3[st.m];
    if (tp.q_off[k] != kNoSlot) ext.q_taped = T + tp.q_off[k];
    if (tp.kv_off[k] != kNoSlot) ext.kv_taped = T + tp.kv_off[k];
    if (tp.xhat_off[k] != kNoSlot) ext.xhat_taped = T + tp.xhat_off[k];
    AttnBwdExt *extp = (dpre || defer || ext.q_taped || ext.xhat_taped || ext.ctx3t) ? &ext : nullptr;
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

}  // namespace hn
