// Staged (zero-padded) models and the hn_fusion_* entry points of the C ABI (include/healnet_hip.h).
#include "api_internal.h"

using namespace hn;

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
