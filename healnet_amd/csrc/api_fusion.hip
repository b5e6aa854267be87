// The inference forward of the whole fusion stack (hn_fusion_forward; healnet/models/healnet.py:190-250): workspace plan, K1 (normalised
// contexts), the one-token prelude, the folded projections, and the launch schedule of the latent side -- per-block chains (chain.hip),
// layer chains with the latent self-attention inside (lchain.hip), or the unfused block sequence -- with the context split.
#include "api_internal.h"

namespace hn {

int plan_fusion(const hn_model *m, const hn_modality_input *in, int b, void *ws, size_t ws_bytes, FusionPlan *fp,
                       bool inference) {
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
    fp->z3[i] = nullptr;
    fp->x6[i] = false;
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
    // (the training forward does the same with V written straight into the tape: taby is carved for both, tabv for inference only)
    if (n == 1 && m->depth <= HN_SKINNY_MAXZ && fp->D[i] >= 512 && b <= 512) {
      bool same = true;
      for (int layer = 1; layer < m->depth; ++layer) {
        const hn_attn_params &a0 = m->cross_attn[i], &al = m->cross_attn[layer * m->n_modalities + i];
        same = same && al.heads == a0.heads && al.dim_head == a0.dim_head && al.query_dim == a0.query_dim &&
               (al.ctx_gamma != nullptr) == (a0.ctx_gamma != nullptr);
      }
      const int inner0 = ap->heads * ap->dim_head;
      if (same && inner0 >= 512) {
        fp->tab_ahead[i] = true;
        if (inference) fp->tabv[i] = ar.take<float>((size_t)m->depth * b * inner0);
        fp->taby[i] = ar.take<float>((size_t)m->depth * b * ap->query_dim);
      }
    }
    if (n > best) { best = n; fp->dominant = i; }
    bool x6 = n > 1 && !fp->ones[i] && !fp->z16[i] && fp->ldz[i] % 4 == 0;
    for (int layer = 0; layer < m->depth; ++layer) {
      AttnPlan pl;
      int rc = plan_attn(&m->cross_attn[layer * m->n_modalities + i], true, fp->ldz[i], b, m->l_c, (int)n, fp->D[i], nullptr,
                         0, &pl, 0);
      if (rc != HN_OK) return rc;
      if (pl.bytes > op_max) op_max = pl.bytes;
      x6 = x6 && !pl.rank_d && gemm_nt_x6_eligible((long)b * n, 2 * pl.heads * pl.dhp, fp->D[i]);
      if (want_bf16) {
        if ((rc = plan_attn(&m->cross_attn[layer * m->n_modalities + i], true, fp->ldz[i], b, m->l_c, (int)n, fp->D[i], nullptr,
                            0, &pl, ns)) != HN_OK) return rc;
        if (pl.bytes > op_max) op_max = pl.bytes;
      }
    }
    // a large patch bag under the explicit binding: its K/V projections (and their weight gradients) run fp32-exact on the bf16
    // pipe from a three-plane image of the normalised rows, built once per forward behind the encode (gemm_x6.hip)
    // (the backward holds the TRANSPOSED image in the same place: G = dKV^T z contracts over the rows)
    fp->x6[i] = x6;
    if (x6) {
      const size_t nt = x6_plane_bytes((long)b * n, fp->D[i], X6_ROW_TILE), tn = gemm_tn_x6_image_bytes((long)b * n, fp->D[i] + 1, 5);
      fp->z3[i] = (uint16_t *)ar.take<char>(nt > tn ? nt : tn);
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

size_t impl_fusion_workspace_bytes(const hn_model *model, const hn_modality_input *inputs, int b) {
  FusionPlan fp;
  if (plan_fusion(model, inputs, b, nullptr, 0, &fp, true) != HN_OK) return 0;
  return fp.bytes;
}

int impl_fusion_forward(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                      int return_embeddings, float *out, float **attn_stats, float **x_trace, void *workspace,
                      size_t workspace_bytes, void *stream, hn_profile *prof, const hn_context_split *cp) {
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

  // The prelude as ONE launch (encode.hip prelude_kernel): the one-token modality's encode + the vfold roles of the first image
  // modality + that image's encode are mutually independent; as three launches in a row they were 38 us in front of the skinny
  // products that only need the first (round 6).  Candidates: fp32 inputs, no context split, the static RGB-image encode.
  int pre_tab = -1, pre_img = -1;
  EncodeCall pre_tc, pre_ic;
  {
    static const bool no_prelude = tuning_env("HN_NO_PRELUDE_MERGE") != nullptr;      // development switch (A/B)
    auto call_of = [&](int i) {
      EncodeCall c;
      c.data = in[i].data; c.dtype = in[i].dtype; c.b = b; c.n_axes = m->num_spatial_axes[i]; c.spatial = in[i].spatial; c.C = m->channel_dims[i];
      c.F = m->num_freq_bands; c.max_freq = m->max_freq; c.fourier = m->fourier_encode_data; c.normalize = 1; c.eps = 1e-5f;
      c.out = fp.z[i]; c.ld_out = fp.ldz[i]; c.ones_col = fp.ones[i] ? fp.ldz[i] - 1 : -1; c.pack_ks = fp.pack[i];
      return c;
    };
    if (!no_prelude && cp == nullptr && fp.chain && !merge_chain_disabled() && !chain_disabled()) {
      int ti = -1, ii = -1;
      for (int i = 0; i < M; ++i) {
        if (!in[i].data || fp.bf16[i] || fp.z16[i] || in[i].dtype != HN_F32) continue;
        if (ti < 0 && fp.N[i] == 1) ti = i;
        if (ii < 0 && fp.N[i] > 1 && fp.wvf[i]) ii = i;
      }
      if (ti >= 0 && ii >= 0) {
        pre_tc = call_of(ti); pre_ic = call_of(ii);
        if (encode_prelude_eligible(pre_tc, pre_ic)) { pre_tab = ti; pre_img = ii; }
      }
    }
  }
  // K1 once per forward: the normalised context of every present modality (layer independent)
  // (measured and dropped, round 4: the long modalities' encode on a side stream beside the one-token prelude -- the two
  // HBM-bound kernels slow each other down (skinny GEMMs 10 -> 18 us) and the join costs what is left: -1 % at cfg2 b = 32)
  for (int i = 0; i < M; ++i) {
    if (!in[i].data || i == pre_tab || i == pre_img) continue;      // (the prelude pair is encoded by the merged launch below)
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
      if (rc == HN_OK && fp.z3[i] && !is_split(i)) rc = launch_x6_split(fp.z[i], fp.ldz[i], nullptr, (long)b * fp.N[i], fp.D[i], X6_ROW_TILE, fp.z3[i], s);
    }
    if (rc != HN_OK) return rc;
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
    if (i == pre_img) rc = launch_encode_prelude(pre_tc, pre_ic, vf, s);      // + both encodes of the prelude pair
    else rc = launch_vfold(vf, s);
    if (rc != HN_OK) return rc;
    vmerge[i] = true;
  }
  HN_REQUIRE(pre_img < 0 || vmerge[pre_img], HN_E_UNSUPPORTED, "fusion: the prelude pair was not launched");
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
                         fp.z16[i], is_split(i) ? nullptr : fp.z3[i]);
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

}  // namespace hn
