// Internal interface between the translation units behind the C ABI (include/healnet_hip.h): the plans (workspace carves), the hooks
// the fused latent chains use to take over parts of a block, and the block-level implementations every schedule is built from.
//   api_blocks.hip   error string, kernel timers, plan_attn / attn_fwd_impl / attn_bwd_impl, feed-forward, the per-op entry points,
//                    hn_latent_block_*
//   api_fusion.hip   plan_fusion (workspace of the whole forward) and the inference schedule impl_fusion_forward
//                    (per-block chains, layer chains, context split)
//   api_train.hip    block schedule, tape plan, the tape-recording forward and the fused backward
//   api_entry.hip    staged (zero-padded) models and the hn_fusion_* entry points
#pragma once
#include "common.h"
#include <mutex>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>
#include <vector>

namespace hn {

static inline DropCfg drop_off() { return make_drop(0.0f, 0, 0, 0); }
static inline DropCfg drop_of(float p, const hn_rng &r, bool ff) {
  return p > 0.0f ? make_drop(p, r.seed, r.offset, ff ? (r.stream | DROP_SID_FF) : (r.stream & ~DROP_SID_FF), r.offset_dev) : drop_off();
}
// development switches, read ONCE per process (no getenv on the launch path)
static inline bool chain_disabled() { static const bool off = getenv("HN_NO_CHAIN") != nullptr; return off; }
static inline bool bchain_disabled() { static const bool off = getenv("HN_NO_BCHAIN") != nullptr; return off; }
static inline bool qfold_chain_disabled() { static const bool off = getenv("HN_NO_QFOLD_CHAIN") != nullptr; return off; }
static inline bool merge_chain_disabled() { static const bool off = tuning_env("HN_NO_MERGE_CHAIN") != nullptr; return off; }
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

// Steps shared by hn_attn_fwd and hn_attn_probs: the scaled query operand of the attention core and
// (explicit path) the projected keys / values.
static inline bool drop_bound_disabled() {      // development switch: dropout on the shared-context binding through the general core
  static const bool off = getenv("HN_NO_DROP_BOUND") != nullptr;
  return off;
}

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
  const uint16_t *ctx3;           // explicit binding of a large patch bag: three-plane bf16 image of the context rows (gemm_x6.hip) -> the fp32-exact K/V projection on the bf16 pipe
  int nsplit, chunk;
  int nq;                         // query tiles per wave the forward split was planned for (0: the kernel's default)
  int nsplit_bwd, chunk_bwd;      // token split of attn_bwd_dq_kernel (≈150 VGPRs: 3 waves per SIMD)
  float cscale;
  // workspace carve
  float *obuf, *q, *qf, *kv, *opart, *mpart, *lpart, *bound;
  float *wstage;                  // explicit cross binding: scratch of launch_gemm_bf16 (bf16 image of to_kv.weight + its beta row)
  size_t bytes;
};

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

struct AttnBwdPlan {
  float *dpre, *dO, *xhat, *dxhat, *lns, *delta, *dOp, *dQpart, *dQ, *dKV, *G, *cs, *Abuf, *dA, *E, *T, *dT, *dyb, *dV, *red;
  uint16_t *dkv3;            // explicit binding of a large patch bag: transposed three-plane image of dKV for G = dKV^T z on the bf16 pipe (gemm_x6.hip), or NULL
  void *fwd_ws; size_t fwd_bytes, bytes;
};

// Hooks of the fused latent backward (bchain.hip) into an attention block's backward: the chain behind the block (in backward
// order: in front of it) has already produced dpre = dy * LeakyReLU'(.) and dO = dpre W_out, and / or the chain in front of it
// will run the projection backward (dx_hat = dQ W_q + dKV W_kv, LayerNorm backward, residual) and the batched weight-gradient
// launch takes dW_q / dW_kv (and dW_out when the block's O is on the tape).
struct AttnBwdExt {
  const uint16_t *ctx3t;     // in: transposed three-plane image of the context rows with the ones column at D (gemm_x6.hip), or NULL: fp32-MFMA products
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

// ------------------------------------------------------------------------------------------------
// feed-forward block, backward
// ------------------------------------------------------------------------------------------------
struct FFBwdPlan { float *u, *h, *dh, *xhat, *dxhat, *lns, *red, *dyd; size_t bytes; };

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
  bool x6[16];         // ... planned for this modality (z3 itself is NULL in a sizing pass)
  uint16_t *z3[16];    // explicit binding of a large patch bag: three-plane bf16 image of the rows of z (gemm_x6.hip: fp32-exact products on the bf16 pipe), or NULL
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

enum StepKind { STEP_CROSS_ATTN, STEP_CROSS_FF, STEP_SELF_ATTN, STEP_SELF_FF };
struct Step { int kind, layer, m; };
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

// ------------------------------------------------------------------------------------------------
// latent block = latent self-attention + feed-forward (healnet.py:241-245), SURVEY.md 8(b) hn_latent_block_fwd / _bwd
// ------------------------------------------------------------------------------------------------
struct LatentBlockPlan { float *q, *kv, *xmid; void *op; size_t op_bytes, bytes; bool chain; };

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

// ---- block level (api_blocks.hip)
int plan_attn(const hn_attn_params *p, bool has_ctx, int ld_ctx, int b, int L, int N, int D, void *ws,
                     size_t ws_bytes, AttnPlan *pl, int bf16core = 0 /* 0: fp32 core, else the number of bf16 operand planes */);
int check_ws(void *ws, size_t ws_bytes, size_t need, const char *who);
GemmArgs gemm_defaults();
float *saved_kv(const AttnPlan &pl, bool has_ctx, bool masked, int b, int L, float *saved);
int project_ctx_kv(const hn_attn_params *p, const AttnPlan &pl, const float *ctx, int ld_ctx, int b, float *kvbuf, float *wstage,
                          const uint16_t *ctx16, hipStream_t s, const uint16_t *ctx3 = nullptr);
bool qfold_core_ok(const hn_attn_params *p, const AttnPlan &pl, int pack_ks, int L);
int attn_prepare(const hn_attn_params *p, const AttnPlan &pl, const float *x_in, const float *ctx, int ld_ctx,
                        int b, int L, hipStream_t s, AttnCoreArgs *core, int pack_ks = 0, float *kv_tape = nullptr,
                        bool kv_ready = false, bool use_bound = false, int *ext_flag = nullptr, const AttnExt *ext = nullptr);
int attn_fwd_impl(const hn_attn_params *p, const float *x_in, float *x_out, int residual, const float *ctx,
                         int ld_ctx, int b, int L, int N, int D, const uint8_t *mask, float *stats, void *ws,
                         size_t ws_bytes, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, float *o_save = nullptr,
                         bool ctx_has_ones = false, int ctx_pack_ks = 0, const Bf16Context *bc = nullptr, int *bound_flag = nullptr,
                         AttnExt *ext = nullptr, const uint16_t *ctx16 = nullptr, const uint16_t *ctx3 = nullptr);
size_t attn_saved_floats(const AttnPlan &pl, bool has_ctx, bool masked, int b, int L);
int plan_attn_bwd(const hn_attn_params *p, const AttnPlan &pl, bool has_ctx, bool masked, int b, int L, void *ws,
                         size_t ws_bytes, AttnBwdPlan *bp);
GemmExArgs gex(const float *A, long a_rs, long a_cs, const float *B, long b_rs, long b_cs, float *C, long ldc, int M,
                      int N, int K, int accumulate);
int attn_bwd_impl(const hn_attn_params *p, const float *x_in, const float *x_out, int residual, const float *ctx,
                         int ld_ctx, int b, int L, int N, int D, const uint8_t *mask, const float *stats, const float *saved,
                         const float *dy, float *dx, const hn_attn_grads *g, void *ws, size_t ws_bytes, hipStream_t s,
                         int ctx_pack_ks = 0, AttnBwdExt *ext = nullptr);
size_t ff_ws_bytes(const hn_ff_params *p, int rows);
int ff_fwd_impl(const hn_ff_params *p, const float *x_in, float *x_out, int residual, int rows, void *ws,
                       size_t ws_bytes, hipStream_t s, bool training = false);
void plan_ff_bwd(const hn_ff_params *p, int rows, void *ws, size_t ws_bytes, FFBwdPlan *pl);
int ff_bwd_impl(const hn_ff_params *p, const float *x_in, const float *dy, float *dx, int residual, int rows,
                       const hn_ff_grads *g, void *ws, size_t ws_bytes, hipStream_t s);

// ---- whole forward (api_fusion.hip)
int context_pitch(int D, int dim_head);
int plan_fusion(const hn_model *m, const hn_modality_input *in, int b, void *ws, size_t ws_bytes, FusionPlan *fp,
                       bool inference = false);
size_t impl_fusion_workspace_bytes(const hn_model *model, const hn_modality_input *inputs, int b);
int impl_fusion_forward(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                      int return_embeddings, float *out, float **attn_stats, float **x_trace, void *workspace,
                      size_t workspace_bytes, void *stream, hn_profile *prof, const hn_context_split *cp = nullptr);

// ---- training (api_train.hip)
int build_schedule(const hn_model *m, const hn_modality_input *in, int skip_self_on_missing, Step *steps, int cap);
int plan_tape(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self, const FusionPlan &fp, TapePlan *tp);
void register_transposes(const hn_model *m, const hn_modality_input *in, int b, int masked, const Step *steps, int nsteps);
size_t bchain_tn_scratch_floats(int rows);
int fusion_bwd_workspace(const hn_model *m, const hn_modality_input *in, int b, int masked, void *ws, size_t ws_bytes,
                                FusionPlan *fp, float **dX, float **head_scratch, void **op_ws, size_t *op_bytes, size_t *total,
                                float **tbuf = nullptr, size_t *tfloats = nullptr, BChainBufs *bb = nullptr);
size_t impl_fusion_tape_bytes(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self_on_missing);
void train_context_layout(const hn_model *m, const FusionPlan &fp, bool *ones, int *pack);
int impl_fusion_tape_layout(const hn_model *m, const hn_modality_input *in, int b, int masked, int skip_self_on_missing,
                          size_t *stats_off, size_t *x_off);
int impl_fusion_forward_train(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                            int return_embeddings, float *out, float **attn_stats, float **x_trace, void *tape,
                            size_t tape_bytes, void *workspace, size_t workspace_bytes, void *stream);
size_t impl_fusion_backward_workspace_bytes(const hn_model *m, const hn_modality_input *in, int b, int masked);
int impl_fusion_backward(const hn_model *m, const hn_modality_input *in, int b, const uint8_t *mask, int skip_self_on_missing,
                       int return_embeddings, const float *dout, const void *tape, const hn_model_grads *g, void *workspace,
                       size_t workspace_bytes, void *stream, const hn_grad_ready *ready);

int plan_latent_block(const hn_attn_params *ap, const hn_ff_params *fp, int b, int L, void *ws, size_t ws_bytes, LatentBlockPlan *lp);

}  // namespace hn
