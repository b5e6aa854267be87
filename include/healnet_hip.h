/*
 * healnet_hip.h -- C ABI of libhealnet_hip.so (gfx950 / MI355X).
 *
 * The reference (konst-int-i/healnet) has no native boundary: its hot path is the Python class
 * surface of healnet/models/healnet.py.  This header is the boundary the MI355X build puts
 * UNDERNEATH that surface; each entry point names the reference code it replaces (file:line are
 * relative to the reference repository, healnet/models/healnet.py unless stated).
 *
 * Conventions (all entry points):
 *   - plain C types only: raw DEVICE pointers (fp32, row-major, innermost dimension contiguous unless a
 *     leading dimension is passed), explicit sizes, a hipStream_t passed as void*;
 *   - return 0 on success or a negative hn_status; never throw, never exit; the message for the
 *     last failure on the calling thread is available from hn_last_error_string();
 *   - never allocate or free device memory: scratch comes from the caller-provided workspace whose
 *     size the matching hn_*_workspace_bytes() returns (256-byte aligned base required);
 *   - no global mutable state besides the thread-local error string: re-entrant across streams,
 *     devices and threads (callable from PyTorch's autograd thread);
 *   - all work is enqueued on `stream`; nothing synchronises the device.
 */
#ifndef HEALNET_HIP_H
#define HEALNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HN_ABI_VERSION 12
#define HN_MAX_AXES 4

typedef enum hn_status {
  HN_OK = 0,
  HN_E_SHAPE = -1,       /* inconsistent or non-positive sizes                                   */
  HN_E_UNSUPPORTED = -2, /* valid request the kernels do not cover (e.g. dim_head > 128)         */
  HN_E_WORKSPACE = -3,   /* workspace NULL, misaligned or smaller than hn_*_workspace_bytes()    */
  HN_E_HIP = -4,         /* a HIP runtime call / kernel launch failed                            */
  HN_E_NULL = -5,        /* a required pointer is NULL                                           */
  HN_E_CORESIDENCY = -6  /* ABI v10: a cluster-mode latent chain launched EARLIER on this device gave up waiting for a
                          * member workgroup (see hn_cluster_status); nothing was launched by the call that returns it  */
} hn_status;

/* Gate of the feed-forward block: SELU (healnet.py:328-331, snn=True) or GELU (:323-326). */
typedef enum hn_gate { HN_GATE_SELU = 0, HN_GATE_GELU = 1 } hn_gate;

/* Element type of a modality tensor handed to the fused forward (weights, latents and outputs are always fp32).
 * HN_U8 is the 8-bit image transport: the value is byte / 255 in fp32, bit-identical to torchvision's ToTensor. */
typedef enum hn_dtype { HN_F32 = 0, HN_BF16 = 1, HN_U8 = 2 } hn_dtype;

/* Matrix-instruction precision of the shared-context (image / volume) cross-attention core of hn_fusion_forward:
 * HN_CORE_F32  fp32 MFMA on the fp32 context (default; the <= 1e-3 parity configuration).  Under this setting (inference and
 *              training alike) the K/V projection of a LARGE patch bag (>= 8192 context rows per call) and its weight
 *              gradient are still fp32 products, but formed on the bf16 pipe: every fp32 operand as three bf16 planes whose sum
 *              is the operand exactly, six bf16 x bf16 products (each exact in fp32) with fp32 accumulation per fp32 product;
 *              the error against an fp64 product is not above the fp32 MFMA's (gemm_x6.hip; tests/test_gpu_x6.py,
 *              tests/test_x6_split_math.py).  HN_NO_X6_GEMM=1 in the environment takes the fp32 MFMA for those two products;
 * HN_CORE_BF16 bf16 MFMA with fp32 accumulation: context, folded queries and probabilities rounded to bf16 once
 *              (BASELINE configs[2], tolerance 2e-2 max-norm against the fp32 oracle); and, for a modality on the explicit
 *              K/V binding (patch bags: BASELINE configs[3] / [4]), the context-side K/V projection on bf16 MFMA -- context
 *              and to_kv.weight (with the context LayerNorm's gain folded in) rounded to bf16 once, fp32 accumulation, the
 *              LayerNorm's bias term kept in fp32 -- when it is large enough to matter (>= 2048 context rows, D >= 256, 2 * inner
 *              a multiple of 128); with heads of 64 (an even number of them) that projection writes bf16
 *              K / V images and the block's attention core runs on bf16 MFMA as well (scaled queries and probabilities
 *              rounded to bf16 once, fp32 softmax statistics and accumulation).  Everything else -- LayerNorms, the other
 *              projections, softmax statistics, feed-forward, head -- stays fp32.  Training always uses HN_CORE_F32.
 * HN_CORE_BF16X3 bf16 MFMA on hi + lo operand pairs (hi = bf16(v), lo = bf16(v - hi)), products hi*hi + hi*lo + lo*hi
 *              accumulated in fp32: 16 operand mantissa bits, fp32-class results (same <= 1e-3 parity bound as
 *              HN_CORE_F32; the tests run it against the same fixtures with the same tolerances). */
typedef enum hn_core_precision { HN_CORE_F32 = 0, HN_CORE_BF16 = 1, HN_CORE_BF16X3 = 2 } hn_core_precision;

int hn_abi_version(void);
/* 16 hex digits: sha256 over the compile flags, headers and translation units this library was built from (the host side
 * rebuilds when it differs from the sources next to it; bench.py prints it). */
const char *hn_build_id(void);
const char *hn_last_error_string(void);

/* ---------------------------------------------------------------------------------------------
 * K1  positional encode + concat + flatten            replaces HealNet.forward :200-222 and
 *                                                     fourier_encode :292-302
 * data (b, S_1..S_a, C) -> ctx (b, N = prod S, D = C + a*(2F+1)), data channels first, then per axis
 * [sin(p s_0 pi) .. sin(p s_{F-1} pi), cos(..) .., p].  fourier == 0 copies the data channels only
 * (fourier_encode_data=False).  ld_out >= D is the row pitch of ctx in floats.
 * ------------------------------------------------------------------------------------------- */
int hn_fourier_encode_concat(const float *data, int b, int n_axes, const int *spatial, int channels,
                             int num_freq_bands, float max_freq, int fourier, float *ctx, int ld_out,
                             void *stream);

/* Same walk, but writes the context already LayerNorm-normalised WITHOUT affine:
 * z = (ctx - mean_D) * rsqrt(var_D + eps), columns D..ld_out-1 zero filled.  This is the
 * layer-independent half of PreNorm.norm_context (:316-319); gamma/beta of each layer are applied
 * inside hn_attn_fwd (hn_attn_params.ctx_gamma / ctx_beta). */
int hn_encode_norm(const float *data, int b, int n_axes, const int *spatial, int channels,
                   int num_freq_bands, float max_freq, int fourier, float eps, float *z, int ld_out,
                   void *stream);

/* hn_encode_norm for a SLAB of a modality along its first spatial axis (context split over ranks, below): `data` holds rows
 * [axis0_begin, axis0_begin + spatial[0]) of an axis of axis0_total positions; the positional features are those of the whole
 * tensor (bit-identical to the rows hn_encode_norm writes for the same tokens). */
int hn_encode_norm_slab(const float *data, int b, int n_axes, const int *spatial, int channels,
                        int num_freq_bands, float max_freq, int fourier, float eps, float *z, int ld_out,
                        int axis0_begin, int axis0_total, void *stream);

/* Row pitch hn_fusion_forward uses for the normalised context of a modality with D encoded channels
 * when attending with `dim_head`: 16/32 for the rank-D reassociated path, else D rounded up to 4. */
int hn_context_pitch(int D, int dim_head);

/* ---------------------------------------------------------------------------------------------
 * Attention block                                      replaces PreNorm.forward :313-321 +
 *                                                      Attention.forward :400-426 (+ residual :236/:244)
 * ------------------------------------------------------------------------------------------- */
/* Counter-based dropout: whether element (row, col) of a block's mask is kept is a pure function of
 * (seed, offset, stream, row, col) -- Philox4x32 with 7 rounds, kept values scaled by 1 / (1 - p).  Feed-forward masks: one call
 * per aligned column quad of a row, keep iff word >= p * 2^32.  Attention masks: 16-bit decisions, one call per aligned column
 * quad of the row PAIR (R, R + 16) (R = row index in the (b * heads * L, N) matrix with bit 4 clear): low halves of the four
 * words decide row R, high halves row R + 16, keep iff half >= p * 2^16 (rounded down) -- inside the attention cores the
 * generator is what dropout costs, and a wave's query tiles are 16 rows apart.  `seed` is the generator seed, `offset` a per-forward counter (a fresh mask every
 * iteration; the backward is called with the forward's value), `stream` tells the blocks of a model apart: the fused
 * entry points use hn_model.rng and set stream = index of the block in execution order (feed-forward blocks have
 * bit 31 set).  The masks are reproducible bit for bit by hn_dropout_mask. */
typedef struct hn_rng {
  uint64_t seed;
  uint32_t offset;
  uint32_t stream;
  const uint32_t *offset_dev;   /* optional (ABI v7): a DEVICE word added to `offset` by every mask-drawing kernel when it runs -- a
                                 * training step captured into a HIP graph bakes `offset` into its kernel arguments, so the host
                                 * advances this word between replays instead (healnet_amd.train.GraphedStep); NULL = none */
} hn_rng;

typedef struct hn_attn_params {
  int heads, dim_head;    /* inner = heads * dim_head                                          */
  int query_dim;          /* l_d                                                               */
  const float *norm_w;    /* LayerNorm(query_dim) on x; NULL -> x is used as given             */
  const float *norm_b;
  const float *ctx_gamma; /* affine of LayerNorm(D) applied to an already normalised context;   */
  const float *ctx_beta;  /* NULL -> context is used as given                                  */
  const float *w_q;       /* to_q.weight      (inner, query_dim)                               */
  const float *w_kv;      /* to_kv.weight     (2*inner, D)   rows [0,inner) = K, rest = V      */
  const float *w_out;     /* to_out.0.weight  (query_dim, inner)                               */
  const float *b_out;     /* to_out.0.bias    (query_dim)                                      */
  float dropout;          /* nn.Dropout on the attention probabilities (:381,:421); training entry points only,  */
  hn_rng rng;             /* 0 = off.  rng: see hn_rng (the fused entry points override it per block)            */
  /* STAGED layout (0 = off; see "Staged models" below): the block's weights are zero-padded images of a narrower block -- */
  int dim_head_valid;     /* the softmax scale is dim_head_valid^-1/2; columns [dim_head_valid, dim_head) of every head are zero  */
  int query_dim_valid;    /* LayerNorm statistics over the first query_dim_valid columns of x; != 0 also promises the padded
                           * allocation: with ip = heads*dim_head rounded up to 128, w_q has ip rows, w_out row pitch ip, a
                           * latent self-attention's w_kv 2*heads*dim_head rounded up to 128 rows (the pad entries zero)   */
} hn_attn_params;

/* y = LeakyReLU_0.01( concat_h( softmax(2 * dim_head^-1/2 * Q_h K_h^T) V_h ) W_out^T + b_out )  [+ x_in]
 *   x_in  (b, L, query_dim); x_out may alias x_in;  residual != 0 adds the un-normalised x_in (:236).
 *   ctx   (b, N, ld_ctx) with D valid columns, or NULL for self-attention (context = normalised x, :404).
 *   mask  (b, N) bytes, 0 = masked out (sim <- -FLT_MAX, :411-415), or NULL.
 *   stats optional (b, heads, L, 2): per row {max of scaled logits in log2 units, sum of 2^(s-max)} (left untouched for a
 *         one-token context without a mask: every probability is 1 and nothing reads it there);
 *         with it hn_attn_probs re-creates Attention.attn_weights (:420) on demand.
 * The kernels pick the rank-D reassociated path when ld_ctx is a hn_context_pitch() pitch of 16/32
 * and D < dim_head, and the explicit K/V path otherwise. */
int hn_attn_fwd(const hn_attn_params *p, const float *x_in, float *x_out, int residual,
                const float *ctx, int ld_ctx, int b, int L, int N, int D, const uint8_t *mask,
                float *stats, void *workspace, size_t workspace_bytes, void *stream);
size_t hn_attn_workspace_bytes(const hn_attn_params *p, int has_ctx, int ld_ctx, int b, int L, int N, int D);

/* Context split over ranks (the optional second axis: b < #GPUs, one huge volume): every rank holds the latent
 * array and the parameters and attends to ITS N tokens only.
 *   hn_attn_partial_fwd: LayerNorm(x), projections and the attention core of :400-424 over the shard -> o_part (b*L, heads*dim_head),
 *     the NORMALISED attention output of the shard (head space: for the shared-context binding after the folded value
 *     projection), and stats (b, heads, L, 2) = {reference exponent M in log2 units, l = sum_t 2^(s_t - M)} of the shard.
 *     Inference only (dropout must be 0); a one-token shard without a mask is rejected (run such a modality whole).
 *   The caller gathers o_part / stats of all ranks (one all-gather of b*L*(inner + 2*heads) floats per rank: 270 KB per sample
 *   with the default model) into o_parts (n_parts, b*L, inner) and stats_parts (n_parts, b, heads, L, 2).
 *   hn_attn_merge_fwd: o = sum_r w_r o_r / sum_r w_r with w_r = 2^(M_r - max M) l_r (exactly the softmax over the union of the
 *     shards), then :425-426 + the residual: x_out = LeakyReLU(o W_out^T + b_out) [+ x_in].  stats (optional) receives the merged
 *     pair.  Parts are folded in index order: every rank computes bit-identical x_out. */
int hn_attn_partial_fwd(const hn_attn_params *p, const float *x_in, const float *ctx, int ld_ctx, int b, int L, int N, int D,
                        const uint8_t *mask, float *o_part, float *stats, void *workspace, size_t workspace_bytes,
                        void *stream);           /* workspace: hn_attn_workspace_bytes(p, 1, ld_ctx, b, L, N, D) */
size_t hn_attn_merge_workspace_bytes(const hn_attn_params *p, int b, int L);
int hn_attn_merge_fwd(const hn_attn_params *p, const float *x_in, float *x_out, int residual, const float *o_parts,
                      const float *stats_parts, int n_parts, int b, int L, float *stats, void *workspace,
                      size_t workspace_bytes, void *stream);

/* fourier_encode(x, max_freq, num_bands) (healnet/models/healnet.py:292-302) on a flat array of n positions:
 * out (n, 2*num_bands+1) = [sin(x s_f pi) .., cos(x s_f pi) .., x], s = linspace(1, max_freq/2, num_bands).
 * (hn_fourier_encode_concat is the fused form the model uses: positions are generated from the token index.) */
int hn_fourier_encode(const float *x, float *out, long n, int num_bands, float max_freq, void *stream);

/* The GELU / SELU gate modules (:323-331): out (rows, hidden) = a * act(g) with x (rows, 2*hidden) = [a | g].
 * Inside hn_ff_fwd the gate is the epilogue of the first GEMM. */
int hn_glu_gate(const float *x, float *out, long rows, int hidden, int gate, void *stream);

/* temperature_softmax(logits, temperature, dim=-1) (healnet/models/healnet.py:354-365) as a stand-alone op: softmax of
 * logits / temperature over the contiguous last dimension of a (rows, n) array.  Attention.forward calls it with
 * temperature = 0.5 (:419); inside hn_attn_fwd it is fused into the attention core. */
int hn_temperature_softmax(const float *logits, float *probs, long rows, int n, float temperature, void *stream);

/* Attention.attn_weights (:420), shape (b*heads, L, N), recomputed from x_in (the block INPUT) and stats. */
int hn_attn_probs(const hn_attn_params *p, const float *x_in, const float *ctx, int ld_ctx, int b, int L,
                  int N, int D, const uint8_t *mask, const float *stats, float *probs, void *workspace,
                  size_t workspace_bytes, void *stream);

/* Reduced export for the explainer: importance (b*heads, N) = mean over the L latent rows of
 * Attention.attn_weights -- what healnet/models/explainer.py:161-164 and :209-211 compute from the full matrix
 * (`torch.mean(w, dim=1)`) -- without materialising the (b*heads, L, N) tensor.  Same arguments as hn_attn_probs. */
int hn_attn_importance(const hn_attn_params *p, const float *x_in, const float *ctx, int ld_ctx, int b, int L,
                       int N, int D, const uint8_t *mask, const float *stats, float *importance, void *workspace,
                       size_t workspace_bytes, void *stream);

/* Training forward: identical to hn_attn_fwd but also keeps what hn_attn_bwd needs: `stats` (required) and `saved`
 * (hn_attn_saved_floats() floats: the normalised attention output O for the explicit binding, the normalised
 * context average P z for the rank-D binding, V for a one-token context). */
size_t hn_attn_saved_floats(const hn_attn_params *p, int has_ctx, int ld_ctx, int b, int L, int N, int D, int masked);
int hn_attn_fwd_train(const hn_attn_params *p, const float *x_in, float *x_out, int residual, const float *ctx,
                      int ld_ctx, int b, int L, int N, int D, const uint8_t *mask, float *stats, float *saved,
                      void *workspace, size_t workspace_bytes, void *stream);

/* Backward of the block (autograd of :313-321 + :400-426).  x_out is the block OUTPUT of the forward (used for the sign
 * of the LeakyReLU pre-activation), dy the gradient w.r.t. it; dx receives the gradient w.r.t. x_in (may alias dy).
 * No gradient flows to the context.  Parameter gradients are ACCUMULATED (+=) into the non-NULL entries. */
typedef struct hn_attn_grads {
  float *norm_w, *norm_b, *ctx_gamma, *ctx_beta, *w_q, *w_kv, *w_out, *b_out;
} hn_attn_grads;
int hn_attn_bwd(const hn_attn_params *p, const float *x_in, const float *x_out, int residual, const float *ctx,
                int ld_ctx, int b, int L, int N, int D, const uint8_t *mask, const float *stats, const float *saved,
                const float *dy, float *dx, const hn_attn_grads *grads, void *workspace, size_t workspace_bytes,
                void *stream);
size_t hn_attn_bwd_workspace_bytes(const hn_attn_params *p, int has_ctx, int ld_ctx, int b, int L, int N, int D, int masked);

/* ---------------------------------------------------------------------------------------------
 * Gated feed-forward block                             replaces PreNorm :313-321 + FeedForward :339-351
 * x_out = ( a * gate(g) ) W2^T + b2 [+ x_in],  [a|g] = LN(x_in) W1^T + b1
 * ------------------------------------------------------------------------------------------- */
typedef struct hn_ff_params {
  int dim;             /* l_d; hidden = 4*dim, W1 (8*dim, dim), W2 (dim, 4*dim)                  */
  int gate;            /* hn_gate                                                               */
  const float *norm_w; /* NULL -> no LayerNorm (bare FeedForward)                               */
  const float *norm_b;
  const float *w1, *b1, *w2, *b2;
  float dropout;       /* nn.Dropout on the block output before the residual (:347); 0 = off                    */
  hn_rng rng;
  int dim_valid;       /* staged layout (0 = off): LayerNorm statistics over the first dim_valid columns                 */
} hn_ff_params;

int hn_ff_fwd(const hn_ff_params *p, const float *x_in, float *x_out, int residual, int rows,
              void *workspace, size_t workspace_bytes, void *stream);
size_t hn_ff_workspace_bytes(const hn_ff_params *p, int rows);

/* Backward of the block (autograd of the same reference lines).  dy = gradient w.r.t. the block output; dx receives the
 * gradient w.r.t. x_in (dx may alias dy); with residual != 0 the skip connection's dy is included.  Parameter
 * gradients are ACCUMULATED (+=) into the non-NULL entries of hn_ff_grads (shapes of the parameters). */
typedef struct hn_ff_grads {
  float *norm_w, *norm_b, *w1, *b1, *w2, *b2;
} hn_ff_grads;
int hn_ff_bwd(const hn_ff_params *p, const float *x_in, const float *dy, float *dx, int residual, int rows,
              const hn_ff_grads *grads, void *workspace, size_t workspace_bytes, void *stream);
size_t hn_ff_bwd_workspace_bytes(const hn_ff_params *p, int rows);

/* ---------------------------------------------------------------------------------------------
 * Latent block                                         replaces the latent self block of one fusion iteration, :241-245:
 *                                                      x_mid = self_attn(x) + x ;  x_out = self_ff(x_mid) + x_mid
 * `attn` without a context (latent self-attention), `ff` of the same width.  With l_d = 128, l_c % 16 == 0, dim_head in
 * {16, 32, 64, 128} and heads * dim_head a multiple of 128 (<= 512) the block runs as three launches: one pass of the fused
 * latent chain (LayerNorm + Q / K / V projections), the attention core, one pass of the chain (out-projection + LeakyReLU +
 * residual + LayerNorm + gated feed-forward + residual, every intermediate in LDS); other shapes run hn_attn_fwd + hn_ff_fwd.
 * The same chain kernel carries the latent side of hn_fusion_forward (there it also absorbs the neighbouring cross blocks'
 * out-projection / query projection).
 *   x_mid, stats, saved: NULL for inference.  The training form (saved != NULL, then x_mid (b, L, d) and stats (b, heads, L, 2)
 *   are required; saved has hn_attn_saved_floats(attn, 0, 0, b, L, L, d, 0) floats) keeps what hn_latent_block_bwd needs
 *   and applies the blocks' dropout.  stats alone (inference) serves hn_attn_probs.
 * The backward is the two block backwards in reverse order (hn_ff_bwd, then hn_attn_bwd): the dQ / dK / dV cores and the
 * weight-gradient GEMMs dominate there, a fused chain would save launches only.
 * ------------------------------------------------------------------------------------------- */
int hn_latent_block_fwd(const hn_attn_params *attn, const hn_ff_params *ff, const float *x_in, float *x_out, int b, int L,
                        float *x_mid, float *stats, float *saved, void *workspace, size_t workspace_bytes, void *stream);
size_t hn_latent_block_workspace_bytes(const hn_attn_params *attn, const hn_ff_params *ff, int b, int L);
int hn_latent_block_bwd(const hn_attn_params *attn, const hn_ff_params *ff, const float *x_in, const float *x_mid, int b, int L,
                        const float *stats, const float *saved, const float *dy, float *dx, const hn_attn_grads *attn_grads,
                        const hn_ff_grads *ff_grads, void *workspace, size_t workspace_bytes, void *stream);
size_t hn_latent_block_bwd_workspace_bytes(const hn_attn_params *attn, const hn_ff_params *ff, int b, int L);

/* ---------------------------------------------------------------------------------------------
 * Head                                                 replaces to_logits :181-185
 * logits = LN(mean_n x) W^T + bias ;  x (b, L, d) -> (b, out_dims)
 * ------------------------------------------------------------------------------------------- */
int hn_head_fwd(const float *x, int b, int L, int d, const float *norm_w, const float *norm_b,
                const float *w, const float *bias, int out_dims, float *logits, void *stream);

/* Backward of the head: dlogits (b, out_dims) -> dx (b, L, d); parameter gradients accumulated into non-NULL pointers. */
int hn_head_bwd(const float *x, int b, int L, int d, const float *norm_w, const float *norm_b, const float *w,
                int out_dims, const float *dlogits, float *dx, float *d_norm_w, float *d_norm_b, float *d_w,
                float *d_bias, void *workspace, size_t workspace_bytes, void *stream);
size_t hn_head_bwd_workspace_bytes(int b, int d, int out_dims);

/* ---------------------------------------------------------------------------------------------
 * Whole fusion forward                                 replaces HealNet.forward :190-250
 * ------------------------------------------------------------------------------------------- */
typedef struct hn_modality_input {
  const void *data;          /* (b, S_1..S_a, C) or NULL for a missing modality (Appendix B-1)   */
  int spatial[HN_MAX_AXES];  /* S_1..S_a (a = num_spatial_axes[m])                               */
  int dtype;                 /* hn_dtype of data: HN_F32 (0), HN_BF16 or HN_U8                   */
} hn_modality_input;

typedef struct hn_model {
  int n_modalities, depth, l_c, l_d;
  int self_per_cross_attn;     /* 0 or 1 (>= 2 fails in the reference, :242)                     */
  int final_classifier_head;   /* 0 -> forward returns the latent array                          */
  int out_dims;
  int num_freq_bands;
  float max_freq;
  int fourier_encode_data;
  const int *channel_dims;       /* [M] */
  const int *num_spatial_axes;   /* [M] */
  const float *latents;          /* (l_c, l_d) */
  const hn_attn_params *cross_attn; /* [depth * M], index layer*M + m                           */
  const hn_ff_params *cross_ff;     /* [depth * M]                                              */
  const hn_attn_params *self_attn;  /* [depth]  (unused when self_per_cross_attn == 0)           */
  const hn_ff_params *self_ff;      /* [depth]                                                  */
  const float *head_norm_w, *head_norm_b, *head_w, *head_b;
  int core_precision;            /* hn_core_precision (inference forward only)                  */
  hn_rng rng;                    /* dropout generator state of hn_fusion_forward_train / _backward (stream unused) */
  int l_d_valid;                 /* staged layout (0 = off): the head's LayerNorm covers the first l_d_valid columns  */
} hn_model;

/* Optional timing hooks: when non-NULL, hn_fusion_forward records ev_start[i] / ev_stop[i]
 * (hipEvent_t, created by the caller) on `stream` around the i-th launch of the dominant kernel
 * (the split-KV attention kernel of each cross-attention block whose context has the most tokens),
 * for i < n_events.  n_recorded returns how many pairs were recorded. */
typedef struct hn_profile {
  void **ev_start;
  void **ev_stop;
  int n_events;
  int n_recorded;
} hn_profile;

/* Measurement hook for kernels launched from inside other entry points (the training forward / backward): while a table is set,
 * every launch of a kernel class named in it is bracketed by the next unused hipEvent pair of that entry, recorded on the launch
 * stream (n_recorded counts the launches seen, also beyond n_events).  Classes: "gemm_nt_glds" (patch-bag K/V projection),
 * "gemm_tn_glds" (patch-bag weight gradient G = dKV^T z).  The table and the event arrays belong to the caller and must stay valid
 * until hn_set_kernel_timers(NULL, 0) clears them.  This is the ONE process-wide registration of the library (a training step's
 * backward is launched from the autograd engine's thread, not the caller's, so the table cannot be thread-local); it is safe beside
 * concurrent callers: (table, n) is published as a consistent pair, an entry brackets launches on ITS stream only (`stream`; NULL =
 * any stream), so threads working on other streams are neither timed nor consume event pairs, and slots are claimed atomically.
 * Every other entry point of this header is re-entrant: no state outlives a call except this table and the per-device cluster
 * status word below (created once per device under a lock, written by kernels only when a bounded wait expires).  bench.py's
 * train_step.roofline is read through this hook.  (ABI v12: the `stream` field.) */
typedef struct hn_kernel_timer {
  const char *kernel;
  void **ev_start;
  void **ev_stop;
  int n_events;
  int n_recorded;
  void *stream;                   /* hipStream_t: only launches on this stream are bracketed; NULL: launches on any stream */
} hn_kernel_timer;
int hn_set_kernel_timers(hn_kernel_timer *timers, int n);

/* ---- Cluster mode of the latent chains and its failure signal (ABI v10) ------------------------------------------------------
 * Small batches (b * l_c / 16 <= 128 row tiles) run the latent chains of hn_fusion_forward / _forward_train / _backward as
 * CLUSTERS: 2 or 4 workgroups share a row tile and exchange partial sums through the L2, each spinning on the others' flags.
 * HIP does not promise that all workgroups of a grid are resident at once.  The grid's dispatch order keeps the members of a
 * tile adjacent, so a partly occupied chip (an RCCL kernel on another stream, a CU mask) only slows the launch down; should a
 * member still not show up within the wait bound (default 100 ms, HN_CLUSTER_TIMEOUT_US / hn_cluster_config), the tile's rows
 * become NaN -- never a silently incomplete sum -- and the launch stores its token into a host-mapped status word of the device
 * (64 bytes of pinned memory, the only allocation this library ever makes; created by the first eager cluster launch or
 * hn_cluster_status call on the device).  From then on:
 *   - the NEXT call of hn_fusion_forward / hn_fusion_forward_cp / hn_fusion_forward_train / hn_fusion_backward / hn_l1_adam_step
 *     on that device launches nothing, switches cluster mode off for the device (sticky), waits for the device to drain
 *     (hipDeviceSynchronize: work enqueued before this call still sees the word set), clears the word and returns
 *     HN_E_CORESIDENCY (hn_last_error_string names the launch).  On a capturing stream nothing can be waited for: the word stays
 *     set, and every entry point keeps returning HN_E_CORESIDENCY, until the caller has drained the device and acknowledged
 *     through hn_cluster_status.  Outputs produced since the lost exchange may hold NaN: repeat
 *     the step (all later launches run the same arithmetic without clusters);
 *   - an hn_l1_adam_step that was ALREADY enqueued when the exchange was lost reads the word on the device and leaves
 *     parameters and moments untouched, so a poisoned gradient never reaches the weights of this process;
 *   - a caller that replays captured graphs (no entry point runs) polls hn_cluster_status itself.
 * hn_cluster_status(device, acknowledge, info): query; with acknowledge != 0 a pending loss is consumed exactly as the entry
 * points do (mode off, word cleared).  hn_cluster_config(device, enable, timeout_us): enable 0 / 1 switches the mode off / back on
 * (-1: keep; 2: on WITH FAULT INJECTION -- the last member of every tile withholds its flag, so every cluster launch loses an
 * exchange after the wait bound: the hook the tests drive this protocol with); timeout_us > 0 sets the wait bound, 0 restores
 * the default (-1: keep).  HN_NO_CHAIN_CLUSTER=1 in the environment
 * disables the mode for the process. */
typedef struct hn_cluster_info {
  int pending;                    /* a lost exchange has been reported by a kernel and not yet consumed              */
  int enabled;                    /* cluster mode currently allowed on the device                                      */
  unsigned lost;                  /* losses consumed so far (entry points + acknowledging status calls)                */
  unsigned last_token;            /* token of the most recent reporting launch (0: none)                               */
  int timeout_us;                 /* wait bound in force                                                               */
  const volatile unsigned *status_word;   /* host address of the mapped word (0 = clean), or NULL before its creation  */
} hn_cluster_info;
int hn_cluster_status(int device, int acknowledge, hn_cluster_info *info);
int hn_cluster_config(int device, int enable, int timeout_us);

/* out: (b, out_dims) logits, or (b, l_c, l_d) when return_embeddings != 0 or the model has no head.
 * skip_self_on_missing reproduces the reference's verbose=True quirk (:229-232): bit i set -> when modality i is
 * missing, that iteration's latent self block is skipped as well (the `continue` under `if verbose`).  The reference
 * only takes that branch for a None entry INSIDE the tensor list; a modality beyond a shorter list fails in the bare
 * try/except (:238) and still runs the self block, so a host sets bit i only for the former (0 = never, -1 = all).
 * attn_stats: optional array of depth*(M+1) pointers (layer-major: cross_0..cross_{M-1}, self), each
 *             NULL or a (b, heads, l_c, 2) buffer receiving that block's softmax statistics.
 * x_trace:    optional; receives the latent array (b, l_c, l_d) fed INTO every attention block
 *             in the same depth*(M+1) order (needed by hn_attn_probs); NULL entries are skipped. */
int hn_fusion_forward(const hn_model *model, const hn_modality_input *inputs, int b, const uint8_t *mask,
                      int skip_self_on_missing, int return_embeddings, float *out, float **attn_stats,
                      float **x_trace, void *workspace, size_t workspace_bytes, void *stream,
                      hn_profile *profile);
size_t hn_fusion_workspace_bytes(const hn_model *model, const hn_modality_input *inputs, int b);

/* Context split: TRAINING, block level (ABI v11).  With the GLOBAL softmax statistics (M, l) and the global attention output of a
 * split cross block -- both replicated once the shards' pairs are merged -- the ordinary backward of the block run on a rank's slab
 * yields exactly that slab's contributions: every gradient that passes through the core backward (dQ and with it dW_q, the query
 * LayerNorm's gradients and dx; dK / dV and with them dW_kv and the context LayerNorm's gradients) is a PARTIAL sum over the slab's
 * tokens, every other one (dW_out, db_out; for the shared-context binding also dW_v and the value side of the context LayerNorm
 * affine) is computed from replicated quantities.  So:
 *   forward   hn_attn_fwd_train on the slab (residual = 0; x_out is scratch) -> stats, saved of the shard;
 *             all-gather the first b * L * heads * hn_attn_saved_part_width() floats of `saved` (the shard's normalised attention
 *             output O, or P z for the shared-context binding) and the statistics: b * L * (heads * width + 2 heads) floats per rank;
 *             hn_attn_merge_parts folds the shards in index order into the same place of `saved` and into `stats` (the tail of
 *             `saved`, the slab's projected K / V, stays the rank's own);  hn_attn_finish_fwd computes the block's output from the
 *             merged `saved`: x_out = LeakyReLU(O W_out^T + b_out) [+ x_in], bit-identical on every rank;
 *   backward  hn_attn_bwd_cp(dy; x_out = the block's output WITH the residual, as hn_attn_finish_fwd(residual = 1) wrote it)
 *             -> dx = the partial gradient through the queries (no residual term) and += the parameter
 *             gradients, the replicated ones only where replicated_owner != 0 (pass it on ONE rank); sum dx and every parameter
 *             gradient of the block over the ranks (one all-reduce), then add dy for the residual.
 * No dropout, no mask; every rank needs at least two tokens.  healnet_amd.dist.context_parallel_forward drives this when gradients
 * are enabled (tests/test_gpu_context_split.py: 2 and 3 ranks against the plain backward and oracle autograd). */
int hn_attn_saved_part_width(const hn_attn_params *p, int ld_ctx, int b, int L, int N, int D);   /* per head; 0: not mergeable */
int hn_attn_merge_parts(const float *o_parts, const float *stats_parts, int n_parts, long o_stride, long stats_stride, int b,
                        int heads, int L, int width, float *o, float *stats, void *stream);
int hn_attn_finish_fwd(const hn_attn_params *p, const float *x_in, float *x_out, int residual, int ld_ctx, int b, int L, int N,
                       int D, const float *saved, void *workspace, size_t workspace_bytes, void *stream);   /* hn_attn_workspace_bytes(p, 1, ...) */
int hn_attn_bwd_cp(const hn_attn_params *p, const float *x_in, const float *x_out, const float *ctx, int ld_ctx, int b, int L, int N,
                   int D, const float *stats, const float *saved, const float *dy, float *dx, const hn_attn_grads *grads,
                   int replicated_owner, void *workspace, size_t workspace_bytes, void *stream);   /* hn_attn_bwd_workspace_bytes(p, 1, ..., 0) */

/* The fused forward with the CONTEXT of some modalities split over ranks (SURVEY.md 8(e) second axis; ABI v9) -- the fused form of
 * hn_encode_norm_slab + hn_attn_partial_fwd + hn_attn_merge_fwd: inputs[i] of a modality in `split_mask` is THIS rank's slab (rows
 * [axis0_begin[i], axis0_begin[i] + spatial[0]) of an axis of axis0_total[i] positions); behind the attention core of each of its
 * cross blocks the entry point writes the rank's (normalised output | statistics) to `local`, calls `exchange(user, floats, stream)`
 * -- the caller enqueues ONE all-gather of `floats` floats per rank from `local` into `parts` (rank-major, stride `floats`) on
 * `stream` -- and folds the parts in rank order; out-projection, feed-forward and the next projections run on the latent chains as
 * in hn_fusion_forward.  Everything else is replicated.  fp32 core, no mask, no missing modality, default (unstaged) shapes; a
 * model the chains do not take is refused (HN_E_UNSUPPORTED: use the block-level entry points).  `local` / `parts` hold
 * hn_context_split_floats() / n_parts times that many floats.  Workspace: hn_fusion_workspace_bytes() of the slab inputs.
 * ABI v10: the callback returns 0, or non-zero when the collective could not be enqueued -- the entry point then stops at once
 * and returns HN_E_HIP (it used to carry on calling `exchange` for the later blocks while the peers sat in the failed one). */
typedef int (*hn_cp_exchange_fn)(void *user, int floats, void *stream);
typedef struct hn_context_split {
  int n_parts;
  unsigned split_mask;
  int axis0_begin[16], axis0_total[16];
  float *local, *parts;
  hn_cp_exchange_fn exchange;
  void *user;
} hn_context_split;
size_t hn_context_split_floats(const hn_model *model, int b);
int hn_fusion_forward_cp(const hn_model *model, const hn_modality_input *inputs, int b, int return_embeddings,
                         const hn_context_split *cp, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* Staged models.  The fused latent side (chain.hip / bchain.hip) is built for l_d = 128, head widths of 16 / 32 / 64 / 128, inner
 * widths that are multiples of 128 and 16-row tiles.  A model outside those shapes that fits them after ZERO PADDING -- l_d <= 128,
 * every head width <= 128 with heads * padded width <= 512 (the reference's tuned TCGA shapes: config/best_hyperparams.yml) -- is
 * run by hn_fusion_forward / _forward_train / _backward as its padded image: one launch per forward copies the latent-side weights
 * into a zero-padded shadow inside the workspace (a training forward: on the tape, for its backward), the shadow model runs the
 * fast path with its LayerNorms over the valid width and the softmax scale of the valid head width, and embeddings / trace
 * slots / gradients come back in the model's own shapes (the padded gradients are accumulated onto the real ones at the end of the
 * backward; gradient-readiness signals then fire after that launch).  Exact: pad entries are zero and stay zero.  The
 * workspace / tape size queries account for it.  hn_fusion_is_staged tells which route a descriptor takes (1 / 0).
 * A host that keeps padded weights itself can skip the per-forward copy by passing the padded descriptor directly: l_d = 128 with
 * hn_model.l_d_valid, hn_attn_params.dim_head_valid / query_dim_valid, hn_ff_params.dim_valid set (layout: see those fields;
 * w1 rows [0, 4 l_d) -> [0, 512) and [4 l_d, 8 l_d) -> [512, 1024), every buffer of b * l_c rows allocated for the count rounded up
 * to 16).  HN_NO_STAGING=1 (environment, development): the generic per-block route instead. */
int hn_fusion_is_staged(const hn_model *model);

/* ---------------------------------------------------------------------------------------------
 * Training: forward that records a tape, and the matching backward   (autograd of HealNet.forward :190-250, driven by
 * surv_loss.backward() at healnet/main.py:464).  The tape holds the latent array before every executed block, the
 * softmax statistics and the small per-attention tensors of hn_attn_fwd_train.  hn_fusion_backward ACCUMULATES (+=)
 * into every non-NULL gradient pointer of hn_model_grads (same shapes / indexing as hn_model; tied weights simply point
 * several entries at one buffer); the modality inputs receive no gradient.
 * ------------------------------------------------------------------------------------------- */
typedef struct hn_model_grads {
  float *latents;
  const hn_attn_grads *cross_attn; /* [depth * M] or NULL */
  const hn_ff_grads *cross_ff;     /* [depth * M] or NULL */
  const hn_attn_grads *self_attn;  /* [depth] or NULL */
  const hn_ff_grads *self_ff;      /* [depth] or NULL */
  float *head_norm_w, *head_norm_b, *head_w, *head_b;
} hn_model_grads;

size_t hn_fusion_tape_bytes(const hn_model *model, const hn_modality_input *inputs, int b, int masked, int skip_self_on_missing);
/* Where the tape of hn_fusion_forward_train keeps what hn_attn_probs needs, as FLOAT offsets from the tape base, one
 * entry per attention slot (layer * (M + 1) + modality, M = the layer's latent self-attention; (size_t)-1: block not
 * executed): stats_off -> the (b, heads, l_c, 2) softmax statistics, x_off -> the (b, l_c, l_d) latent array the block
 * read.  Lets a host expose Attention.attn_weights (healnet.py:420) after a training forward without copying either
 * out (pass attn_stats = x_trace = NULL to hn_fusion_forward_train). */
int hn_fusion_tape_layout(const hn_model *model, const hn_modality_input *inputs, int b, int masked, int skip_self_on_missing,
                          size_t *stats_off, size_t *x_off);
int hn_fusion_forward_train(const hn_model *model, const hn_modality_input *inputs, int b, const uint8_t *mask,
                            int skip_self_on_missing, int return_embeddings, float *out, float **attn_stats,
                            float **x_trace, void *tape, size_t tape_bytes, void *workspace, size_t workspace_bytes,
                            void *stream);
size_t hn_fusion_backward_workspace_bytes(const hn_model *model, const hn_modality_input *inputs, int b, int masked);
/* Gradient-readiness signals of hn_fusion_backward (the data-parallel gradient exchange overlapped with the
 * backward; healnet/main.py:464-465 is the call site it serves).  Index depth = the head's parameter gradients are final;
 * index l < depth = every block of layers >= l has run its backward, so nothing accumulates any more into the gradients of
 * parameters used only by layers >= l (layers finish in reverse order; with weight tying the shared blocks belong to
 * layer 1).  For each index, in that order while the backward is being enqueued:
 *   events[idx] (hipEvent_t created by the caller, NULL entries skipped) is recorded on `stream`, then
 *   notify(idx, user) is called on the calling host thread (may enqueue work on other streams, e.g. an all-reduce that
 *   waits on the event; must not synchronise `stream`).
 * The latent array's gradient is final when the call's last kernel has run (stream order). */
typedef struct hn_grad_ready {
  void **events;                          /* [depth + 1] or NULL */
  void (*notify)(int index, void *user);  /* or NULL */
  void *user;
} hn_grad_ready;

int hn_fusion_backward(const hn_model *model, const hn_modality_input *inputs, int b, const uint8_t *mask,
                       int skip_self_on_missing, int return_embeddings, const float *dout, const void *tape,
                       const hn_model_grads *grads, void *workspace, size_t workspace_bytes, void *stream,
                       const hn_grad_ready *ready);

/* ---------------------------------------------------------------------------------------------
 * Training-step tail                  replaces healnet/main.py:439-447 + :464-467
 * ------------------------------------------------------------------------------------------- */
/* hazards = sigmoid(logits), survival = cumprod(1 - hazards), risk = -sum(survival) (main.py:439-441), the discrete
 * survival NLL `nll_loss(hazards, S, Y, c, weights, alpha, eps)` (healnet/models/survival_loss.py:9-43; batch mean) and
 * grad_scale * d loss / d logits in one launch.
 *   logits (b, n_bins); y (b) int64 bin index in [0, n_bins) -- a label outside that range (torch.gather raises in the
 *   reference) turns the loss and that sample's dlogits row into NaN, nothing is read out of bounds;
 *   censorship (b) float {0, 1}; class_weights (n_bins) or NULL;
 *   loss (1); dlogits (b, n_bins) or NULL; hazards / survival (b, n_bins) or NULL; risk (b) or NULL. */
int hn_surv_nll(const float *logits, const int64_t *y, const float *censorship, const float *class_weights, int b,
                int n_bins, float alpha, float eps, float grad_scale, float *loss, float *dlogits, float *hazards,
                float *survival, float *risk, void *stream);

/* One fused pass over flat fp32 buffers of n elements: g = grad_scale * grads + l1 * sign(params) (the gradient of
 * calc_reg_loss, healnet/utils/train_utils.py:5-14), then torch.optim.Adam's update (main.py:390; step >= 1 is the
 * 1-based update count; lr / beta1 are per-call because OneCycleLR cycles both, main.py:391-394), and
 * reg_loss[0] = l1 * sum |params| of the parameters BEFORE the update (the value main.py:449 logs); reg_loss may be
 * NULL.  Buffers 16-byte aligned; workspace >= hn_l1_adam_workspace_bytes(). */
int hn_l1_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, long n, double l1,
                    double grad_scale, double lr, double beta1, double beta2, double eps, int step, float *reg_loss,
                    void *workspace, size_t workspace_bytes, void *stream);
size_t hn_l1_adam_workspace_bytes(void);

/* The dropout mask of one block as bytes (1 = kept): (rows, cols) = (b*heads*L, N) for an attention block,
 * (b*L, dim) for a feed-forward block (pass is_ff != 0, which sets bit 31 of rng.stream).  Test / debugging aid. */
int hn_dropout_mask(float p, hn_rng rng, int is_ff, long rows, int cols, uint8_t *mask, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HEALNET_HIP_H */
