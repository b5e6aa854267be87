"""CPU oracle for the HEALNet fusion hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is an op-for-op restatement (plain PyTorch fp32 on the host, unfused,
materialised score matrix) of the algorithm in the reference's
``healnet/models/healnet.py``.  It exists only so that

  * ``tests/``                     can check the HIP path against it,
  * ``__graft_entry__.smoke()``    can check one small invocation against it,
  * ``bench.py``'s ``cpu_baseline`` leg can time it on the GPU box's host cores.

Nothing under ``healnet_amd/`` imports it; the product path never routes through it.

Parity pin: PINNED.  ``tools/gen_goldens.py`` (run in the build container, where
``/root/reference`` exists) imports the reference by file path, checks this
restatement against it to fp32 noise and writes the fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks the restatement
against those committed fixtures on every run (CPU and GPU box alike).

The oracle is *functional*: every routine takes tensors / a ``state_dict`` with the
reference's key layout rather than building modules, so it shares no
code with the product's module classes.

Reference map (all lines in /root/reference/healnet/models/healnet.py):
  fourier_features        <- fourier_encode                 :292-302
  encode_modality         <- HealNet.forward preprocessing  :200-222
  layer_norm              <- nn.LayerNorm in PreNorm        :306-321, :183
  attention               <- Attention.forward              :400-426 (+ temperature_softmax :354-365)
  feed_forward            <- FeedForward / SELU / GELU      :323-351
  fusion_forward          <- HealNet.forward fusion loop    :225-250
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration mirror of HealNet.__init__ keyword arguments (healnet.py:15-38)
# --------------------------------------------------------------------------------------
@dataclass
class FusionConfig:
    n_modalities: int
    channel_dims: Sequence[int]
    num_spatial_axes: Sequence[int]
    out_dims: int
    depth: int = 3
    num_freq_bands: int = 2
    max_freq: float = 10.0
    l_c: int = 128
    l_d: int = 128
    x_heads: int = 8
    l_heads: int = 8
    cross_dim_head: int = 64
    latent_dim_head: int = 64
    attn_dropout: float = 0.0
    ff_dropout: float = 0.0
    weight_tie_layers: bool = False
    fourier_encode_data: bool = True
    self_per_cross_attn: int = 1
    final_classifier_head: bool = True
    snn: bool = True

    def context_dim(self, m: int) -> int:
        # healnet.py:133-139: D_m = C_m + axes_m * (2F + 1) when fourier encoding is on
        extra = self.num_spatial_axes[m] * (2 * self.num_freq_bands + 1) if self.fourier_encode_data else 0
        return int(self.channel_dims[m]) + extra


# --------------------------------------------------------------------------------------
# a2/a3: positional encoding
# --------------------------------------------------------------------------------------
def fourier_features(pos: Tensor, max_freq: float, num_bands: int) -> Tensor:
    """healnet.py:292-302.  pos (..., ) -> (..., 2*num_bands + 1) = [sin bands | cos bands | pos]."""
    p = pos.unsqueeze(-1)
    sigma = torch.linspace(1.0, max_freq / 2, num_bands, dtype=pos.dtype)          # :296
    arg = p * sigma * math.pi                                                      # :299  (p*sigma)*pi
    return torch.cat([arg.sin(), arg.cos(), p], dim=-1)                            # :300-301


def encode_modality(data: Tensor, num_bands: int, max_freq: float, fourier: bool = True) -> Tensor:
    """healnet.py:204-222.  (b, *S, C) -> (b, prod(S), C + axes*(2F+1)); data channels first."""
    b = data.shape[0]
    spatial = list(data.shape[1:-1])
    if fourier:
        lines = [torch.linspace(-1.0, 1.0, steps=s, dtype=data.dtype) for s in spatial]      # :212
        grid = torch.stack(torch.meshgrid(*lines, indexing="ij"), dim=-1)                     # :213  (*S, axes)
        enc = fourier_features(grid, max_freq, num_bands)                                     # (*S, axes, 2F+1)
        enc = enc.reshape(*spatial, -1)                                                       # :215 axis-major
        enc = enc.unsqueeze(0).expand(b, *enc.shape)                                          # :216
        data = torch.cat([data, enc], dim=-1)                                                 # :217
    return data.reshape(b, -1, data.shape[-1])                                                # :221 row-major


# --------------------------------------------------------------------------------------
# a4: LayerNorm (biased variance, eps 1e-5, affine)
# --------------------------------------------------------------------------------------
def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


# --------------------------------------------------------------------------------------
# a6/a7: attention with temperature softmax
# --------------------------------------------------------------------------------------
def attention(x: Tensor, context: Optional[Tensor], w_q: Tensor, w_kv: Tensor, w_out: Tensor, b_out: Tensor,
              heads: int, mask: Optional[Tensor] = None, temperature: float = 0.5,
              return_weights: bool = False, drop_mult: Optional[Tensor] = None):
    """healnet.py:400-426.  x (b, L, dq) already normalised; context (b, N, D) already normalised
    (or None -> x).  Returns LeakyReLU_{0.01}(concat_heads(P V) W_out^T + b_out) and optionally P
    with shape (b*heads, L, N) (batch-major head index b*heads + h, :407)."""
    ctx = x if context is None else context                                         # :404
    b, L, _ = x.shape
    n = ctx.shape[1]
    inner = w_q.shape[0]
    e = inner // heads
    q = x @ w_q.t()                                                                 # :403
    kv = ctx @ w_kv.t()                                                             # :405
    k, v = kv[..., :inner], kv[..., inner:]                                         # chunk(2): first half = K

    def split(t: Tensor, rows: int) -> Tensor:                                      # :407 'b n (h d) -> (b h) n d'
        return t.reshape(b, rows, heads, e).permute(0, 2, 1, 3).reshape(b * heads, rows, e)

    qh, kh, vh = split(q, L), split(k, n), split(v, n)
    sim = torch.bmm(qh, kh.transpose(1, 2)) * (e ** -0.5)                           # :409
    if mask is not None:                                                            # :411-415
        flat = mask.reshape(b, -1)
        neg = -torch.finfo(sim.dtype).max
        keep = flat[:, None, None, :].expand(b, heads, 1, n).reshape(b * heads, 1, n)
        sim = sim.masked_fill(~keep, neg)
    attn = torch.softmax(sim / temperature, dim=-1)                                 # :419 / :364-365
    pv = attn if drop_mult is None else attn * drop_mult                            # :421 nn.Dropout with a GIVEN mask:
    out = torch.bmm(pv, vh)                                                         # :424   drop_mult = keep / (1 - p)
    out = out.reshape(b, heads, L, e).permute(0, 2, 1, 3).reshape(b, L, inner)      # :425
    out = F.leaky_relu(out @ w_out.t() + b_out, negative_slope=1e-2)                # :383-386, :426
    return (out, attn) if return_weights else out


# --------------------------------------------------------------------------------------
# a8: gated feed-forward
# --------------------------------------------------------------------------------------
def feed_forward(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, snn: bool = True,
                 drop_mult: Optional[Tensor] = None) -> Tensor:
    """healnet.py:339-351 with the SELU (:328-331) or GELU (:323-326) gate; x already normalised.
    drop_mult: the nn.Dropout of :347 with a given mask (keep / (1 - p), shape of the output)."""
    u = x @ w1.t() + b1
    half = u.shape[-1] // 2
    a, g = u[..., :half], u[..., half:]                                             # chunk(2): value first, gate second
    z = a * (F.selu(g) if snn else F.gelu(g))
    y = z @ w2.t() + b2
    return y if drop_mult is None else y * drop_mult


# --------------------------------------------------------------------------------------
# a9/a10: the fusion loop over a reference-layout state_dict
# --------------------------------------------------------------------------------------
def _cross_block(sd: Dict[str, Tensor], pfx: str, x: Tensor, ctx: Tensor, heads: int, mask, keep: Optional[list], dm=None):
    xn = layer_norm(x, sd[pfx + "norm.weight"], sd[pfx + "norm.bias"])                                  # :314
    cn = layer_norm(ctx, sd[pfx + "norm_context.weight"], sd[pfx + "norm_context.bias"])              # :316-319
    y, p = attention(xn, cn, sd[pfx + "fn.to_q.weight"], sd[pfx + "fn.to_kv.weight"],
                     sd[pfx + "fn.to_out.0.weight"], sd[pfx + "fn.to_out.0.bias"], heads, mask,
                     return_weights=True, drop_mult=dm)
    if keep is not None:
        keep.append(p)
    return y + x                                                                                         # :236 residual


def _self_block(sd: Dict[str, Tensor], pfx: str, x: Tensor, heads: int, keep: Optional[list], dm=None):
    xn = layer_norm(x, sd[pfx + "norm.weight"], sd[pfx + "norm.bias"])
    y, p = attention(xn, None, sd[pfx + "fn.to_q.weight"], sd[pfx + "fn.to_kv.weight"],
                     sd[pfx + "fn.to_out.0.weight"], sd[pfx + "fn.to_out.0.bias"], heads, None,
                     return_weights=True, drop_mult=dm)
    if keep is not None:
        keep.append(p)
    return y + x                                                                                         # :244


def _ff_block(sd: Dict[str, Tensor], pfx: str, x: Tensor, snn: bool, dm=None):
    xn = layer_norm(x, sd[pfx + "norm.weight"], sd[pfx + "norm.bias"])
    return feed_forward(xn, sd[pfx + "fn.net.0.weight"], sd[pfx + "fn.net.0.bias"],
                        sd[pfx + "fn.net.2.weight"], sd[pfx + "fn.net.2.bias"], snn, dm) + x             # :237/:245


@dataclass
class FusionTrace:
    contexts: List[Optional[Tensor]] = field(default_factory=list)   # encoded (b, N, D) per modality
    attn: List[Tensor] = field(default_factory=list)                 # every P in execution order
    attn_tags: List[tuple] = field(default_factory=list)             # (layer, "cross"|"self", modality) per P
    blocks: List[Tensor] = field(default_factory=list)               # x after every executed block


def fusion_forward(sd: Dict[str, Tensor], cfg: FusionConfig, tensors: Sequence[Optional[Tensor]],
                   mask: Optional[Tensor] = None, return_embeddings: bool = False, verbose: bool = False,
                   trace: Optional[FusionTrace] = None, drop: Optional[Dict[int, Tensor]] = None) -> Tensor:
    """healnet.py:190-250 including its observable quirks:

      * ``drop`` (training mode with dropout > 0): {index of the executed block, counted in execution order ->
        multiplier tensor keep / (1 - p)} -- attention blocks (b*heads, L, N) on the probabilities (:421), feed-forward
        blocks (b, L, d) on the block output (:347).  The masks are GIVEN (the tests export the build's own Philox masks),
        so this restates nn.Dropout exactly for any mask;
      * a ``None`` modality skips its cross-attention + cross-FF, but the latent self block of that
        (layer, modality) iteration still runs (:235-245); with ``verbose=True`` the ``continue`` at
        :232 skips the self block too;
      * a list shorter than ``n_modalities`` behaves like trailing ``None`` entries -- except under ``verbose=True``: the
        reference's ``missing_idx`` (:193) only holds the ``None`` entries INSIDE the list, so a modality beyond a shorter
        list fails in the bare try/except (:238) and still runs its latent self block (fixture g9_verbose_shortlist);
      * ``self_per_cross_attn == 0`` -> no latent blocks; ``>= 2`` is a ValueError in the reference.
    The bare ``except`` of :238 is NOT restated: real shape errors raise here.
    """
    M = cfg.n_modalities
    present = [i < len(tensors) and tensors[i] is not None for i in range(M)]
    ctxs: List[Optional[Tensor]] = []
    b = None
    for i in range(M):
        if not present[i]:
            ctxs.append(None)
            continue
        data = tensors[i]
        assert data.dim() - 2 == cfg.num_spatial_axes[i], "axis count mismatch"       # :206-208
        b = data.shape[0]
        ctxs.append(encode_modality(data, cfg.num_freq_bands, cfg.max_freq, cfg.fourier_encode_data))
    if b is None:
        raise ValueError("at least one modality must be present")
    if trace is not None:
        trace.contexts = ctxs
    keep = trace.attn if trace is not None else None
    tags = trace.attn_tags if trace is not None else []

    x = sd["latents"].unsqueeze(0).expand(b, -1, -1)                                  # :225
    if cfg.self_per_cross_attn not in (0, 1):
        raise ValueError("self_per_cross_attn >= 2 fails in the reference (healnet.py:242)")
    step = 0
    dm = (lambda k: None) if drop is None else (lambda k: drop.get(k))
    for layer in range(cfg.depth):
        for m in range(M):
            if not present[m] and verbose and m < len(tensors):                       # :229-232 (missing_idx, :193)
                continue
            if present[m]:
                x = _cross_block(sd, f"layers.{layer}.{2 * m}.", x, ctxs[m], cfg.x_heads, mask, keep, dm(step))
                step += 1
                tags.append((layer, "cross", m))
                if trace is not None:
                    trace.blocks.append(x)
                x = _ff_block(sd, f"layers.{layer}.{2 * m + 1}.", x, cfg.snn, dm(step))
                step += 1
                if trace is not None:
                    trace.blocks.append(x)
            if cfg.self_per_cross_attn > 0:                                           # :241-245
                x = _self_block(sd, f"layers.{layer}.{2 * M}.0.", x, cfg.l_heads, keep, dm(step))
                step += 1
                tags.append((layer, "self", m))
                if trace is not None:
                    trace.blocks.append(x)
                x = _ff_block(sd, f"layers.{layer}.{2 * M}.1.", x, cfg.snn, dm(step))
                step += 1
                if trace is not None:
                    trace.blocks.append(x)
    if return_embeddings or not cfg.final_classifier_head:                            # :247-250, :181-185
        return x
    pooled = x.mean(dim=1)
    pooled = layer_norm(pooled, sd["to_logits.1.weight"], sd["to_logits.1.bias"])
    return pooled @ sd["to_logits.2.weight"].t() + sd["to_logits.2.bias"]


# --------------------------------------------------------------------------------------
# deterministic closed-form fillers (shared by the golden generator and the tests so that
# default-size fixtures only need to store outputs)
# --------------------------------------------------------------------------------------
def _key_phase(key: str) -> float:
    h = 0
    for ch in key:
        h = (h * 131 + ord(ch)) % 1000003
    return (h % 6283) / 1000.0


def attention_weights_in_module_order(trace: FusionTrace, cfg: FusionConfig) -> List[Tensor]:
    """healnet.py:252-262: ``get_attention_weights`` walks ``self.modules()``, i.e. per layer
    [cross_0 .. cross_{M-1}, self]; each module holds the P of its LAST execution (the self block of a
    layer runs once per modality, :241-245).  Untied weights only."""
    last = {}
    for tag, p in zip(trace.attn_tags, trace.attn):
        layer, kind, m = tag
        last[(layer, kind, m if kind == "cross" else -1)] = p
    out = []
    for layer in range(cfg.depth):
        for m in range(cfg.n_modalities):
            if (layer, "cross", m) in last:
                out.append(last[(layer, "cross", m)])
        if (layer, "self", -1) in last:
            out.append(last[(layer, "self", -1)])
    return out


def filler_tensor(key: str, shape: Sequence[int], gain: float = 1.0) -> Tensor:
    """Closed-form pseudo-weights: sin(0.37*idx + phase(key)) scaled like a Linear init.
    LayerNorm weights are 1 + 0.25*sin, biases 0.1*sin, latents sin*1.0."""
    n = 1
    for s in shape:
        n *= int(s)
    idx = torch.arange(n, dtype=torch.float64)
    wave = torch.sin(0.37 * idx + _key_phase(key)) * torch.cos(0.011 * idx + 0.5 * _key_phase(key))
    leaf = key.split(".")[-1]
    is_norm = ".norm" in key or key.startswith("to_logits.1")
    if key == "latents":
        out = wave * 1.5
    elif is_norm and leaf == "weight":
        out = 1.0 + 0.25 * wave
    elif leaf == "bias":
        out = 0.1 * wave
    else:
        fan_in = int(shape[-1])
        out = gain * wave * (3.0 / fan_in) ** 0.5 * 1.4
    return out.reshape(*shape).to(torch.float32)


def filler_input(shape: Sequence[int], salt: int) -> Tensor:
    """Closed-form U[0,1)-like inputs: frac(idx * golden_ratio + salt*0.1234)."""
    n = 1
    for s in shape:
        n *= int(s)
    idx = torch.arange(n, dtype=torch.float64)
    v = torch.remainder(idx * 0.6180339887498949 + salt * 0.1234, 1.0)
    return v.reshape(*shape).to(torch.float32)


def state_dict_shapes(cfg: FusionConfig) -> Dict[str, tuple]:
    """Key -> shape for the reference's state_dict layout, untied weights."""
    d, M = cfg.l_d, cfg.n_modalities
    shapes: Dict[str, tuple] = {"latents": (cfg.l_c, d)}

    def attn(pfx, inner, ctx_dim, with_ctx_norm):
        shapes[pfx + "fn.to_q.weight"] = (inner, d)
        shapes[pfx + "fn.to_kv.weight"] = (2 * inner, ctx_dim)
        shapes[pfx + "fn.to_out.0.weight"] = (d, inner)
        shapes[pfx + "fn.to_out.0.bias"] = (d,)
        shapes[pfx + "norm.weight"] = (d,)
        shapes[pfx + "norm.bias"] = (d,)
        if with_ctx_norm:
            shapes[pfx + "norm_context.weight"] = (ctx_dim,)
            shapes[pfx + "norm_context.bias"] = (ctx_dim,)

    def ff(pfx):
        shapes[pfx + "fn.net.0.weight"] = (8 * d, d)
        shapes[pfx + "fn.net.0.bias"] = (8 * d,)
        shapes[pfx + "fn.net.2.weight"] = (d, 4 * d)
        shapes[pfx + "fn.net.2.bias"] = (d,)
        shapes[pfx + "norm.weight"] = (d,)
        shapes[pfx + "norm.bias"] = (d,)

    for L in range(cfg.depth):
        for m in range(M):
            attn(f"layers.{L}.{2 * m}.", cfg.x_heads * cfg.cross_dim_head, cfg.context_dim(m), True)
            ff(f"layers.{L}.{2 * m + 1}.")
        for k in range(cfg.self_per_cross_attn):
            attn(f"layers.{L}.{2 * M}.{2 * k}.", cfg.l_heads * cfg.latent_dim_head, d, False)
            ff(f"layers.{L}.{2 * M}.{2 * k + 1}.")
    if cfg.final_classifier_head:
        shapes["to_logits.1.weight"] = (d,)
        shapes["to_logits.1.bias"] = (d,)
        shapes["to_logits.2.weight"] = (cfg.out_dims, d)
        shapes["to_logits.2.bias"] = (cfg.out_dims,)
    return shapes


def filler_state_dict(cfg: FusionConfig, gain: float = 1.0) -> Dict[str, Tensor]:
    return {k: filler_tensor(k, s, gain) for k, s in state_dict_shapes(cfg).items()}
