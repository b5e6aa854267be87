"""Drop-in ``HealNet`` / ``Attention`` for MI355X.

Mirrors the class surface of the reference's ``healnet/models/healnet.py`` (constructor keywords
:15-38, ``forward(tensors, mask, return_embeddings, verbose)`` :190-195, ``get_attention_weights()``
:252-262, ``Attention(query_dim, context_dim, heads, dim_head, dropout)`` :369-370) and its
``state_dict`` key layout, so reference checkpoints load unchanged and
``torch.manual_seed(s)`` yields bit-identical initial weights (same RNG consumption order).

All arithmetic runs in the hand-written HIP kernels of ``libhealnet_hip.so`` through its C ABI
(``include/healnet_hip.h``); the modules below only hold parameters and marshal pointers.  There is no
CPU or eager-PyTorch fallback: CPU tensors or a missing library raise.

Deliberate deviations from the reference (documented in DESIGN.md):
  * the bare ``except`` around each cross block (:238) is not reproduced -- a missing modality
    (``None`` or a shorter list) is skipped explicitly with the reference's observable semantics
    (incl. the ``verbose=True`` quirk), while real shape errors raise;
  * the caller's list is not overwritten with the encoded context (:222) unless
    ``compat_mutate_inputs=True``;
  * ``attn_weights`` (:420) is materialised lazily on request instead of being kept for every block;
  * autograd: every forward is ONE torch.ops.healnet_hip.* call (ops.py); the backward is registered on the operator and runs
    hn_fusion_backward / hn_attn_bwd / hn_ff_bwd.  No gradient flows to modality inputs / attention contexts.
"""
from __future__ import annotations

import ctypes as C
import json
import weakref
from typing import Callable, Dict, List, Optional, Sequence

import torch
from torch import nn

from . import _capi

__all__ = ["HealNet", "Attention", "PreNorm", "FeedForward", "fourier_encode_concat", "latent_block"]


# ------------------------------------------------------------------------------------------------
# helpers (shared with ops.py)
# ------------------------------------------------------------------------------------------------
from ._rt import WS as _WS, WS_AUX as _WS_AUX, f32c as _f32c, mask_bytes as _mask_bytes, ptr as _ptr  # noqa: E402
from ._rt import require_gpu as _require_gpu, stream_ptr as _stream_ptr  # noqa: E402
from . import ops as _ops  # noqa: E402  (registers torch.ops.healnet_hip.*; every forward below dispatches through them)

_hip = torch.ops.healnet_hip


def fourier_encode_concat(data: torch.Tensor, num_freq_bands: int = 2, max_freq: float = 10.0,
                          fourier_encode_data: bool = True) -> torch.Tensor:
    """(b, *S, C) -> (b, prod S, C + axes*(2F+1)); HIP restatement of healnet.py:204-222 / :292-302."""
    _require_gpu(data, "modality tensor")
    return _hip.fourier_encode_concat(data, int(num_freq_bands), float(max_freq), bool(fourier_encode_data))


def exists(val) -> bool:                     # healnet.py:270-271
    return val is not None


def default(val, d):                         # healnet.py:273-274
    return val if exists(val) else d


def cache_fn(f):
    """Name-compatibility shim for healnet.py:276-290 (the product's weight tying goes through ``_Memo`` below): the returned
    callable takes the reference's ``_cache`` / ``key`` keywords; results are stored per ``key`` only."""
    store: dict = {}

    def cached_fn(*args, _cache=True, key=None, **kwargs):
        return _Memo(lambda: f(*args, **kwargs), store)(bool(_cache), key)
    cached_fn.__name__ = getattr(f, "__name__", "cached_fn")
    cached_fn.__doc__ = getattr(f, "__doc__", None)
    return cached_fn


def fourier_encode(x: torch.Tensor, max_freq: float, num_bands: int = 4) -> torch.Tensor:
    """healnet.py:292-302: ``(*S) -> (*S, 2*num_bands+1)`` = [sin(x s pi), cos(x s pi), x], s = linspace(1, max_freq/2, num_bands)."""
    _require_gpu(x, "x")
    xf = _f32c(x)
    out = torch.empty(*xf.shape, 2 * num_bands + 1, dtype=torch.float32, device=xf.device)
    _capi.check(_capi.lib().hn_fourier_encode(xf.data_ptr(), out.data_ptr(), xf.numel(), int(num_bands), float(max_freq),
                                              _stream_ptr(xf.device)), "hn_fourier_encode")
    return out


class _Gate(nn.Module):
    _gate = 0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_gpu(x, "x")
        xf = _f32c(x)
        hid = xf.shape[-1] // 2
        out = torch.empty(*xf.shape[:-1], hid, dtype=torch.float32, device=xf.device)
        _capi.check(_capi.lib().hn_glu_gate(xf.data_ptr(), out.data_ptr(), xf.numel() // (2 * hid), hid, self._gate,
                                            _stream_ptr(xf.device)), "hn_glu_gate")
        return out


class SELU(_Gate):
    """``x, gates = x.chunk(2, -1); x * F.selu(gates)`` (healnet.py:328-331)."""
    _gate = 0


class GELU(_Gate):
    """``x, gates = x.chunk(2, -1); x * F.gelu(gates)`` (healnet.py:323-326)."""
    _gate = 1


def temperature_softmax(logits: torch.Tensor, temperature: float = 1.0, dim: int = -1) -> torch.Tensor:
    """``F.softmax(logits / temperature, dim)`` (healnet/models/healnet.py:354-365) on the GPU; ``Attention.forward`` uses
    temperature 0.5 (:419), where the same function is fused into the attention core."""
    _require_gpu(logits, "logits")
    x = _f32c(logits)
    moved = dim not in (-1, x.dim() - 1)
    if moved:
        x = x.movedim(dim, -1).contiguous()
    y = torch.empty_like(x)
    n = x.shape[-1]
    _capi.check(_capi.lib().hn_temperature_softmax(x.data_ptr(), y.data_ptr(), x.numel() // n, n, float(temperature),
                                                   _stream_ptr(x.device)), "hn_temperature_softmax")
    return y.movedim(-1, dim) if moved else y


def _normalise_context(ctx: torch.Tensor, pitch: int) -> torch.Tensor:
    """Affine-free LayerNorm over the last dim of an already encoded (b, N, D) context -> (b, N, pitch)."""
    b, n, d = ctx.shape
    z = torch.empty(b, n, pitch, dtype=torch.float32, device=ctx.device)
    sp = (C.c_int * 1)(n)
    _capi.check(_capi.lib().hn_encode_norm(ctx.data_ptr(), b, 1, sp, d, 0, 0.0, 0, 1e-5, z.data_ptr(), pitch,
                                           _stream_ptr(ctx.device)), "hn_encode_norm")
    return z


# Every (re-)registration of a submodule or parameter anywhere in the process bumps this counter; HealNet._params() re-resolves its
# (module, name) slots when it has moved, so a replaced submodule / added parameter is never read through a stale slot.
_REGISTRATION_VERSION = [0]


def _bump_registration(*_args):
    _REGISTRATION_VERSION[0] += 1
    return None


nn.modules.module.register_module_module_registration_hook(_bump_registration)
nn.modules.module.register_module_parameter_registration_hook(_bump_registration)


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's state_dict keys
# ------------------------------------------------------------------------------------------------
class Attention(nn.Module):
    """Multi-head attention with temperature-0.5 softmax and a LeakyReLU(0.01) output projection.

    Keys: ``to_q.weight``, ``to_kv.weight``, ``to_out.0.weight``, ``to_out.0.bias`` (healnet.py:369-389)."""

    def __init__(self, query_dim: int, context_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 dropout: float = 0.0):
        super().__init__()
        inner = dim_head * heads
        self.query_dim = query_dim
        self.context_dim = query_dim if context_dim is None else context_dim
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.dropout_p = float(dropout)
        # construction order == RNG order of the reference: to_q, to_kv, to_out
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_kv = nn.Linear(self.context_dim, inner * 2, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.LeakyReLU(negative_slope=1e-2))
        self._probs_fn: Optional[Callable[[], torch.Tensor]] = None
        self._probs_owner_last = None      # the owner's `_last` record at the time of the last stand-alone call
        self._hn_owner = None              # weakref to the HealNet that runs this block (set by HealNet._bind_owner)
        self._hn_slots: List[int] = []     # its attention slots there, layer order (a tied block serves several layers)

    def __getstate__(self):                # weak references / closures do not pickle (torch.save(model), copy.deepcopy)
        state = dict(self.__dict__)
        state["_probs_fn"] = state["_probs_owner_last"] = state["_hn_owner"] = None
        return state

    # -- lazy Attention.attn_weights (:420) ------------------------------------------------------
    def _lazy(self, reduced: bool) -> Optional[torch.Tensor]:
        owner = self._hn_owner() if self._hn_owner is not None else None
        if self._probs_fn is not None and (owner is None or owner._last is self._probs_owner_last):
            return self._probs_fn(reduced=reduced)            # the stand-alone call is the most recent use of this module
        if owner is not None:
            return owner._block_probs(self, reduced)
        return None

    @property
    def attn_weights(self) -> Optional[torch.Tensor]:
        return self._lazy(False)

    @property
    def attn_importance(self) -> Optional[torch.Tensor]:
        """``attn_weights.mean(dim=1)`` -- the (b*heads, N) row-mean every consumer in the reference's explainer takes
        (explainer.py:161-164, :209-211) -- computed without materialising the (b*heads, L, N) matrix."""
        return self._lazy(True)

    def _params(self, norm: Optional[nn.LayerNorm], norm_context: Optional[nn.LayerNorm],
                dropout: float = 0.0) -> _capi.AttnParams:
        return _capi.AttnParams(
            dropout=float(dropout),
            heads=self.heads, dim_head=self.dim_head, query_dim=self.query_dim,
            norm_w=_ptr(norm.weight) if norm is not None else None,
            norm_b=_ptr(norm.bias) if norm is not None else None,
            ctx_gamma=_ptr(norm_context.weight) if norm_context is not None else None,
            ctx_beta=_ptr(norm_context.bias) if norm_context is not None else None,
            w_q=_ptr(self.to_q.weight), w_kv=_ptr(self.to_kv.weight),
            w_out=_ptr(self.to_out[0].weight), b_out=_ptr(self.to_out[0].bias))

    def _draw(self):
        """nn.Dropout(dropout) on the probabilities (healnet.py:381, :421) of a stand-alone call in TRAINING mode: (p, [seed, offset,
        stream]) for the kernels' counter-based generator (include/healnet_hip.h hn_rng) -- torch's seed, a per-module call counter
        (fresh masks every call; the backward replays the forward's), stream 0.  ``_last_rng`` keeps the triple of the last call:
        hn_dropout_mask rebuilds the mask from it (tests/test_gpu_dropout.py)."""
        return _draw_rng(self, 0)

    def _run(self, x: torch.Tensor, context: Optional[torch.Tensor], mask: Optional[torch.Tensor],
             norm: Optional[nn.LayerNorm], norm_context: Optional[nn.LayerNorm], residual: bool) -> torch.Tensor:
        """torch.ops.healnet_hip.attention_fwd (hn_attn_fwd / hn_attn_fwd_train): differentiable w.r.t. x and the parameters
        (hn_attn_bwd); the context gets no gradient -- a context that requires grad is refused."""
        drop_p, drop_rng = self._draw()
        if torch.is_grad_enabled() and context is not None and context.requires_grad:
            raise RuntimeError("healnet_amd: no gradient flows to the context of an Attention block (hn_attn_bwd; HEALNet's contexts "
                               "are encoded modality inputs) -- detach it")
        _require_gpu(x, "x")
        _require_gpu(self.to_q.weight, "Attention parameters")
        if x.dim() != 3 or x.shape[-1] != self.query_dim:
            raise ValueError(f"x must be (b, n, {self.query_dim}), got {tuple(x.shape)}")
        b, L, _ = x.shape
        ctx_z, ld, N, D = None, 0, L, self.query_dim
        if context is not None:
            _require_gpu(context, "context")
            if context.dim() != 3 or context.shape[0] != b or context.shape[-1] != self.context_dim:
                raise ValueError(f"context must be (b={b}, N, {self.context_dim}), got {tuple(context.shape)}")
            N, D = context.shape[1], context.shape[2]
            if norm_context is not None:
                ld = _capi.lib().hn_context_pitch(D, self.dim_head)
                ctx_z = _normalise_context(_f32c(context.detach()), ld)
            else:
                ctx_z, ld = _f32c(context.detach()), D
        mask_u8 = _mask_bytes(mask, b, N)
        opt = lambda m_, name: getattr(m_, name) if m_ is not None else None      # noqa: E731
        args = (opt(norm, "weight"), opt(norm, "bias"), opt(norm_context, "weight"), opt(norm_context, "bias"),
                self.to_q.weight, self.to_kv.weight, self.to_out[0].weight, self.to_out[0].bias)
        train = torch.is_grad_enabled() and (x.requires_grad or any(t is not None and t.requires_grad for t in args))
        # (dropout thins the probabilities in the training form of the entry point, which also keeps what the backward needs;
        # a training-mode call under no_grad still drops, as nn.Dropout does)
        out, stats, _ = _hip.attention_fwd(x, ctx_z, mask_u8, *args, self.heads, bool(residual), train or drop_p > 0.0, drop_p, drop_rng)
        xin = x.detach()

        def probs(reduced: bool = False) -> torch.Tensor:
            lib = _capi.lib()
            xf = _f32c(xin)
            pr = torch.empty((b * self.heads, N) if reduced else (b * self.heads, L, N), dtype=torch.float32, device=xf.device)
            pp = self._params(norm, norm_context)
            need = lib.hn_attn_workspace_bytes(C.byref(pp), int(ctx_z is not None), ld, b, L, N, D)
            aux = _WS_AUX.get(xf.device, need)
            fn = lib.hn_attn_importance if reduced else lib.hn_attn_probs
            _capi.check(fn(C.byref(pp), xf.data_ptr(), _ptr(ctx_z), ld, b, L, N, D, _ptr(mask_u8), stats.data_ptr(),
                           pr.data_ptr(), aux.data_ptr(), aux.numel(), _stream_ptr(xf.device)),
                        "hn_attn_importance" if reduced else "hn_attn_probs")
            return pr

        self._probs_fn = probs
        owner = self._hn_owner() if self._hn_owner is not None else None
        self._probs_owner_last = owner._last if owner is not None else None
        return out

    def forward(self, x: torch.Tensor, context: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self._run(x, context, mask, None, None, residual=False)


class FeedForward(nn.Module):
    """Linear(dim, 8 dim) -> a * gate(g) -> Linear(4 dim, dim); keys ``net.0.*`` / ``net.2.*`` (:339-351)."""

    def __init__(self, dim: int, mult: int = 4, dropout: float = 0.0, snn: bool = False):
        super().__init__()
        if mult != 4:
            raise NotImplementedError("healnet_amd: FeedForward mult must be 4 (the only value HealNet uses)")
        self.dim = dim
        self.snn = snn
        self.dropout_p = float(dropout)
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), nn.Identity(), nn.Linear(dim * mult, dim), nn.Identity())

    def _params(self, norm: Optional[nn.LayerNorm], dropout: float = 0.0) -> _capi.FFParams:
        return _capi.FFParams(dropout=float(dropout), dim=self.dim, gate=0 if self.snn else 1,
                              norm_w=_ptr(norm.weight) if norm is not None else None,
                              norm_b=_ptr(norm.bias) if norm is not None else None,
                              w1=_ptr(self.net[0].weight), b1=_ptr(self.net[0].bias),
                              w2=_ptr(self.net[2].weight), b2=_ptr(self.net[2].bias))

    def _run(self, x: torch.Tensor, norm: Optional[nn.LayerNorm], residual: bool) -> torch.Tensor:
        """torch.ops.healnet_hip.feed_forward (hn_ff_fwd, autograd through hn_ff_bwd)."""
        drop_p, drop_rng = _draw_rng(self, 0)       # nn.Dropout(dropout) on the block output (healnet.py:347), training mode only
        _require_gpu(x, "x")
        if x.shape[-1] != self.dim:
            raise ValueError(f"last dim must be {self.dim}, got {tuple(x.shape)}")
        return _hip.feed_forward(x, norm.weight if norm is not None else None, norm.bias if norm is not None else None,
                                 self.net[0].weight, self.net[0].bias, self.net[2].weight, self.net[2].bias, not self.snn,
                                 bool(residual), drop_p, drop_rng)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, None, residual=False)


class PreNorm(nn.Module):
    """LayerNorm(dim) on x (and LayerNorm(context_dim) on the context) fused into the wrapped block;
    keys ``fn.*``, ``norm.*``, ``norm_context.*`` (healnet.py:306-321)."""

    def __init__(self, dim: int, fn: nn.Module, context_dim: Optional[int] = None):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)
        self.norm_context = nn.LayerNorm(context_dim) if context_dim is not None else None

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if isinstance(self.fn, Attention):
            context = kwargs.get("context")
            if self.norm_context is not None and context is None:
                raise TypeError("PreNorm with context_dim needs a context tensor")      # reference: LayerNorm(None)
            return self.fn._run(x, context, kwargs.get("mask"), self.norm, self.norm_context, residual=False)
        if isinstance(self.fn, FeedForward):
            return self.fn._run(x, self.norm, residual=False)
        raise TypeError(f"PreNorm cannot wrap {type(self.fn).__name__}")


def _draw_rng(mod: nn.Module, stream: int, force: bool = False):
    """(dropout rate, [seed, offset, stream]) of one stand-alone call of a module with a ``dropout_p``: the rate is 0 (and nothing is
    drawn) outside training mode or without dropout."""
    p = float(mod.dropout_p) if (mod.training or force) else 0.0
    if p <= 0.0 and not force:
        return 0.0, []
    off = (mod.__dict__.get("_rng_offset", 0) + 1) & 0xFFFFFFFF
    mod.__dict__["_rng_offset"] = off
    rng = [_wrap64(torch.initial_seed()), off, int(stream)]
    mod.__dict__["_last_rng"] = (rng[0], rng[1], rng[2])
    return p, rng


def _wrap64(seed: int) -> int:
    """torch.initial_seed() is an unsigned 64-bit value; int64 tensors carry it two's-complement."""
    seed &= 0xFFFFFFFFFFFFFFFF
    return seed - (1 << 64) if seed >= (1 << 63) else seed


def latent_block(self_attn: "PreNorm", self_ff: "PreNorm", x: torch.Tensor) -> torch.Tensor:
    """One latent self block of the fusion loop, ``x = self_attn(x) + x; x = self_ff(x) + x`` (healnet.py:241-245), as ONE
    operator: torch.ops.healnet_hip.latent_block -> hn_latent_block_fwd (the fused latent chain for l_d = 128; differentiable
    through hn_latent_block_bwd).  ``self_attn`` / ``self_ff`` are the PreNorm-wrapped blocks, e.g. ``model.layers[l][-1]``."""
    att, ff = self_attn.fn, self_ff.fn
    if not isinstance(att, Attention) or not isinstance(ff, FeedForward) or self_attn.norm_context is not None:
        raise TypeError("latent_block takes PreNorm(Attention without context) and PreNorm(FeedForward)")
    _require_gpu(x, "x")
    args = (x, self_attn.norm.weight, self_attn.norm.bias, att.to_q.weight, att.to_kv.weight, att.to_out[0].weight, att.to_out[0].bias,
            att.heads, self_ff.norm.weight, self_ff.norm.bias, ff.net[0].weight, ff.net[0].bias, ff.net[2].weight, ff.net[2].bias, not ff.snn)
    pa = att.dropout_p if att.training else 0.0
    pf = ff.dropout_p if ff.training else 0.0
    if pa > 0.0 or pf > 0.0:
        # one (seed, offset) for the block: the attention draws on stream 0, the feed-forward block on stream 1 (ops.py _LB_DROP);
        # both modules remember their triple (hn_dropout_mask rebuilds the masks from it)
        _, rng = _draw_rng(att, 0, force=True)
        ff._last_rng = (rng[0], rng[1], 1)
        return _hip.latent_block_fwd(*args, True, pa, pf, rng)[0]
    return _hip.latent_block(*args)


class _MeanPool(nn.Module):
    """Parameter-free stand-in for the reference's einops ``Reduce('b n d -> b d', 'mean')`` at
    ``to_logits.0`` (keeps the LayerNorm / Linear at indices 1 / 2)."""

    def forward(self, x):  # pragma: no cover - the fused head kernel is used instead
        raise RuntimeError("healnet_amd: to_logits runs inside hn_head_fwd")


class _Memo:
    """Weight-tying helper with the reference's ``cache_fn`` semantics (healnet.py:278-290): a factory
    result is stored / reused per key only when caching is requested for that call."""

    def __init__(self, factory: Callable[[], nn.Module], store: Optional[dict] = None):
        self._factory = factory
        self._store: Dict[object, nn.Module] = {} if store is None else store

    def __call__(self, use_cache: bool, key=None) -> nn.Module:
        if not use_cache:
            return self._factory()
        if key not in self._store:
            self._store[key] = self._factory()
        return self._store[key]


# ------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------
class HealNet(nn.Module):
    def __init__(self, *, n_modalities: int, channel_dims: List, num_spatial_axes: List, out_dims: int, depth: int = 3,
                 num_freq_bands: int = 2, max_freq: float = 10., l_c: int = 128, l_d: int = 128, x_heads: int = 8,
                 l_heads: int = 8, cross_dim_head: int = 64, latent_dim_head: int = 64, attn_dropout: float = 0.,
                 ff_dropout: float = 0., weight_tie_layers: bool = False, fourier_encode_data: bool = True,
                 self_per_cross_attn: int = 1, final_classifier_head: bool = True, snn: bool = True,
                 core_precision: str = "fp32"):
        super().__init__()
        if core_precision not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("core_precision must be 'fp32', 'bf16' or 'bf16x3'")
        # Extension over the reference signature: matrix-instruction precision of the image / volume cross-attention
        # core in the no-grad forward ('bf16' = BASELINE configs[2]; 'bf16x3' = bf16 MFMA on hi+lo operand pairs, fp32-class
        # results; parameters, statistics and outputs stay fp32).
        self._core_precision = core_precision
        assert len(channel_dims) == len(num_spatial_axes), 'input channels and input axis must be of the same length'
        assert len(num_spatial_axes) == n_modalities, 'input axis must be of the same length as the number of modalities'

        self.input_axes = list(num_spatial_axes)
        self.input_channels = list(channel_dims)
        self.max_freq = max_freq
        self.num_freq_bands = num_freq_bands
        self.modalities = n_modalities
        self.self_per_cross_attn = self_per_cross_attn
        self.fourier_encode_data = fourier_encode_data
        self.depth = depth
        self.l_c, self.l_d = l_c, l_d
        self.out_dims = out_dims
        self.snn = snn
        self.final_classifier_head = final_classifier_head
        self.compat_mutate_inputs = False
        self.keep_attention_stats = True

        pos_channels = [(ax * (2 * num_freq_bands + 1)) if fourier_encode_data else 0 for ax in num_spatial_axes]
        self.context_dims = [int(c) + p for c, p in zip(channel_dims, pos_channels)]

        # RNG order of the reference constructor: latents, then per layer {latent attn, latent ff,
        # per modality {cross attn, cross ff}}, then the head Linear.
        self.latents = nn.Parameter(torch.randn(l_c, l_d))

        make_cross = [_Memo(lambda m=m: PreNorm(l_d, Attention(l_d, self.context_dims[m], heads=x_heads,
                                                               dim_head=cross_dim_head, dropout=attn_dropout),
                                                context_dim=self.context_dims[m])) for m in range(n_modalities)]
        make_cross_ff = _Memo(lambda: PreNorm(l_d, FeedForward(l_d, dropout=ff_dropout, snn=snn)))
        make_latent_attn = _Memo(lambda: PreNorm(l_d, Attention(l_d, heads=l_heads, dim_head=latent_dim_head,
                                                                dropout=attn_dropout)))
        make_latent_ff = _Memo(lambda: PreNorm(l_d, FeedForward(l_d, dropout=ff_dropout, snn=snn)))

        self.layers = nn.ModuleList([])
        for layer in range(depth):
            tie = layer > 0 and weight_tie_layers
            latent_blocks = nn.ModuleList([])
            for k in range(self_per_cross_attn):
                latent_blocks.append(make_latent_attn(tie, key=k))
                latent_blocks.append(make_latent_ff(tie, key=k))
            blocks: List[nn.Module] = []
            for m in range(n_modalities):
                blocks.append(make_cross[m](tie))
                blocks.append(make_cross_ff(tie))
            self.layers.append(nn.ModuleList([*blocks, latent_blocks]))

        self.to_logits = nn.Sequential(_MeanPool(), nn.LayerNorm(l_d), nn.Linear(l_d, out_dims)) \
            if final_classifier_head else nn.Identity()

        self._last: Optional[dict] = None
        self._any_dropout = attn_dropout > 0.0 or ff_dropout > 0.0
        self._rng_offset = 0
        self._last_rng = None
        self._bind_owner()
        self._spec_text = self._build_spec()

    # -- structure description handed to torch.ops.healnet_hip.fusion_* ---------------------------------
    @property
    def core_precision(self) -> str:
        return self._core_precision

    @core_precision.setter
    def core_precision(self, value: str) -> None:
        if value not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("core_precision must be 'fp32', 'bf16' or 'bf16x3'")
        self._core_precision = value
        if "_spec_text" in self.__dict__:
            self._spec_text = self._build_spec()

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.pop("_hn_param_slots", None)
        self._bind_owner()                 # weak back-references do not survive pickling / deepcopy (Attention.__getstate__)

    def _bind_owner(self) -> None:
        """Every Attention block learns which model runs it and in which attention slots (layer * (M + 1) + modality, M = the
        latent self-attention), so that ``Attention.attn_weights`` (:420) can be served lazily from the last forward."""
        M = self.modalities
        for mod in self.modules():
            if isinstance(mod, Attention):
                mod._hn_slots = []
        ref = weakref.ref(self)
        for layer in range(self.depth):
            mods = self.layers[layer]
            owned = [mods[2 * m].fn for m in range(M)] + ([mods[2 * M][0].fn] if self.self_per_cross_attn >= 1 else [])
            for j, att in enumerate(owned):
                att._hn_owner = ref
                att._hn_slots.append(layer * (M + 1) + (j if j < M else M))

    def _build_spec(self) -> str:
        """JSON description of the module structure for the fusion operators (ops.Spec): sizes plus, for every pointer field of
        hn_model, the position of its tensor in ``list(self.parameters())`` (tied blocks repeat positions)."""
        index = {id(p): i for i, p in enumerate(self.parameters())}
        ix = lambda t: -1 if t is None else index[id(t)]      # noqa: E731
        M, depth = self.modalities, self.depth

        def attn(blk, cross):
            a = blk.fn
            nc = blk.norm_context if cross else None
            return dict(heads=a.heads, dim_head=a.dim_head, query_dim=a.query_dim, dropout=a.dropout_p,
                        p=[ix(blk.norm.weight), ix(blk.norm.bias), ix(nc.weight if nc is not None else None),
                           ix(nc.bias if nc is not None else None), ix(a.to_q.weight), ix(a.to_kv.weight), ix(a.to_out[0].weight),
                           ix(a.to_out[0].bias)])

        def ff(blk):
            f = blk.fn
            return dict(dim=f.dim, gate=0 if f.snn else 1, dropout=f.dropout_p,
                        p=[ix(blk.norm.weight), ix(blk.norm.bias), ix(f.net[0].weight), ix(f.net[0].bias), ix(f.net[2].weight),
                           ix(f.net[2].bias)])

        cross_attn, cross_ff, self_attn, self_ff = [], [], [], []
        for layer in range(depth):
            mods = self.layers[layer]
            for m in range(M):
                cross_attn.append(attn(mods[2 * m], True))
                cross_ff.append(ff(mods[2 * m + 1]))
            if self.self_per_cross_attn >= 1:
                self_attn.append(attn(mods[2 * M][0], False))
                self_ff.append(ff(mods[2 * M][1]))
            else:   # never executed; keeps the arrays depth long
                self_attn.append(dict(heads=1, dim_head=1, query_dim=self.l_d, dropout=0.0, p=[-1] * 8))
                self_ff.append(dict(dim=self.l_d, gate=0, dropout=0.0, p=[-1] * 6))
        head = self.final_classifier_head
        return json.dumps(dict(
            M=M, depth=depth, l_c=self.l_c, l_d=self.l_d, self_per_cross_attn=self.self_per_cross_attn, head=int(head),
            out_dims=self.out_dims, num_freq_bands=self.num_freq_bands, max_freq=float(self.max_freq),
            fourier=int(self.fourier_encode_data), channels=[int(c) for c in self.input_channels],
            axes=[int(a) for a in self.input_axes], latents=ix(self.latents), cross_attn=cross_attn, cross_ff=cross_ff,
            self_attn=self_attn, self_ff=self_ff,
            head_p=[ix(self.to_logits[1].weight), ix(self.to_logits[1].bias), ix(self.to_logits[2].weight),
                    ix(self.to_logits[2].bias)] if head else None,
            core_precision={"fp32": _capi.HN_CORE_F32, "bf16": _capi.HN_CORE_BF16, "bf16x3": _capi.HN_CORE_BF16X3}[self.core_precision]),
            sort_keys=True)

    def _params(self) -> List[torch.Tensor]:
        """``list(self.parameters())`` without the module-tree walk (~0.2 ms of host time per forward -- the walk visits ~100
        modules): the (module, name) slots are resolved once, the Parameter objects are read from them on every call, so
        re-assigned / re-homed parameters are seen.  Same order and de-duplication (tied blocks) as ``parameters()``."""
        slots = self.__dict__.get("_hn_param_slots")
        if slots is not None and self.__dict__.get("_hn_param_ver") != _REGISTRATION_VERSION[0]:
            slots = None                                     # some module / parameter was (re-)registered since: resolve again
        if slots is None:
            slots, seen = [], set()
            for mod in self.modules():                       # de-duplicated, registration order: what named_parameters() walks
                for name, prm in mod._parameters.items():
                    if prm is None or id(prm) in seen:
                        continue
                    seen.add(id(prm))
                    slots.append((mod._parameters, name))
            self.__dict__["_hn_param_slots"] = slots
            self.__dict__["_hn_param_ver"] = _REGISTRATION_VERSION[0]
        return [d[n] for d, n in slots]

    def _descriptor(self, rng=None):
        """(hn_model, keep-alive) for this module's parameters; rng = (seed, offset) of a training forward: blocks then carry
        their dropout rates (include/healnet_hip.h hn_rng); None = inference descriptor."""
        r = None if rng is None else torch.tensor([_wrap64(rng[0]), int(rng[1])], dtype=torch.int64)
        return _ops.spec_of(self._spec_text).model(list(self.parameters()), r)

    def runs_staged(self) -> bool:
        """True when the fused entry points run this model as its zero-padded image (include/healnet_hip.h "Staged models": the
        reference's tuned shapes -- odd latent widths, one narrow cross head, a latent count that is not a multiple of 16)."""
        model, keep = self._descriptor()
        return bool(_capi.lib().hn_fusion_is_staged(C.byref(model)))

    def _check_mode(self) -> None:
        if self.self_per_cross_attn >= 2:
            raise ValueError("self_per_cross_attn >= 2 fails in the reference as well (healnet.py:242: "
                             "`self_attn, self_ff = layer[-1]`)")

    def _dropout_active(self) -> bool:
        """nn.Dropout semantics: masks are drawn in training mode only (healnet.py:381,421 attention, :347 feed-forward)."""
        return self.training and self._any_dropout

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, tensors: List[Optional[torch.Tensor]], mask: Optional[torch.Tensor] = None,
                return_embeddings: bool = False, verbose: bool = False, _profile=None):
        """One operator call: torch.ops.healnet_hip.fusion_forward (no autograd) or fusion_forward_train (tape + registered
        backward) -- healnet.py:190-250 of the reference."""
        self._check_mode()
        M = self.modalities
        if len(tensors) > M:
            raise ValueError(f"{len(tensors)} tensors passed to a model with {M} modalities")
        n_given = len(tensors)
        # the reference's missing_idx holds the None entries INSIDE the list (:193); a modality beyond a shorter list fails in
        # the bare try/except (:238) instead -- it is skipped too, but the verbose=True `continue` quirk never applies to it
        none_idx = [i for i in range(n_given) if tensors[i] is None]
        missing_idx = none_idx + list(range(n_given, M))
        if verbose:
            print(f"Missing modalities indices: {none_idx}")
        _require_gpu(self.latents, "HealNet parameters")
        device = self.latents.device
        held: List[Optional[torch.Tensor]] = [None] * M
        b = None
        for i in range(n_given):              # the reference's sanity check (:206-208), all modalities first
            if tensors[i] is not None:
                assert tensors[i].dim() - 2 == self.input_axes[i], (f'input data for modality {i + 1} must hav'
                                                                    f' the same number of axis as the input axis parameter')
        for i in range(n_given):
            data = tensors[i]
            if data is None:
                continue
            _require_gpu(data, f"modality {i + 1}")
            if data.device != device:
                raise RuntimeError(f"healnet_amd: modality {i + 1} lives on {data.device}, the model on {device}")
            bb, ch = data.shape[0], data.shape[-1]
            if ch != self.input_channels[i]:
                raise ValueError(f"modality {i + 1}: expected {self.input_channels[i]} channels, got {ch} (the reference "
                                 "would silently skip every block of this modality, Appendix B-7)")
            if data.dim() - 2 > _capi.HN_MAX_AXES:
                raise NotImplementedError(f"at most {_capi.HN_MAX_AXES} spatial axes are supported")
            if b is None:
                b = bb
            elif bb != b:
                raise ValueError("batch dim must be identical across modalities")
            # bf16 tensors are read as they are, uint8 tensors as byte / 255 (8-bit image transport, == ToTensor);
            # anything else is staged as fp32
            held[i] = data.contiguous() if data.dtype in (torch.bfloat16, torch.uint8) else _f32c(data)
        if b is None:
            raise ValueError("at least one modality must be present")
        mask_u8 = None
        if mask is not None:
            flat = mask.reshape(mask.shape[0], -1)
            for i in range(M):
                if held[i] is not None:
                    n_i = held[i].numel() // (b * held[i].shape[-1])
                    if flat.shape[0] != b or flat.shape[1] != n_i:
                        raise ValueError(f"mask {tuple(mask.shape)} does not match modality {i + 1} (b={b}, N={n_i}); "
                                         "the mask is applied to every modality's cross-attention (Appendix B-5)")
            mask_u8 = flat.to(device=device, dtype=torch.uint8).contiguous()

        params = self._params()
        embeddings = bool(return_embeddings) or not self.final_classifier_head
        skip_bits = sum(1 << i for i in none_idx) if verbose else 0
        dropping = self._dropout_active()
        taping = dropping or (torch.is_grad_enabled() and any(p.requires_grad for p in params))
        if taping:
            # autograd route (train or eval mode alike, as in PyTorch): tape-recording forward, backward registered on the op.
            # Dropout masks are drawn in training mode only, grad mode or not (nn.Dropout); every forward advances the
            # Philox offset, the seed is torch's global seed (torch.manual_seed).
            if _profile is not None:
                raise ValueError("profiling hooks are only available under torch.no_grad() in eval mode")
            rng, rng_t = None, None
            if dropping:
                self._rng_offset = (self._rng_offset + 1) & 0xFFFFFFFF
                rng = (torch.initial_seed(), self._rng_offset)
                word = self.__dict__.get("_hn_rng_word")      # healnet_amd.train.GraphedStep: the device word added to the offset
                rng_t = torch.tensor([_wrap64(rng[0]), rng[1]] + ([word.data_ptr()] if word is not None else []), dtype=torch.int64)
            self._last_rng = rng
            # healnet_amd.train.flatten_parameters(): every p.grad is a view of one flat buffer -> the backward accumulates
            # straight into it (no per-parameter zero tensors, no AccumulateGrad pass) and autograd gets None back
            gbuf, goffs = None, []
            flatp = self.__dict__.get("_hn_flat")
            if flatp is not None and not torch.compiler.is_compiling():
                goffs = flatp.direct_offsets(params)
                gbuf = flatp.grads if goffs else None
            if _ops.FORCE_TORCH_OPS or torch.compiler.is_compiling():
                out, tape, layout = _hip.fusion_forward_train(held, mask_u8, params, self._spec_text, skip_bits, embeddings, rng_t, gbuf,
                                                              goffs or [])
            else:      # the same implementation behind a lean autograd.Function (ops.FusionTrainFn explains: ~0.7 ms of dispatcher time per step)
                out, tape, layout = _ops.FusionTrainFn.apply((len(held), mask_u8, rng_t, self._spec_text, skip_bits, embeddings, gbuf,
                                                              goffs or []), *held, *params)
            self._last = dict(tape=tape, layout=layout, inputs=held, mask=mask_u8, b=b) if self.keep_attention_stats else None
        else:
            if _profile is not None:
                _ops.set_profile(_profile)
            # (nothing to differentiate on this branch: outside torch.compile the implementation is called directly, without the
            # dispatcher's ~0.1 ms of list / pytree handling per call -- the b <= 4 forwards are host-bound)
            fwd = _hip.fusion_forward if (_ops.FORCE_TORCH_OPS or torch.compiler.is_compiling()) else _ops._fusion_forward
            out, stats, trace = fwd(held, mask_u8, params, self._spec_text, skip_bits, embeddings, bool(self.keep_attention_stats))
            self._last = dict(stats=stats, trace=trace, inputs=held, mask=mask_u8, b=b) if self.keep_attention_stats else None
        if verbose:
            for layer in range(self.depth):
                for i in none_idx:
                    print(f"Skipping update in fusion layer {layer + 1} for missing modality {i + 1}")
        if self.compat_mutate_inputs:
            for i in range(n_given):
                if held[i] is not None:
                    tensors[i] = fourier_encode_concat(_f32c(held[i]), self.num_freq_bands, self.max_freq, self.fourier_encode_data)
        return out

    # -- small-batch latency path: graph replay owned by the model ------------------------------------
    def capture(self, tensors: List[Optional[torch.Tensor]], mask: Optional[torch.Tensor] = None,
                return_embeddings: bool = False) -> "GraphedForward":
        """Capture ONE inference forward for these input shapes into a HIP graph and return a callable that replays it
        (``hipGraphLaunch``: one host call instead of the ~30 kernel launches + descriptor build of an eager forward).

        At b <= 4 -- the reference's README call is b = 1 (README.md:96-110), BASELINE configs[0] is b = 4 -- the eager forward
        is host-bound: ~0.8 ms to enqueue against 0.3-0.5 ms of GPU work.  The replay uses a static workspace, static softmax
        statistics / trace buffers (so ``get_attention_weights()`` keeps working after a replay) and static input buffers that
        ``__call__`` copies into (pass the graph's own ``.inputs`` tensors to skip the copy).  Inference only: eval mode, no
        autograd; parameters are read in place at replay time, so in-place weight updates are seen, re-homed parameters
        (``.to()``, ``flatten_parameters``) need a new capture."""
        return GraphedForward(self, tensors, mask, return_embeddings)

    # -- attention weights on demand ---------------------------------------------------------------
    def _slot_buffers(self, slot: int):
        """(stats (b, heads, l_c, 2), block input (b, l_c, l_d)) of an attention slot of the last forward, or None."""
        last = self._last
        if last is None:
            return None
        M, b = self.modalities, last["b"]
        layer, j = divmod(slot, M + 1)
        heads = self.layers[layer][2 * j].fn.heads if j < M else self.layers[layer][2 * M][0].fn.heads
        n_stats, n_x = b * heads * self.l_c * 2, b * self.l_c * self.l_d
        if "tape" in last:
            n_slots = self.depth * (M + 1)
            so, xo = int(last["layout"][slot]), int(last["layout"][n_slots + slot])
            if so < 0:
                return None
            tf = last["tape"].view(torch.float32)
            return tf[so:so + n_stats].view(b, heads, self.l_c, 2), tf[xo:xo + n_x].view(b, self.l_c, self.l_d)
        live = (j < M and last["inputs"][j] is not None) or (j == M and self.self_per_cross_attn > 0)
        if not live:
            return None
        return last["stats"][slot, :n_stats].view(b, heads, self.l_c, 2), last["trace"][slot]

    def _block_probs(self, att: "Attention", reduced: bool) -> Optional[torch.Tensor]:
        """``att.attn_weights`` (or its latent-row mean) from the last forward: the most recent slot the block ran in."""
        last = self._last
        if last is None:
            return None
        M = self.modalities
        for slot in reversed(att._hn_slots):
            bufs = self._slot_buffers(slot)
            if bufs is None:
                continue
            stats, xin = bufs
            layer, j = divmod(slot, M + 1)
            blk = self.layers[layer][2 * j] if j < M else self.layers[layer][2 * M][0]
            lib = _capi.lib()
            b, L, dev = last["b"], self.l_c, xin.device
            with torch.cuda.device(dev):
                if j < M:
                    zc = last.setdefault("zcache", {})
                    if j not in zc:
                        d = self.context_dims[j]
                        ld = lib.hn_context_pitch(d, att.dim_head)
                        z = _hip.encode_norm(_f32c(last["inputs"][j]), self.num_freq_bands, float(self.max_freq),
                                             bool(self.fourier_encode_data), ld)
                        zc[j] = (z, ld, z.shape[1], d)
                    z, ld, n, d = zc[j]
                    p, msk = att._params(blk.norm, blk.norm_context), last["mask"]
                else:
                    z, ld, n, d = None, 0, L, self.l_d
                    p, msk = att._params(blk.norm, None), None
                need = lib.hn_attn_workspace_bytes(C.byref(p), int(z is not None), ld, b, L, n, d)
                aux = _WS_AUX.get(dev, need)
                pr = torch.empty((b * att.heads, n) if reduced else (b * att.heads, L, n), dtype=torch.float32, device=dev)
                fn = lib.hn_attn_importance if reduced else lib.hn_attn_probs
                _capi.check(fn(C.byref(p), xin.data_ptr(), _ptr(z), ld, b, L, n, d, _ptr(msk), stats.data_ptr(), pr.data_ptr(),
                               aux.data_ptr(), aux.numel(), _stream_ptr(dev)), "hn_attn_importance" if reduced else "hn_attn_probs")
            return pr
        return None

    def get_attention_weights(self) -> List[Optional[torch.Tensor]]:
        """Every ``Attention.attn_weights`` in ``self.modules()`` order (healnet.py:252-262): per layer
        [cross_0 .. cross_{M-1}, self]; shape (b*heads, l_c, N) each, ``None`` for blocks that did not run."""
        return [mod.attn_weights for mod in self.modules() if isinstance(mod, Attention)]

    def get_attention_importance(self) -> List[Optional[torch.Tensor]]:
        """``[w.mean(dim=1) for w in get_attention_weights()]`` -- shape (b*heads, N) each -- without ever forming the
        (b*heads, l_c, N) matrices: the reduction the reference's explainer applies to every entry (explainer.py:161-164,
        :209-211).  Same order and ``None`` convention as ``get_attention_weights``."""
        return [mod.attn_importance for mod in self.modules() if isinstance(mod, Attention)]


class GraphedForward:
    """A captured inference forward of one HealNet for fixed input shapes (``HealNet.capture``): healnet.py:190-250 of the
    reference as ONE graph launch.  ``out`` is a static tensor that the next replay overwrites -- clone it to keep it."""

    def __init__(self, model: HealNet, tensors, mask=None, return_embeddings: bool = False, warmup: int = 2):
        if model.training and model._any_dropout:
            raise RuntimeError("HealNet.capture: the graph replays the inference forward (call model.eval() first)")
        self.model = model
        self.return_embeddings = bool(return_embeddings)
        dev = model.latents.device
        self.device = dev
        self.inputs: List[Optional[torch.Tensor]] = []
        for t in tensors:
            if t is None:
                self.inputs.append(None)
                continue
            _require_gpu(t, "modality tensor")
            self.inputs.append((t if t.dtype in (torch.bfloat16, torch.uint8) else t.float()).contiguous().clone())
        self.mask = None if mask is None else mask.to(dev).clone()
        self.warmup = max(1, int(warmup))
        self.captures = 0
        self._capture()

    def _dev_index(self) -> int:
        return self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _capture(self) -> None:
        model, dev = self.model, self.device
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(self.warmup):             # first-call work (LDS opt-in of the chain kernel, workspace growth) stays outside
                model(list(self.inputs), mask=self.mask, return_embeddings=self.return_embeddings)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.cuda.device(dev):
            st = _capi.cluster_status(self._dev_index())
            if st["pending"]:                        # a warm-up run lost a cluster exchange: consume it, capture without clusters
                _capi.note_coresidency(_capi.CoresidencyLost(_capi.HN_E_CORESIDENCY, "GraphedForward", "a warm-up run lost a cluster "
                                                             "exchange; capturing without cluster mode"), self._dev_index(), lambda: None)
                st = _capi.cluster_status(self._dev_index())
        # the cluster decision is baked into the captured launches: remember the state it was taken under (see __call__)
        self._cluster_epoch = (st["lost"], st["enabled"])
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = model(list(self.inputs), mask=self.mask, return_embeddings=self.return_embeddings)
        self._last = model._last                       # statistics / trace buffers of the capture: rewritten by every replay
        self.captures += 1

    def __call__(self, tensors=None, mask=None) -> torch.Tensor:
        """Replay on new values of the captured shapes (``None`` / omitted: whatever the static inputs hold)."""
        if tensors is not None:
            if len(tensors) != len(self.inputs):
                raise ValueError(f"captured with {len(self.inputs)} modality entries, called with {len(tensors)}")
            for dst, src in zip(self.inputs, tensors):
                if (dst is None) != (src is None):
                    raise ValueError("the set of missing modalities is part of the captured graph")
                if dst is None or src is dst:
                    continue
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError(f"captured for shape {tuple(dst.shape)}, got {tuple(src.shape)}: capture again")
                dst.copy_(src, non_blocking=True)
        if mask is not None:
            if self.mask is None:
                raise ValueError("captured without a mask")
            self.mask.copy_(mask, non_blocking=True)
        # No entry point of the library runs during a replay, so the cluster-mode failure signal (include/healnet_hip.h) is polled
        # here (one read of a host-mapped word): after a lost exchange -- reported by a replay of this graph, or consumed elsewhere
        # since the capture -- the graph still holds cluster launches and is captured again without them.  The outputs of the
        # replay that LOST the exchange hold NaN rows (never a silently incomplete sum); the next call is clean.
        st = _capi.cluster_status(self._dev_index())
        if st["pending"] or self._cluster_epoch != (st["lost"], st["enabled"]):
            torch.cuda.synchronize(self.device)
            if st["pending"]:
                _capi.note_coresidency(_capi.CoresidencyLost(_capi.HN_E_CORESIDENCY, "GraphedForward", "a replayed cluster-mode latent "
                                                             "chain lost an exchange; the forward is captured again without cluster mode"),
                                       self._dev_index(), lambda: None)
            self._capture()
        self.graph.replay()
        if self._last is not None:
            self._last.pop("zcache", None)             # normalised contexts cached by an attention-weight export of older inputs
        self.model._last = self._last
        return self.out
