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
  * dropout > 0 in training mode is not implemented yet (raises); autograd runs through hn_fusion_backward.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional, Sequence

import torch
from torch import nn

from . import _capi

__all__ = ["HealNet", "Attention", "PreNorm", "FeedForward", "fourier_encode_concat"]


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"healnet_amd: {what} must live on a HIP device (got {t.device}); "
                           "the MI355X path has no CPU fallback")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32, contiguous view of an input (positions are always computed in fp32, Appendix B-8); uint8 means byte / 255."""
    if t.dtype == torch.uint8:
        t = t.float().div(255)
    elif t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address of a parameter / auxiliary tensor handed to the C ABI, which reads fp32, dense, row-major memory:
    anything else (model.half() / .bfloat16() / .double(), a transposed view) would be read as garbage -- refuse it."""
    if t is None:
        return None
    if t.is_floating_point() and t.dtype != torch.float32:
        raise TypeError(f"healnet_amd: parameters must be float32 (got {t.dtype}); the kernels read fp32 memory -- keep the module in "
                        "fp32 (bf16 modality TENSORS are fine: pass them to forward as they are)")
    if not t.is_contiguous():
        raise ValueError("healnet_amd: parameters / auxiliary tensors must be contiguous")
    return t.data_ptr()


_POISON = os.environ.get("HN_POISON_WS", "0") == "1"


def _reject_autograd_inputs(what: str, *tensors: Optional[torch.Tensor]) -> None:
    """The stand-alone blocks are forward-only ops: an input that requires grad means the caller expects gradients to
    flow through them, which would silently not happen -- refuse instead (training runs through HealNet.forward)."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise RuntimeError(f"healnet_amd: the stand-alone {what} block is a forward-only op (no autograd); an input requires grad. "
                           "Train through HealNet.forward (fused tape + backward), or detach the input / use torch.no_grad().")


class _Workspace:
    """One growing scratch allocation per (device, stream) (the C ABI never allocates).  Per stream, because calls on
    different streams of one device may run concurrently (two micro-batches, a serving thread per stream) and must not
    share scratch; calls on one stream are ordered, so they can."""

    def __init__(self) -> None:
        self._buf: Dict[tuple, torch.Tensor] = {}

    def get(self, device: torch.device, nbytes: int) -> torch.Tensor:
        key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        if _POISON:      # development aid (HN_POISON_WS=1): every call starts from an all-NaN workspace, so a kernel that
            buf.fill_(0xFF)   # reads scratch it has not written shows up deterministically
        return buf


_WS = _Workspace()
_WS_AUX = _Workspace()   # for on-demand attention-weight export (must not clobber the forward scratch)


def _mask_bytes(mask: Optional[torch.Tensor], b: int, n: int) -> Optional[torch.Tensor]:
    if mask is None:
        return None
    flat = mask.reshape(mask.shape[0], -1)
    if flat.shape[0] != b or flat.shape[1] != n:
        raise ValueError(f"mask of shape {tuple(mask.shape)} does not match context tokens (b={b}, N={n})")
    return flat.to(torch.uint8).contiguous()


def fourier_encode_concat(data: torch.Tensor, num_freq_bands: int = 2, max_freq: float = 10.0,
                          fourier_encode_data: bool = True) -> torch.Tensor:
    """(b, *S, C) -> (b, prod S, C + axes*(2F+1)); HIP restatement of healnet.py:204-222 / :292-302."""
    _require_gpu(data, "modality tensor")
    x = _f32c(data)
    b, spatial, ch = x.shape[0], list(x.shape[1:-1]), x.shape[-1]
    if not 1 <= len(spatial) <= _capi.HN_MAX_AXES:
        raise ValueError(f"1..{_capi.HN_MAX_AXES} spatial axes supported, got {len(spatial)}")
    n = 1
    for s in spatial:
        n *= s
    d = ch + (len(spatial) * (2 * num_freq_bands + 1) if fourier_encode_data else 0)
    out = torch.empty(b, n, d, dtype=torch.float32, device=x.device)
    sp = (C.c_int * len(spatial))(*spatial)
    _capi.check(_capi.lib().hn_fourier_encode_concat(x.data_ptr(), b, len(spatial), sp, ch, num_freq_bands, float(max_freq),
                                                     int(fourier_encode_data), out.data_ptr(), d, _stream_ptr(x.device)),
                "hn_fourier_encode_concat")
    return out


def exists(val) -> bool:                     # healnet.py:270-271
    return val is not None


def default(val, d):                         # healnet.py:273-274
    return val if exists(val) else d


def cache_fn(f):
    """healnet.py:276-290: memoise a factory by ``key`` while ``_cache`` is true -- how weight tying is implemented."""
    import functools
    cache: dict = {}

    @functools.wraps(f)
    def cached_fn(*args, _cache=True, key=None, **kwargs):
        if not _cache:
            return f(*args, **kwargs)
        if key in cache:
            return cache[key]
        result = f(*args, **kwargs)
        cache[key] = result
        return result
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

    # -- lazy Attention.attn_weights (:420) ------------------------------------------------------
    @property
    def attn_weights(self) -> Optional[torch.Tensor]:
        return None if self._probs_fn is None else self._probs_fn()

    @property
    def attn_importance(self) -> Optional[torch.Tensor]:
        """``attn_weights.mean(dim=1)`` -- the (b*heads, N) row-mean every consumer in the reference's explainer takes
        (explainer.py:161-164, :209-211) -- computed without materialising the (b*heads, L, N) matrix."""
        return None if self._probs_fn is None else self._probs_fn(reduced=True)

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

    def _grads(self, norm: Optional[nn.LayerNorm], norm_context: Optional[nn.LayerNorm], gmap) -> _capi.AttnGrads:
        gp = lambda t: None if t is None or id(t) not in gmap else gmap[id(t)].data_ptr()   # noqa: E731
        return _capi.AttnGrads(
            norm_w=gp(norm.weight) if norm is not None else None, norm_b=gp(norm.bias) if norm is not None else None,
            ctx_gamma=gp(norm_context.weight) if norm_context is not None else None,
            ctx_beta=gp(norm_context.bias) if norm_context is not None else None,
            w_q=gp(self.to_q.weight), w_kv=gp(self.to_kv.weight), w_out=gp(self.to_out[0].weight), b_out=gp(self.to_out[0].bias))

    def _check_mode(self) -> None:
        if self.training and self.dropout_p > 0.0:
            raise NotImplementedError("healnet_amd: the stand-alone Attention / FeedForward modules are inference ops; dropout "
                                      "(and autograd) run through HealNet's fused training path -- call .eval() here")

    def _run(self, x: torch.Tensor, context: Optional[torch.Tensor], mask: Optional[torch.Tensor],
             norm: Optional[nn.LayerNorm], norm_context: Optional[nn.LayerNorm], residual: bool) -> torch.Tensor:
        self._check_mode()
        _reject_autograd_inputs("Attention", x, context)
        _require_gpu(x, "x")
        _require_gpu(self.to_q.weight, "Attention parameters")
        x = _f32c(x)
        if x.dim() != 3 or x.shape[-1] != self.query_dim:
            raise ValueError(f"x must be (b, n, {self.query_dim}), got {tuple(x.shape)}")
        b, L, _ = x.shape
        lib = _capi.lib()
        ctx_z, ld, N, D = None, 0, L, self.query_dim
        if context is not None:
            _require_gpu(context, "context")
            ctx = _f32c(context)
            if ctx.dim() != 3 or ctx.shape[0] != b or ctx.shape[-1] != self.context_dim:
                raise ValueError(f"context must be (b={b}, N, {self.context_dim}), got {tuple(ctx.shape)}")
            N, D = ctx.shape[1], ctx.shape[2]
            if norm_context is not None:
                ld = lib.hn_context_pitch(D, self.dim_head)
                ctx_z = _normalise_context(ctx, ld)
            else:
                ctx_z, ld = ctx, D
        mask_u8 = _mask_bytes(mask, b, N)
        p = self._params(norm, norm_context)
        need = lib.hn_attn_workspace_bytes(C.byref(p), int(ctx_z is not None), ld, b, L, N, D)
        if need == 0:
            _capi.check(-1, "hn_attn_workspace_bytes")
        ws = _WS.get(x.device, need)
        out = torch.empty_like(x)
        stats = torch.empty(b, self.heads, L, 2, dtype=torch.float32, device=x.device)
        _capi.check(lib.hn_attn_fwd(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), _ptr(ctx_z), ld, b, L, N, D,
                                    _ptr(mask_u8), stats.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(x.device)),
                    "hn_attn_fwd")

        def probs(reduced: bool = False) -> torch.Tensor:
            pr = torch.empty((b * self.heads, N) if reduced else (b * self.heads, L, N), dtype=torch.float32, device=x.device)
            pp = self._params(norm, norm_context)
            aux = _WS_AUX.get(x.device, need)
            fn = lib.hn_attn_importance if reduced else lib.hn_attn_probs
            _capi.check(fn(C.byref(pp), x.data_ptr(), _ptr(ctx_z), ld, b, L, N, D, _ptr(mask_u8), stats.data_ptr(),
                           pr.data_ptr(), aux.data_ptr(), aux.numel(), _stream_ptr(x.device)),
                        "hn_attn_importance" if reduced else "hn_attn_probs")
            return pr

        self._probs_fn = probs
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

    def _grads(self, norm: Optional[nn.LayerNorm], gmap) -> _capi.FFGrads:
        gp = lambda t: None if t is None or id(t) not in gmap else gmap[id(t)].data_ptr()   # noqa: E731
        return _capi.FFGrads(norm_w=gp(norm.weight) if norm is not None else None, norm_b=gp(norm.bias) if norm is not None else None,
                             w1=gp(self.net[0].weight), b1=gp(self.net[0].bias), w2=gp(self.net[2].weight), b2=gp(self.net[2].bias))

    def _run(self, x: torch.Tensor, norm: Optional[nn.LayerNorm], residual: bool) -> torch.Tensor:
        if self.training and self.dropout_p > 0.0:
            raise NotImplementedError("healnet_amd: the stand-alone FeedForward module is an inference op; dropout (and autograd) run "
                                      "through HealNet's fused training path -- call .eval() here")
        _reject_autograd_inputs("FeedForward", x)
        _require_gpu(x, "x")
        x = _f32c(x)
        if x.shape[-1] != self.dim:
            raise ValueError(f"last dim must be {self.dim}, got {tuple(x.shape)}")
        rows = x.numel() // self.dim
        lib = _capi.lib()
        p = self._params(norm)
        ws = _WS.get(x.device, lib.hn_ff_workspace_bytes(C.byref(p), rows))
        out = torch.empty_like(x)
        _capi.check(lib.hn_ff_fwd(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), rows, ws.data_ptr(), ws.numel(),
                                  _stream_ptr(x.device)), "hn_ff_fwd")
        return out

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


class _MeanPool(nn.Module):
    """Parameter-free stand-in for the reference's einops ``Reduce('b n d -> b d', 'mean')`` at
    ``to_logits.0`` (keeps the LayerNorm / Linear at indices 1 / 2)."""

    def forward(self, x):  # pragma: no cover - the fused head kernel is used instead
        raise RuntimeError("healnet_amd: to_logits runs inside hn_head_fwd")


class _Memo:
    """Weight-tying helper with the reference's ``cache_fn`` semantics (healnet.py:278-290): a factory
    result is stored / reused per key only when caching is requested for that call."""

    def __init__(self, factory: Callable[[], nn.Module]):
        self._factory = factory
        self._store: Dict[object, nn.Module] = {}

    def __call__(self, use_cache: bool, key=None) -> nn.Module:
        if not use_cache:
            return self._factory()
        if key not in self._store:
            self._store[key] = self._factory()
        return self._store[key]


# ------------------------------------------------------------------------------------------------
# autograd: one Function for the whole fusion stack (hn_fusion_forward_train / hn_fusion_backward)
# ------------------------------------------------------------------------------------------------
class _FusionFunction(torch.autograd.Function):
    """Differentiable w.r.t. every parameter of the model (incl. the latent array); the modality inputs get no
    gradient, as in the reference's training loop (healnet/main.py:432-465)."""

    @staticmethod
    def forward(ctx, module, inputs, held, mask_u8, b, skip_self, embeddings, stats_ptrs, x_ptrs, rng, *params):
        lib = _capi.lib()
        device = module.latents.device
        model, keep = module._descriptor(rng=rng)
        ctx.rng = rng
        masked = int(mask_u8 is not None)
        tape_bytes = lib.hn_fusion_tape_bytes(C.byref(model), inputs, b, masked, int(skip_self))
        need = lib.hn_fusion_workspace_bytes(C.byref(model), inputs, b)
        if tape_bytes == 0 or need == 0:
            _capi.check(-1, "hn_fusion_tape_bytes")
        tape = torch.empty(tape_bytes, dtype=torch.uint8, device=device)
        ws = _WS.get(device, need)
        out = torch.empty((b, module.l_c, module.l_d) if embeddings else (b, module.out_dims), dtype=torch.float32,
                          device=device)
        _capi.check(lib.hn_fusion_forward_train(C.byref(model), inputs, b, _ptr(mask_u8), int(skip_self), int(embeddings),
                                                out.data_ptr(), stats_ptrs, x_ptrs, tape.data_ptr(), tape.numel(), ws.data_ptr(),
                                                ws.numel(), _stream_ptr(device)), "hn_fusion_forward_train")
        ctx.module, ctx.inputs, ctx.held, ctx.mask_u8 = module, inputs, held, mask_u8
        ctx.b, ctx.skip_self, ctx.embeddings, ctx.tape, ctx.params = b, skip_self, embeddings, tape, params
        module._last_tape = (tape, masked, int(skip_self))      # forward() views the statistics / block inputs in place
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _capi.lib()
        module, params = ctx.module, ctx.params
        device = module.latents.device
        model, keep = module._descriptor(rng=ctx.rng)
        # healnet_amd.train.flatten_parameters(): every p.grad is a view of one flat buffer -> the kernels accumulate
        # straight into it (no per-parameter zero tensors, no AccumulateGrad add pass) and autograd gets None back.
        flat = getattr(module, "_hn_flat", None)
        direct = False
        if flat is not None:
            lo, hi = flat.grads.data_ptr(), flat.grads.data_ptr() + flat.grads.numel() * 4
            direct = all((not p.requires_grad) or (p.grad is not None and p.grad.is_contiguous()
                                                   and lo <= p.grad.data_ptr() < hi) for p in params)
        if direct:
            gmap = {id(p): p.grad for p in params if p.requires_grad}
        else:
            gmap = {id(p): torch.zeros_like(p, dtype=torch.float32) for p in params if p.requires_grad}
        grads, keep_g = module._grad_descriptor(gmap)
        masked = int(ctx.mask_u8 is not None)
        need = lib.hn_fusion_backward_workspace_bytes(C.byref(model), ctx.inputs, ctx.b, masked)
        if need == 0:
            _capi.check(-1, "hn_fusion_backward_workspace_bytes")
        ws = _WS.get(device, need)
        dout = dout.contiguous().float()
        _capi.check(lib.hn_fusion_backward(C.byref(model), ctx.inputs, ctx.b, _ptr(ctx.mask_u8), int(ctx.skip_self),
                                           int(ctx.embeddings), dout.data_ptr(), ctx.tape.data_ptr(), C.byref(grads),
                                           ws.data_ptr(), ws.numel(), _stream_ptr(device)), "hn_fusion_backward")
        if direct:
            return (None,) * (10 + len(params))
        return (None,) * 10 + tuple(gmap.get(id(p)) for p in params)


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
        self.core_precision = core_precision
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

    # -- C-ABI descriptor ------------------------------------------------------------------------
    def _descriptor(self, rng=None):
        """rng = (seed, offset) of a training forward / its backward: blocks then carry their dropout rates (Philox masks,
        include/healnet_hip.h hn_rng); None = inference descriptor (no dropout)."""
        M, depth = self.modalities, self.depth
        drop = (lambda mod: mod.dropout_p) if rng is not None else (lambda mod: 0.0)
        keep = []   # keep ctypes arrays alive for the duration of the call
        cross_attn = (_capi.AttnParams * (depth * M))()
        cross_ff = (_capi.FFParams * (depth * M))()
        self_attn = (_capi.AttnParams * depth)()
        self_ff = (_capi.FFParams * depth)()
        for layer in range(depth):
            mods = self.layers[layer]
            for m in range(M):
                blk, ffn = mods[2 * m], mods[2 * m + 1]
                cross_attn[layer * M + m] = blk.fn._params(blk.norm, blk.norm_context, drop(blk.fn))
                cross_ff[layer * M + m] = ffn.fn._params(ffn.norm, drop(ffn.fn))
            if self.self_per_cross_attn >= 1:
                blk, ffn = mods[2 * M][0], mods[2 * M][1]
                self_attn[layer] = blk.fn._params(blk.norm, None, drop(blk.fn))
                self_ff[layer] = ffn.fn._params(ffn.norm, drop(ffn.fn))
        cd = (C.c_int * M)(*[int(c) for c in self.input_channels])
        ax = (C.c_int * M)(*[int(a) for a in self.input_axes])
        model = _capi.Model(
            n_modalities=M, depth=depth, l_c=self.l_c, l_d=self.l_d, self_per_cross_attn=self.self_per_cross_attn,
            final_classifier_head=int(self.final_classifier_head), out_dims=self.out_dims,
            num_freq_bands=self.num_freq_bands, max_freq=float(self.max_freq),
            fourier_encode_data=int(self.fourier_encode_data), channel_dims=cd, num_spatial_axes=ax,
            latents=_ptr(self.latents), cross_attn=cross_attn, cross_ff=cross_ff, self_attn=self_attn, self_ff=self_ff,
            head_norm_w=_ptr(self.to_logits[1].weight) if self.final_classifier_head else None,
            head_norm_b=_ptr(self.to_logits[1].bias) if self.final_classifier_head else None,
            head_w=_ptr(self.to_logits[2].weight) if self.final_classifier_head else None,
            head_b=_ptr(self.to_logits[2].bias) if self.final_classifier_head else None,
            core_precision={"fp32": _capi.HN_CORE_F32, "bf16": _capi.HN_CORE_BF16, "bf16x3": _capi.HN_CORE_BF16X3}[self.core_precision],
            rng=_capi.Rng(seed=int(rng[0]) & 0xFFFFFFFFFFFFFFFF, offset=int(rng[1]) & 0xFFFFFFFF, stream=0) if rng is not None
            else _capi.Rng(0, 0, 0))
        keep.extend([cross_attn, cross_ff, self_attn, self_ff, cd, ax])
        return model, keep

    def _grad_descriptor(self, gmap):
        """hn_model_grads whose entries point into the per-parameter gradient buffers of ``gmap`` (id(param) -> tensor);
        tied parameters share one buffer, into which the backward accumulates."""
        M, depth = self.modalities, self.depth
        cross_attn = (_capi.AttnGrads * (depth * M))()
        cross_ff = (_capi.FFGrads * (depth * M))()
        self_attn = (_capi.AttnGrads * depth)()
        self_ff = (_capi.FFGrads * depth)()
        for layer in range(depth):
            mods = self.layers[layer]
            for m in range(M):
                blk, ffn = mods[2 * m], mods[2 * m + 1]
                cross_attn[layer * M + m] = blk.fn._grads(blk.norm, blk.norm_context, gmap)
                cross_ff[layer * M + m] = ffn.fn._grads(ffn.norm, gmap)
            if self.self_per_cross_attn >= 1:
                blk, ffn = mods[2 * M][0], mods[2 * M][1]
                self_attn[layer] = blk.fn._grads(blk.norm, None, gmap)
                self_ff[layer] = ffn.fn._grads(ffn.norm, gmap)
        gp = lambda t: None if id(t) not in gmap else gmap[id(t)].data_ptr()   # noqa: E731
        head = self.final_classifier_head
        grads = _capi.ModelGrads(latents=gp(self.latents), cross_attn=cross_attn, cross_ff=cross_ff, self_attn=self_attn,
                                 self_ff=self_ff, head_norm_w=gp(self.to_logits[1].weight) if head else None,
                                 head_norm_b=gp(self.to_logits[1].bias) if head else None,
                                 head_w=gp(self.to_logits[2].weight) if head else None,
                                 head_b=gp(self.to_logits[2].bias) if head else None)
        return grads, [cross_attn, cross_ff, self_attn, self_ff]

    def _check_mode(self) -> None:
        if self.self_per_cross_attn >= 2:
            raise ValueError("self_per_cross_attn >= 2 fails in the reference as well (healnet.py:242: "
                             "`self_attn, self_ff = layer[-1]`)")

    def _dropout_active(self) -> bool:
        """nn.Dropout semantics: masks are drawn in training mode only (healnet.py:381,421 attention, :347 feed-forward)."""
        return self.training and any(isinstance(mod, (Attention, FeedForward)) and mod.dropout_p > 0.0 for mod in self.modules())

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, tensors: List[Optional[torch.Tensor]], mask: Optional[torch.Tensor] = None,
                return_embeddings: bool = False, verbose: bool = False, _profile=None):
        # kernels are launched on the CURRENT device of the calling thread: make that the model's device for the call
        # (a model on cuda:1 called while cuda:0 is current would otherwise launch on the wrong GPU)
        if self.latents.is_cuda and torch.cuda.current_device() != self.latents.device.index:
            with torch.cuda.device(self.latents.device):
                return self._forward(tensors, mask, return_embeddings, verbose, _profile)
        return self._forward(tensors, mask, return_embeddings, verbose, _profile)

    def _forward(self, tensors, mask, return_embeddings, verbose, _profile):
        self._check_mode()
        M = self.modalities
        if len(tensors) > M:
            raise ValueError(f"{len(tensors)} tensors passed to a model with {M} modalities")
        missing_idx = [i for i in range(M) if i >= len(tensors) or tensors[i] is None]
        if verbose:
            print(f"Missing modalities indices: {[i for i in missing_idx if i < len(tensors)]}")
        _require_gpu(self.latents, "HealNet parameters")
        device = self.latents.device
        inputs = (_capi.ModalityInput * M)()
        held: List[Optional[torch.Tensor]] = [None] * M
        b = None
        for i in range(M):                    # the reference's sanity check (:206-208), all modalities first
            if i in missing_idx:
                continue
            assert tensors[i].dim() - 2 == self.input_axes[i], (f'input data for modality {i + 1} must hav'
                                                                f' the same number of axis as the input axis parameter')
        for i in range(M):
            if i in missing_idx:
                continue
            data = tensors[i]
            _require_gpu(data, f"modality {i + 1}")
            if data.device != device:
                raise RuntimeError(f"healnet_amd: modality {i + 1} lives on {data.device}, the model on {device}")
            bb, *axis, ch = data.shape
            if ch != self.input_channels[i]:
                raise ValueError(f"modality {i + 1}: expected {self.input_channels[i]} channels, got {ch} (the reference "
                                 "would silently skip every block of this modality, Appendix B-7)")
            if len(axis) > _capi.HN_MAX_AXES:
                raise NotImplementedError(f"at most {_capi.HN_MAX_AXES} spatial axes are supported")
            if b is None:
                b = bb
            elif bb != b:
                raise ValueError("batch dim must be identical across modalities")
            # bf16 tensors are read as they are, uint8 tensors as byte / 255 (8-bit image transport, == ToTensor);
            # anything else is staged as fp32
            x = data.contiguous() if data.dtype in (torch.bfloat16, torch.uint8) else _f32c(data)
            held[i] = x
            inputs[i].data = x.data_ptr()
            inputs[i].dtype = {torch.bfloat16: _capi.HN_BF16, torch.uint8: _capi.HN_U8}.get(x.dtype, _capi.HN_F32)
            for a, s in enumerate(axis):
                inputs[i].spatial[a] = int(s)
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

        lib = _capi.lib()
        model, keep = self._descriptor()
        need = lib.hn_fusion_workspace_bytes(C.byref(model), inputs, b)
        if need == 0:
            _capi.check(-1, "hn_fusion_workspace_bytes")
        ws = _WS.get(device, need)
        embeddings = return_embeddings or not self.final_classifier_head
        out = torch.empty((b, self.l_c, self.l_d) if embeddings else (b, self.out_dims), dtype=torch.float32, device=device)

        n_slots = self.depth * (M + 1)
        stats_ptrs = x_ptrs = None
        stats_t: List[Optional[torch.Tensor]] = [None] * n_slots
        trace_t: List[Optional[torch.Tensor]] = [None] * n_slots
        dropping = self._dropout_active()
        taping = dropping or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
        if self.keep_attention_stats and not taping:
            stats_ptrs = (C.c_void_p * n_slots)()
            x_ptrs = (C.c_void_p * n_slots)()
            for layer in range(self.depth):
                for j in range(M + 1):
                    live = (j < M and held[j] is not None) or (j == M and self.self_per_cross_attn > 0)
                    if not live:
                        continue
                    heads = self.layers[layer][2 * j].fn.heads if j < M else self.layers[layer][2 * M][0].fn.heads
                    stats_t[layer * (M + 1) + j] = torch.empty(b, heads, self.l_c, 2, dtype=torch.float32, device=device)
                    trace_t[layer * (M + 1) + j] = torch.empty(b, self.l_c, self.l_d, dtype=torch.float32, device=device)
                    stats_ptrs[layer * (M + 1) + j] = stats_t[layer * (M + 1) + j].data_ptr()
                    x_ptrs[layer * (M + 1) + j] = trace_t[layer * (M + 1) + j].data_ptr()

        if taping:
            # autograd path (train or eval mode alike, as in PyTorch): tape-recording forward + hn_fusion_backward.
            # Dropout masks are drawn in training mode only, grad mode or not (nn.Dropout); every forward advances the
            # Philox offset, the seed is torch's global seed (torch.manual_seed).
            if _profile is not None:
                raise ValueError("profiling hooks are only available under torch.no_grad() in eval mode")
            rng = None
            if dropping:
                self._rng_offset = (getattr(self, "_rng_offset", 0) + 1) & 0xFFFFFFFF
                rng = (torch.initial_seed(), self._rng_offset)
            self._last_rng = rng
            out = _FusionFunction.apply(self, inputs, held, mask_u8, b, bool(verbose), embeddings, None, None, rng,
                                        *list(self.parameters()))
            if self.keep_attention_stats:
                # the tape already holds every block's softmax statistics and input: view them, no copies
                tape, masked, skip = self._last_tape
                so, xo = (C.c_size_t * n_slots)(), (C.c_size_t * n_slots)()
                _capi.check(lib.hn_fusion_tape_layout(C.byref(model), inputs, b, masked, skip, so, xo), "hn_fusion_tape_layout")
                tf = tape.view(torch.float32)
                for slot in range(n_slots):
                    if so[slot] == C.c_size_t(-1).value:
                        continue
                    j = slot % (M + 1)
                    heads = self.layers[slot // (M + 1)][2 * j].fn.heads if j < M else self.layers[slot // (M + 1)][2 * M][0].fn.heads
                    stats_t[slot] = tf[so[slot]:so[slot] + b * heads * self.l_c * 2].view(b, heads, self.l_c, 2)
                    trace_t[slot] = tf[xo[slot]:xo[slot] + b * self.l_c * self.l_d].view(b, self.l_c, self.l_d)
            self._last_tape = None
        else:
            _capi.check(lib.hn_fusion_forward(C.byref(model), inputs, b, _ptr(mask_u8), int(bool(verbose)), int(embeddings),
                                              out.data_ptr(), stats_ptrs, x_ptrs, ws.data_ptr(), ws.numel(),
                                              _stream_ptr(device), _profile), "hn_fusion_forward")
        if verbose:
            for layer in range(self.depth):
                for i in missing_idx:
                    print(f"Skipping update in fusion layer {layer + 1} for missing modality {i + 1}")
        self._last = dict(inputs=held, mask=mask_u8, stats=stats_t, trace=trace_t, b=b, skipped_self=bool(verbose) and bool(missing_idx))
        self._bind_lazy_probs()
        if self.compat_mutate_inputs:
            for i in range(min(M, len(tensors))):
                if held[i] is not None:
                    tensors[i] = fourier_encode_concat(_f32c(held[i]), self.num_freq_bands, self.max_freq, self.fourier_encode_data)
        return out

    # -- attention weights on demand ---------------------------------------------------------------
    def _bind_lazy_probs(self) -> None:
        last = self._last
        M = self.modalities
        for mod in self.modules():
            if isinstance(mod, Attention):
                mod._probs_fn = None
        if last is None or not self.keep_attention_stats:
            return
        lib = _capi.lib()
        zcache: Dict[int, tuple] = {}
        for layer in range(self.depth):
            for j in range(M + 1):
                slot = layer * (M + 1) + j
                if last["stats"][slot] is None:
                    continue
                blk = self.layers[layer][2 * j] if j < M else self.layers[layer][2 * M][0]
                att: Attention = blk.fn

                def probs(reduced=False, layer=layer, j=j, slot=slot, blk=blk, att=att):
                    b, L = last["b"], self.l_c
                    xin, stats = last["trace"][slot], last["stats"][slot]
                    dev = xin.device
                    if j < M:
                        if j not in zcache:
                            data = _f32c(last["inputs"][j])
                            n = data.numel() // (b * data.shape[-1])
                            d = self.context_dims[j]
                            ld = lib.hn_context_pitch(d, att.dim_head)
                            z = torch.empty(b, n, ld, dtype=torch.float32, device=dev)
                            sp = (C.c_int * len(data.shape[1:-1]))(*data.shape[1:-1])
                            _capi.check(lib.hn_encode_norm(data.data_ptr(), b, len(data.shape) - 2, sp, data.shape[-1],
                                                           self.num_freq_bands, float(self.max_freq),
                                                           int(self.fourier_encode_data), 1e-5, z.data_ptr(), ld,
                                                           _stream_ptr(dev)), "hn_encode_norm")
                            zcache[j] = (z, ld, n, d)
                        z, ld, n, d = zcache[j]
                        p = att._params(blk.norm, blk.norm_context)
                        msk = last["mask"]
                    else:
                        z, ld, n, d = None, 0, L, self.l_d
                        p = att._params(blk.norm, None)
                        msk = None
                    need = lib.hn_attn_workspace_bytes(C.byref(p), int(z is not None), ld, b, L, n, d)
                    aux = _WS_AUX.get(dev, need)
                    pr = torch.empty((b * att.heads, n) if reduced else (b * att.heads, L, n), dtype=torch.float32, device=dev)
                    fn = lib.hn_attn_importance if reduced else lib.hn_attn_probs
                    _capi.check(fn(C.byref(p), xin.data_ptr(), _ptr(z), ld, b, L, n, d, _ptr(msk), stats.data_ptr(),
                                   pr.data_ptr(), aux.data_ptr(), aux.numel(), _stream_ptr(dev)),
                                "hn_attn_importance" if reduced else "hn_attn_probs")
                    return pr

                att._probs_fn = probs

    def get_attention_weights(self) -> List[Optional[torch.Tensor]]:
        """Every ``Attention.attn_weights`` in ``self.modules()`` order (healnet.py:252-262): per layer
        [cross_0 .. cross_{M-1}, self]; shape (b*heads, l_c, N) each, ``None`` for blocks that did not run."""
        return [mod.attn_weights for mod in self.modules() if isinstance(mod, Attention)]

    def get_attention_importance(self) -> List[Optional[torch.Tensor]]:
        """``[w.mean(dim=1) for w in get_attention_weights()]`` -- shape (b*heads, N) each -- without ever forming the
        (b*heads, l_c, N) matrices: the reduction the reference's explainer applies to every entry (explainer.py:161-164,
        :209-211).  Same order and ``None`` convention as ``get_attention_weights``."""
        return [mod.attn_importance for mod in self.modules() if isinstance(mod, Attention)]
