"""``torch.ops.healnet_hip.*`` -- the C-ABI entry points registered as PyTorch operators.  This is THE route of the package:
``HealNet.forward`` / ``Attention.forward`` / ``FeedForward.forward`` dispatch through these ops (healnet/models/healnet.py:190-195,
:400, :349 of the reference are the call sites they serve), autograd is attached with ``torch.library.register_autograd``, shape
inference with ``torch.library.register_fake`` (so ``torch.compile`` traces a model into one opaque node per call).

CUDA/HIP dispatch key only: there is no CPU kernel, calling an op with CPU tensors raises NotImplementedError from the dispatcher.

Whole fusion stack (``spec`` = JSON description of the module structure, see ``HealNet._spec``; ``params`` = ``list(model.parameters())``):
    fusion_forward(tensors, mask?, params, spec, skip_self, embeddings, keep_stats) -> (out, stats, trace)       hn_fusion_forward
    fusion_forward_train(tensors, mask?, params, spec, skip_self, embeddings, rng?, grad_buffer?, grad_offsets)
                                                                                   -> (out, tape, layout)       hn_fusion_forward_train
    fusion_backward(dout, tape, tensors, mask?, params, spec, skip_self, embeddings, rng?, needs_grad) -> Tensor[]   hn_fusion_backward
    fusion_backward_into(dout, tape, tensors, mask?, params, spec, skip_self, embeddings, rng?, grad_buffer!, grad_offsets) -> ()
                    (the same backward ACCUMULATING into a flat gradient buffer: healnet_amd.train.FlatParameters)
Blocks (differentiable w.r.t. x and the parameters; no gradient flows to the context, as in the C ABI):
    attention(x, context?, mask?, norm_w?, norm_b?, ctx_gamma?, ctx_beta?, w_q, w_kv, w_out, b_out, heads, residual) -> (b, L, query_dim)
    attention_fwd(..., train) -> (out, stats, saved)       attention_bwd(dy, x, out, ..., stats, saved) -> Tensor[9]
    latent_block(x, a_norm_w?, a_norm_b?, w_q, w_kv, w_out, b_out, heads, f_norm_w?, f_norm_b?, w1, b1, w2, b2, gelu) -> like x
                 (latent self-attention + feed-forward of one fusion iteration, hn_latent_block_fwd: the fused latent chain)
    latent_block_fwd(..., train) -> (out, x_mid, stats, saved)      latent_block_bwd(dy, x_mid, stats, saved, ...) -> Tensor[13]
    feed_forward(x, norm_w?, norm_b?, w1, b1, w2, b2, gelu, residual) -> like x        feed_forward_bwd(...) -> Tensor[7]
    head(x, norm_w, norm_b, w, bias) -> (b, out_dims)                                   head_bwd(...) -> Tensor[5]
Elementwise / encode helpers (forward only):
    fourier_encode_concat(data, num_freq_bands, max_freq, fourier_encode_data) -> (b, N, D)
    encode_norm(data, num_freq_bands, max_freq, fourier_encode_data, pitch) -> (b, N, pitch)
    temperature_softmax(logits, temperature) -> like logits
"""
from __future__ import annotations

import ctypes as C
import functools
import json
import os
import threading
from typing import Dict, List, Optional, Sequence

import torch

from . import _capi
from ._rt import WS, f32c as _f32c, ptr as _ptr, stream_ptr as _stream_ptr

_lib = torch.library.Library("healnet_hip", "DEF")
_lib.define("fourier_encode_concat(Tensor data, int num_freq_bands, float max_freq, bool fourier_encode_data) -> Tensor")
_lib.define("encode_norm(Tensor data, int num_freq_bands, float max_freq, bool fourier_encode_data, int pitch) -> Tensor")
_lib.define("encode_norm_slab(Tensor data, int num_freq_bands, float max_freq, bool fourier_encode_data, int pitch, int axis0_begin, "
            "int axis0_total) -> Tensor")
_lib.define("attention_partial(Tensor x, Tensor context, Tensor? mask, Tensor? norm_w, Tensor? norm_b, Tensor? ctx_gamma, "
            "Tensor? ctx_beta, Tensor w_q, Tensor w_kv, Tensor w_out, Tensor b_out, int heads) -> (Tensor, Tensor)")
_lib.define("attention_merge(Tensor x, Tensor o_parts, Tensor stats_parts, Tensor w_q, Tensor w_out, Tensor b_out, int heads, "
            "bool residual) -> (Tensor, Tensor)")
_ATTN_ARGS = ("Tensor x, Tensor? context, Tensor? mask, Tensor? norm_w, Tensor? norm_b, Tensor? ctx_gamma, Tensor? ctx_beta, "
              "Tensor w_q, Tensor w_kv, Tensor w_out, Tensor b_out, int heads, bool residual")
_lib.define(f"attention({_ATTN_ARGS}) -> Tensor")
# dropout / rng (trailing, defaulted): nn.Dropout of the block in TRAINING calls -- rng = [seed, offset, stream] of hn_rng; the
# backward replays the forward's masks from the same triple (stand-alone modules: VERDICT r5 item 4)
_DROP = "float dropout=0.0, int[] rng=[]"
_lib.define(f"attention_fwd({_ATTN_ARGS}, bool train, {_DROP}) -> (Tensor, Tensor, Tensor)")
_lib.define("attention_bwd(Tensor dy, Tensor x, Tensor out, Tensor? context, Tensor? mask, Tensor? norm_w, Tensor? norm_b, "
            "Tensor? ctx_gamma, Tensor? ctx_beta, Tensor w_q, Tensor w_kv, Tensor w_out, Tensor b_out, int heads, bool residual, "
            "Tensor stats, Tensor saved, " + _DROP + ") -> Tensor[]")
_FF_ARGS = "Tensor x, Tensor? norm_w, Tensor? norm_b, Tensor w1, Tensor b1, Tensor w2, Tensor b2, bool gelu, bool residual"
_lib.define(f"feed_forward({_FF_ARGS}, {_DROP}) -> Tensor")
_lib.define(f"feed_forward_bwd(Tensor dy, {_FF_ARGS}, {_DROP}) -> Tensor[]")
_LB_ARGS = ("Tensor x, Tensor? a_norm_w, Tensor? a_norm_b, Tensor w_q, Tensor w_kv, Tensor w_out, Tensor b_out, int heads, "
            "Tensor? f_norm_w, Tensor? f_norm_b, Tensor w1, Tensor b1, Tensor w2, Tensor b2, bool gelu")
_lib.define(f"latent_block({_LB_ARGS}) -> Tensor")
_LB_DROP = "float attn_dropout=0.0, float ff_dropout=0.0, int[] rng=[]"      # rng.stream: the attention block's, the feed-forward block takes stream + 1
_lib.define(f"latent_block_fwd({_LB_ARGS}, bool train, {_LB_DROP}) -> (Tensor, Tensor, Tensor, Tensor)")
_lib.define(f"latent_block_bwd(Tensor dy, Tensor x_mid, Tensor stats, Tensor saved, {_LB_ARGS}, {_LB_DROP}) -> Tensor[]")
_lib.define("head(Tensor x, Tensor norm_w, Tensor norm_b, Tensor w, Tensor bias) -> Tensor")
_lib.define("head_bwd(Tensor dlogits, Tensor x, Tensor norm_w, Tensor norm_b, Tensor w) -> Tensor[]")
_lib.define("temperature_softmax(Tensor logits, float temperature) -> Tensor")
_lib.define("fusion_forward(Tensor?[] tensors, Tensor? mask, Tensor[] params, str spec, int skip_self, bool embeddings, "
            "bool keep_stats) -> (Tensor, Tensor, Tensor)")
_lib.define("fusion_forward_train(Tensor?[] tensors, Tensor? mask, Tensor[] params, str spec, int skip_self, bool embeddings, "
            "Tensor? rng, Tensor? grad_buffer, int[] grad_offsets) -> (Tensor, Tensor, Tensor)")
_lib.define("fusion_backward(Tensor dout, Tensor tape, Tensor?[] tensors, Tensor? mask, Tensor[] params, str spec, int skip_self, "
            "bool embeddings, Tensor? rng, bool[] needs_grad) -> Tensor[]")
_lib.define("fusion_backward_into(Tensor dout, Tensor tape, Tensor?[] tensors, Tensor? mask, Tensor[] params, str spec, "
            "int skip_self, bool embeddings, Tensor? rng, Tensor(a!) grad_buffer, int[] grad_offsets) -> ()")

_NO_HOST_CACHE = os.environ.get("HN_NO_HOST_CACHE", "0") == "1"      # development switch: size / layout queries on every call
_FAKE_PTR = 256          # stands in for device addresses while tracing with FakeTensors (size queries only, nothing is launched)


# ------------------------------------------------------------------------------------------------
# model structure <-> C descriptors
# ------------------------------------------------------------------------------------------------
def _rng_dev(rng: torch.Tensor):
    """The generator state travels as a CPU int64 tensor [seed, offset] or [seed, offset, address of a device uint32 word]: the
    third entry is hn_rng.offset_dev (healnet_amd.train.GraphedStep keeps the word alive and advances it between replays)."""
    return (int(rng[2]) or None) if rng.numel() > 2 else None


class Spec:
    """Parsed ``spec`` string of the fusion ops: the structure of a HealNet (sizes, per-block head counts / dropout rates, and
    for every pointer field of hn_model the index of the tensor in ``params`` that backs it -- tied blocks simply repeat
    indices).  Builds the ctypes descriptors of include/healnet_hip.h from a parameter list."""

    ATTN_FIELDS = ("norm_w", "norm_b", "ctx_gamma", "ctx_beta", "w_q", "w_kv", "w_out", "b_out")
    FF_FIELDS = ("norm_w", "norm_b", "w1", "b1", "w2", "b2")
    HEAD_FIELDS = ("head_norm_w", "head_norm_b", "head_w", "head_b")

    def __init__(self, text: str):
        d = json.loads(text)
        self.d = d
        self.M, self.depth, self.l_c, self.l_d = d["M"], d["depth"], d["l_c"], d["l_d"]
        self.spca = d["self_per_cross_attn"]
        self.head = bool(d["head"])
        self.out_dims = d["out_dims"]
        self.n_slots = self.depth * (self.M + 1)
        self.max_heads = max([a["heads"] for a in d["cross_attn"]] + [a["heads"] for a in d["self_attn"]] + [1])

    def slot_heads(self, slot: int) -> int:
        layer, j = divmod(slot, self.M + 1)
        return self.d["cross_attn"][layer * self.M + j]["heads"] if j < self.M else self.d["self_attn"][layer]["heads"]

    def model(self, params: Optional[Sequence[torch.Tensor]], rng: Optional[torch.Tensor] = None):
        """(hn_model, keep-alive list).  ``params`` None -> placeholder addresses (fake-tensor tracing).  ``rng`` = CPU int64
        tensor (seed, offset) of a training forward / its backward: blocks then carry their dropout rates."""
        d = self.d
        if params is None:
            P = lambda i: None if i is None or i < 0 else _FAKE_PTR      # noqa: E731
        else:
            P = lambda i: None if i is None or i < 0 else _ptr(params[i])  # noqa: E731
        dropping = rng is not None
        M, depth = self.M, self.depth
        ca = (_capi.AttnParams * max(1, depth * M))()
        cf = (_capi.FFParams * max(1, depth * M))()
        sa = (_capi.AttnParams * depth)()
        sf = (_capi.FFParams * depth)()

        def attn(dst, a):
            dst.heads, dst.dim_head, dst.query_dim = a["heads"], a["dim_head"], a["query_dim"]
            for f, i in zip(self.ATTN_FIELDS, a["p"]):
                setattr(dst, f, P(i))
            dst.dropout = float(a["dropout"]) if dropping else 0.0

        def ff(dst, a):
            dst.dim, dst.gate = a["dim"], a["gate"]
            for f, i in zip(self.FF_FIELDS, a["p"]):
                setattr(dst, f, P(i))
            dst.dropout = float(a["dropout"]) if dropping else 0.0

        for k, a in enumerate(d["cross_attn"]):
            attn(ca[k], a)
        for k, a in enumerate(d["cross_ff"]):
            ff(cf[k], a)
        for k, a in enumerate(d["self_attn"]):
            attn(sa[k], a)
        for k, a in enumerate(d["self_ff"]):
            ff(sf[k], a)
        cd = (C.c_int * M)(*d["channels"])
        ax = (C.c_int * M)(*d["axes"])
        hp = d["head_p"] if self.head else [None] * 4
        if rng is not None:
            seed, offset = int(rng[0]) & 0xFFFFFFFFFFFFFFFF, int(rng[1]) & 0xFFFFFFFF
            r = _capi.Rng(seed=seed, offset=offset, stream=0, offset_dev=_rng_dev(rng))
        else:
            r = _capi.Rng(0, 0, 0, None)
        model = _capi.Model(
            n_modalities=M, depth=depth, l_c=self.l_c, l_d=self.l_d, self_per_cross_attn=self.spca,
            final_classifier_head=int(self.head), out_dims=self.out_dims, num_freq_bands=d["num_freq_bands"],
            max_freq=float(d["max_freq"]), fourier_encode_data=int(d["fourier"]), channel_dims=cd, num_spatial_axes=ax,
            latents=P(d["latents"]), cross_attn=ca, cross_ff=cf, self_attn=sa, self_ff=sf,
            head_norm_w=P(hp[0]), head_norm_b=P(hp[1]), head_w=P(hp[2]), head_b=P(hp[3]),
            core_precision=d["core_precision"], rng=r)
        return model, [ca, cf, sa, sf, cd, ax]

    def model_cached(self, params: Sequence[torch.Tensor]):
        """Inference descriptor of ``params``, memoised on the parameters' addresses: rebuilding the ~125 pointer fields through
        ctypes costs a large part of the ~0.8 ms a forward spends on the host, which is what bounds the path at b <= 4 (VERDICT
        r2).  The descriptor only holds addresses, so it stays valid exactly as long as the key (a re-homed parameter changes
        the key; a handful of entries are kept: the fp32 model, its flat-parameter twin, ...)."""
        key = self._param_key(params)
        cache = self._thread_cache("_model_cache")
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 8:
                cache.clear()
            for p in params:
                _ptr(p)                                  # the dtype / contiguity checks of the uncached route
            hit = cache[key] = self.model(params)
        return hit

    @staticmethod
    def _param_key(params: Sequence[torch.Tensor]):
        """Cache key of a parameter list: addresses AND what the kernels assume about them (fp32, contiguous, one device) -- a
        parameter re-typed or re-strided in place at an unchanged address must miss."""
        dev = params[0].device.index if len(params) else -1
        return (dev,) + tuple((p.data_ptr(), p.dtype is torch.float32 and p.is_contiguous()) for p in params)

    def _thread_cache(self, name: str) -> dict:
        """Per-thread memo: the cached descriptor of a training forward is patched in place (generator state), so two threads
        sharing one Spec must not share the struct."""
        tls = self.__dict__.get("_tls")
        if tls is None:
            tls = self.__dict__.setdefault("_tls", threading.local())
        d = tls.__dict__.get(name)
        if d is None:
            d = tls.__dict__[name] = {}
        return d

    def model_train_cached(self, params: Sequence[torch.Tensor], rng: Optional[torch.Tensor]):
        """Descriptor of a training forward / its backward, memoised like ``model_cached``: with dropout the blocks carry their
        rates and only the generator state (seed, per-forward offset) changes from call to call -- it is patched into the cached
        struct.  (At the reference's tuned TCGA shapes a training step is ~90 launches of 5-50 us: rebuilding ~125 pointer fields
        through ctypes twice per step was a fifth of the host time that bounds it.)"""
        if rng is None:
            return self.model_cached(params)
        key = ("drop",) + self._param_key(params)
        cache = self._thread_cache("_model_cache")
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 8:
                cache.clear()
            hit = cache[key] = self.model(params, rng)
        model = hit[0]
        model.rng.seed, model.rng.offset = int(rng[0]) & 0xFFFFFFFFFFFFFFFF, int(rng[1]) & 0xFFFFFFFF
        model.rng.offset_dev = _rng_dev(rng)
        return hit

    def grads_cached(self, gptr: Sequence[Optional[int]]):
        key = tuple(gptr)
        cache = self._thread_cache("_grads_cache")
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 8:
                cache.clear()
            hit = cache[key] = self.grads(gptr)
        return hit

    def sizes_cached(self, kind: str, key, compute):
        """Workspace / tape sizes and the tape layout of one call shape (they depend on the structure, the batch, the input shapes
        and whether blocks drop -- not on addresses): each is a planning pass inside the library."""
        if _NO_HOST_CACHE:
            return compute()
        cache = self.__dict__.setdefault("_size_cache", {})
        k = (kind, key)
        hit = cache.get(k)
        if hit is None:
            if len(cache) >= 64:
                cache.clear()
            hit = cache[k] = compute()
        return hit

    def grads(self, gptr: Sequence[Optional[int]]):
        """hn_model_grads whose entries are the addresses ``gptr[param index]`` (None = no gradient wanted)."""
        d = self.d
        G = lambda i: None if i is None or i < 0 else gptr[i]      # noqa: E731
        M, depth = self.M, self.depth
        ca = (_capi.AttnGrads * max(1, depth * M))()
        cf = (_capi.FFGrads * max(1, depth * M))()
        sa = (_capi.AttnGrads * depth)()
        sf = (_capi.FFGrads * depth)()
        for arr, blocks, fields in ((ca, d["cross_attn"], self.ATTN_FIELDS), (cf, d["cross_ff"], self.FF_FIELDS),
                                    (sa, d["self_attn"], self.ATTN_FIELDS), (sf, d["self_ff"], self.FF_FIELDS)):
            for k, a in enumerate(blocks):
                for f, i in zip(fields, a["p"]):
                    setattr(arr[k], f, G(i))
        hp = d["head_p"] if self.head else [None] * 4
        g = _capi.ModelGrads(latents=G(d["latents"]), cross_attn=ca, cross_ff=cf, self_attn=sa, self_ff=sf,
                             head_norm_w=G(hp[0]), head_norm_b=G(hp[1]), head_w=G(hp[2]), head_b=G(hp[3]))
        return g, [ca, cf, sa, sf]

    def inputs(self, tensors: Sequence[Optional[torch.Tensor]], fake: bool = False):
        """(hn_modality_input[M], held tensors, b).  Entries beyond a shorter list / None = missing modality."""
        M = self.M
        inp = (_capi.ModalityInput * M)()
        held: List[Optional[torch.Tensor]] = [None] * M
        b = None
        for i in range(min(M, len(tensors))):
            t = tensors[i]
            if t is None:
                continue
            if not fake:
                t = t.contiguous() if t.dtype in (torch.bfloat16, torch.uint8) else _f32c(t)
            held[i] = t
            inp[i].data = _FAKE_PTR if fake else t.data_ptr()
            inp[i].dtype = {torch.bfloat16: _capi.HN_BF16, torch.uint8: _capi.HN_U8}.get(t.dtype, _capi.HN_F32)
            axes = list(t.shape[1:-1])
            if len(axes) > _capi.HN_MAX_AXES:
                raise NotImplementedError(f"at most {_capi.HN_MAX_AXES} spatial axes are supported")
            for a, s in enumerate(axes):
                inp[i].spatial[a] = int(s)
            b = int(t.shape[0]) if b is None else b
        if b is None:
            raise ValueError("at least one modality must be present")
        return inp, held, b


@functools.lru_cache(maxsize=64)
def spec_of(text: str) -> Spec:
    return Spec(text)


# Side channels of the fusion ops (host-only objects that cannot travel through an operator schema):
#   profile hook  : bench.py's hipEvent pairs around the dominant kernel (hn_profile), consumed by the next fusion_forward call
#   backward hooks: gradient-readiness callbacks of the overlapped data-parallel all-reduce (healnet_amd.dist), keyed by the
#                   address of the flat gradient buffer the backward accumulates into
_tls = threading.local()
_BACKWARD_HOOKS: Dict[int, object] = {}


def set_profile(profile) -> None:
    """``profile``: ctypes pointer to a _capi.Profile (or None) handed to the NEXT hn_fusion_forward on this thread."""
    _tls.profile = profile


def register_backward_hook(grad_buffer: torch.Tensor, hook) -> None:
    """``hook.ready`` must be a _capi.GradReady (kept alive by the hook); ``hook.begin()`` / ``hook.end()`` bracket the call."""
    _BACKWARD_HOOKS[grad_buffer.data_ptr()] = hook


def unregister_backward_hook(grad_buffer: torch.Tensor) -> None:
    _BACKWARD_HOOKS.pop(grad_buffer.data_ptr(), None)


# ------------------------------------------------------------------------------------------------
# encode / elementwise helpers
# ------------------------------------------------------------------------------------------------
def _spatial(x):
    spatial = list(x.shape[1:-1])
    if not 1 <= len(spatial) <= _capi.HN_MAX_AXES:
        raise ValueError(f"1..{_capi.HN_MAX_AXES} spatial axes supported, got {len(spatial)}")
    n = 1
    for s in spatial:
        n *= s
    return spatial, n


def _fourier_encode_concat(data, num_freq_bands, max_freq, fourier_encode_data):
    x = _f32c(data)
    spatial, n = _spatial(x)
    b, ch = x.shape[0], x.shape[-1]
    d = ch + (len(spatial) * (2 * num_freq_bands + 1) if fourier_encode_data else 0)
    out = torch.empty(b, n, d, dtype=torch.float32, device=x.device)
    sp = (C.c_int * len(spatial))(*spatial)
    _capi.check(_capi.lib().hn_fourier_encode_concat(x.data_ptr(), b, len(spatial), sp, ch, num_freq_bands, float(max_freq),
                                                     int(fourier_encode_data), out.data_ptr(), d, _stream_ptr(x.device)),
                "hn_fourier_encode_concat")
    return out


@torch.library.register_fake("healnet_hip::fourier_encode_concat")
def _(data, num_freq_bands, max_freq, fourier_encode_data):
    spatial, n = _spatial(data)
    d = data.shape[-1] + (len(spatial) * (2 * num_freq_bands + 1) if fourier_encode_data else 0)
    return data.new_empty((data.shape[0], n, d), dtype=torch.float32)


def _encode_norm(data, num_freq_bands, max_freq, fourier_encode_data, pitch):
    x = _f32c(data)
    spatial, n = _spatial(x)
    b, ch = x.shape[0], x.shape[-1]
    z = torch.empty(b, n, pitch, dtype=torch.float32, device=x.device)
    sp = (C.c_int * len(spatial))(*spatial)
    _capi.check(_capi.lib().hn_encode_norm(x.data_ptr(), b, len(spatial), sp, ch, num_freq_bands, float(max_freq),
                                           int(fourier_encode_data), 1e-5, z.data_ptr(), pitch, _stream_ptr(x.device)),
                "hn_encode_norm")
    return z


@torch.library.register_fake("healnet_hip::encode_norm")
def _(data, num_freq_bands, max_freq, fourier_encode_data, pitch):
    _, n = _spatial(data)
    return data.new_empty((data.shape[0], n, pitch), dtype=torch.float32)


def _encode_norm_slab(data, num_freq_bands, max_freq, fourier_encode_data, pitch, axis0_begin, axis0_total):
    """hn_encode_norm_slab: rows [axis0_begin, axis0_begin + data.shape[1]) of a modality whose first spatial axis has axis0_total
    positions (context split over ranks): the tokens get the positional features they have in the whole tensor."""
    x = _f32c(data)
    spatial, n = _spatial(x)
    b, ch = x.shape[0], x.shape[-1]
    z = torch.empty(b, n, pitch, dtype=torch.float32, device=x.device)
    sp = (C.c_int * len(spatial))(*spatial)
    _capi.check(_capi.lib().hn_encode_norm_slab(x.data_ptr(), b, len(spatial), sp, ch, num_freq_bands, float(max_freq),
                                                int(fourier_encode_data), 1e-5, z.data_ptr(), pitch, int(axis0_begin),
                                                int(axis0_total), _stream_ptr(x.device)), "hn_encode_norm_slab")
    return z


@torch.library.register_fake("healnet_hip::encode_norm_slab")
def _(data, num_freq_bands, max_freq, fourier_encode_data, pitch, axis0_begin, axis0_total):
    _, n = _spatial(data)
    return data.new_empty((data.shape[0], n, pitch), dtype=torch.float32)


def _temperature_softmax(logits, temperature):
    x = _f32c(logits)
    y = torch.empty_like(x)
    n = x.shape[-1]
    _capi.check(_capi.lib().hn_temperature_softmax(x.data_ptr(), y.data_ptr(), x.numel() // n, n, float(temperature),
                                                   _stream_ptr(x.device)), "hn_temperature_softmax")
    return y


@torch.library.register_fake("healnet_hip::temperature_softmax")
def _(logits, temperature):
    return torch.empty_like(logits, dtype=torch.float32, memory_format=torch.contiguous_format)


# ------------------------------------------------------------------------------------------------
# attention block
# ------------------------------------------------------------------------------------------------
def _attn_params(x, context, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads, fake=False):
    b, L, qd = x.shape
    inner = w_q.shape[0]
    if context is None:
        N, D, ld = L, qd, 0
    else:
        N, D, ld = context.shape[1], w_kv.shape[1], context.shape[2]
    pf = (lambda t: None if t is None else _FAKE_PTR) if fake else _ptr
    p = _capi.AttnParams(heads=heads, dim_head=inner // heads, query_dim=qd, norm_w=pf(norm_w), norm_b=pf(norm_b),
                         ctx_gamma=pf(ctx_gamma), ctx_beta=pf(ctx_beta), w_q=pf(w_q), w_kv=pf(w_kv), w_out=pf(w_out),
                         b_out=pf(b_out))
    return p, (int(b), int(L), int(N), int(D), int(ld))


def _set_drop(p, dropout, rng, stream_add=0):
    """hn_attn_params / hn_ff_params .dropout + .rng from the ops' trailing arguments (rng = [seed, offset, stream])."""
    if dropout and dropout > 0.0:
        if len(rng) != 3:
            raise ValueError("healnet_hip: dropout > 0 needs rng = [seed, offset, stream]")
        p.dropout = float(dropout)
        p.rng = _capi.Rng(seed=int(rng[0]) & 0xFFFFFFFFFFFFFFFF, offset=int(rng[1]) & 0xFFFFFFFF, stream=(int(rng[2]) + stream_add) & 0x7FFFFFFF,
                          offset_dev=None)
    return p


def _mask_u8(mask, b):
    return None if mask is None else mask.reshape(b, -1).to(torch.uint8).contiguous()


def _attention_fwd(x, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads, residual, train, dropout=0.0, rng=()):
    lib = _capi.lib()
    x = _f32c(x)
    ctx = None if context is None else _f32c(context)
    p, (b, L, N, D, ld) = _attn_params(x, ctx, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads)
    if train:
        _set_drop(p, dropout, rng)
    m = _mask_u8(mask, b)
    has_ctx = int(ctx is not None)
    need = lib.hn_attn_workspace_bytes(C.byref(p), has_ctx, ld, b, L, N, D)
    if need == 0:
        _capi.check(-1, "hn_attn_workspace_bytes")
    ws = WS.get(x.device, need)
    out = torch.empty_like(x)
    stats = torch.empty(b, heads, L, 2, dtype=torch.float32, device=x.device)
    if train:
        saved = torch.empty(lib.hn_attn_saved_floats(C.byref(p), has_ctx, ld, b, L, N, D, int(m is not None)),
                            dtype=torch.float32, device=x.device)
        _capi.check(lib.hn_attn_fwd_train(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), _ptr(ctx), ld, b, L, N, D,
                                          _ptr(m), stats.data_ptr(), saved.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _stream_ptr(x.device)), "hn_attn_fwd_train")
    else:
        saved = torch.empty(0, dtype=torch.float32, device=x.device)
        _capi.check(lib.hn_attn_fwd(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), _ptr(ctx), ld, b, L, N, D, _ptr(m),
                                    stats.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(x.device)), "hn_attn_fwd")
    return out, stats, saved


@torch.library.register_fake("healnet_hip::attention_fwd")
def _(x, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads, residual, train, dropout=0.0, rng=()):
    p, (b, L, N, D, ld) = _attn_params(x, context, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads, fake=True)
    if train and dropout > 0.0:
        p.dropout = float(dropout)
    n_saved = 0
    if train:
        n_saved = _capi.lib().hn_attn_saved_floats(C.byref(p), int(context is not None), ld, b, L, N, D, int(mask is not None))
    return (x.new_empty(x.shape, dtype=torch.float32), x.new_empty((b, heads, L, 2), dtype=torch.float32),
            x.new_empty((n_saved,), dtype=torch.float32))


def _attention_partial(x, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads):
    """hn_attn_partial_fwd: the block of healnet.py:400-424 over THIS rank's tokens -> (normalised output of the shard
    (b, L, inner), softmax statistics (b, heads, L, 2)).  Inference only."""
    lib = _capi.lib()
    x = _f32c(x)
    ctx = _f32c(context)
    p, (b, L, N, D, ld) = _attn_params(x, ctx, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads)
    m = _mask_u8(mask, b)
    need = lib.hn_attn_workspace_bytes(C.byref(p), 1, ld, b, L, N, D)
    if need == 0:
        _capi.check(-1, "hn_attn_workspace_bytes")
    ws = WS.get(x.device, need)
    o = torch.empty(b, L, w_q.shape[0], dtype=torch.float32, device=x.device)
    stats = torch.empty(b, heads, L, 2, dtype=torch.float32, device=x.device)
    _capi.check(lib.hn_attn_partial_fwd(C.byref(p), x.data_ptr(), ctx.data_ptr(), ld, b, L, N, D, _ptr(m), o.data_ptr(),
                                        stats.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(x.device)), "hn_attn_partial_fwd")
    return o, stats


@torch.library.register_fake("healnet_hip::attention_partial")
def _(x, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads):
    b, L = x.shape[0], x.shape[1]
    return x.new_empty((b, L, w_q.shape[0]), dtype=torch.float32), x.new_empty((b, heads, L, 2), dtype=torch.float32)


def _attention_merge(x, o_parts, stats_parts, w_q, w_out, b_out, heads, residual):
    """hn_attn_merge_fwd: fold the shards' (output, statistics) pairs -- (G, b, L, inner), (G, b, heads, L, 2) -- and finish the
    block (healnet.py:425-426 + the residual).  Returns (x_out, merged statistics)."""
    lib = _capi.lib()
    x = _f32c(x)
    o_parts, stats_parts = _f32c(o_parts), _f32c(stats_parts)
    b, L, qd = x.shape
    inner = w_q.shape[0]
    p = _capi.AttnParams(heads=heads, dim_head=inner // heads, query_dim=qd, w_q=_ptr(w_q), w_kv=_ptr(w_q), w_out=_ptr(w_out),
                         b_out=_ptr(b_out))
    need = lib.hn_attn_merge_workspace_bytes(C.byref(p), b, L)
    ws = WS.get(x.device, need)
    out = torch.empty_like(x)
    stats = torch.empty(b, heads, L, 2, dtype=torch.float32, device=x.device)
    _capi.check(lib.hn_attn_merge_fwd(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), o_parts.data_ptr(),
                                      stats_parts.data_ptr(), int(o_parts.shape[0]), b, L, stats.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _stream_ptr(x.device)), "hn_attn_merge_fwd")
    return out, stats


@torch.library.register_fake("healnet_hip::attention_merge")
def _(x, o_parts, stats_parts, w_q, w_out, b_out, heads, residual):
    return x.new_empty(x.shape, dtype=torch.float32), x.new_empty((x.shape[0], heads, x.shape[1], 2), dtype=torch.float32)


# ---- context split, TRAINING (include/healnet_hip.h "Context split: TRAINING, block level"; healnet_amd.dist drives it)
def cp_local_forward(x, ctx_slab, wts, heads):
    """The training forward of a cross block on THIS rank's slab -> (stats, saved, part, width): `part` = a view of the first
    b * L * heads * width floats of `saved` (the shard's normalised O, or P z on the shared-context binding), what the ranks exchange."""
    lib = _capi.lib()
    x, ctx = _f32c(x), _f32c(ctx_slab)
    p, (b, L, N, D, ld) = _attn_params(x, ctx, *wts, heads)
    width = lib.hn_attn_saved_part_width(C.byref(p), ld, b, L, N, D)
    if width <= 0:
        raise RuntimeError("healnet_amd: this block's context cannot be split (one token per rank)")
    _, stats, saved = _attention_fwd(x, ctx, None, *wts, heads, False, True)
    return stats, saved, saved[: b * L * heads * width], width


def cp_merge(parts, stats_parts, saved, stats, b, heads, L, width):
    """Fold the shards' (O or P z, statistics) pairs -- (G, b * L * heads * width), (G, b, heads, L, 2), index order -- into the
    head of `saved` and into `stats` (both overwritten in place)."""
    parts, stats_parts = _f32c(parts), _f32c(stats_parts)
    g = int(parts.shape[0])
    _capi.check(_capi.lib().hn_attn_merge_parts(parts.data_ptr(), stats_parts.data_ptr(), g, parts[0].numel(), stats_parts[0].numel(), b,
                                                heads, L, width, saved.data_ptr(), stats.data_ptr(), _stream_ptr(saved.device)),
                "hn_attn_merge_parts")


def cp_finish(x, ctx_slab, wts, heads, saved, residual=True):
    """x_out = LeakyReLU(O W_out^T + b_out) [+ x] from the merged `saved` (bit-identical on every rank)."""
    lib = _capi.lib()
    x, ctx = _f32c(x), _f32c(ctx_slab)
    p, (b, L, N, D, ld) = _attn_params(x, ctx, *wts, heads)
    ws = WS.get(x.device, lib.hn_attn_workspace_bytes(C.byref(p), 1, ld, b, L, N, D))
    out = torch.empty_like(x)
    _capi.check(lib.hn_attn_finish_fwd(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), ld, b, L, N, D, saved.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _stream_ptr(x.device)), "hn_attn_finish_fwd")
    return out


def cp_backward(dy, x, out, ctx_slab, wts, heads, stats, saved, owner):
    """hn_attn_bwd_cp -> [dx through the queries (partial, no residual term), d norm_w, d norm_b, d ctx_gamma, d ctx_beta, d w_q,
    d w_kv, d w_out, d b_out]: this rank's terms of the sums over ranks (the replicated ones only on the owner, zeros elsewhere)."""
    lib = _capi.lib()
    x, dy, out, ctx = _f32c(x), _f32c(dy), _f32c(out), _f32c(ctx_slab)
    p, (b, L, N, D, ld) = _attn_params(x, ctx, *wts, heads)
    ws = WS.get(x.device, lib.hn_attn_bwd_workspace_bytes(C.byref(p), 1, ld, b, L, N, D, 0))
    dx = torch.empty_like(x)
    g = [torch.zeros_like(t) for t in wts]
    grads = _capi.AttnGrads(norm_w=g[0].data_ptr(), norm_b=g[1].data_ptr(), ctx_gamma=g[2].data_ptr(), ctx_beta=g[3].data_ptr(),
                            w_q=g[4].data_ptr(), w_kv=g[5].data_ptr(), w_out=g[6].data_ptr(), b_out=g[7].data_ptr())
    _capi.check(lib.hn_attn_bwd_cp(C.byref(p), x.data_ptr(), out.data_ptr(), ctx.data_ptr(), ld, b, L, N, D, stats.data_ptr(),
                                   saved.data_ptr(), dy.data_ptr(), dx.data_ptr(), C.byref(grads), int(bool(owner)), ws.data_ptr(),
                                   ws.numel(), _stream_ptr(x.device)), "hn_attn_bwd_cp")
    return [dx] + g


class ContextSplitAttentionFn(torch.autograd.Function):
    """PreNorm(Attention) + residual of a cross block whose context is split over ranks, with its backward.
    apply(gather, reduce, owner, heads, ctx_slab, x, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out):
      gather(part, stats) -> (parts (G, ...), stats_parts (G, ...)) in rank order; reduce(list of tensors): sum over ranks, in place."""

    @staticmethod
    def forward(ctx, gather, reduce, owner, heads, ctx_slab, x, *wts):
        x = _f32c(x.detach())
        wts = tuple(t.detach() for t in wts)
        stats, saved, part, width = cp_local_forward(x, ctx_slab, wts, heads)
        parts, stats_parts = gather(part, stats)
        b, L = x.shape[0], x.shape[1]
        cp_merge(parts.reshape(parts.shape[0], -1), stats_parts, saved, stats, b, heads, L, width)
        out = cp_finish(x, ctx_slab, wts, heads, saved, True)
        ctx.save_for_backward(x, out, ctx_slab, stats, saved, *wts)
        ctx.cp = (reduce, owner, heads)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, ctx_slab, stats, saved, *wts = ctx.saved_tensors
        reduce, owner, heads = ctx.cp
        dout = dout.contiguous()
        g = cp_backward(dout, x, out, ctx_slab, wts, heads, stats, saved, owner)
        reduce(g)
        g[0] = g[0] + dout          # the residual path is replicated
        return (None, None, None, None, None, *g)


def _opt_out(t: Optional[torch.Tensor], like: Optional[torch.Tensor]):
    """Gradient buffer for an optional parameter: zeros like it, or an empty placeholder when the parameter is absent."""
    return torch.zeros_like(like) if like is not None else t


def _attention_bwd(dy, x, out, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads, residual,
                   stats, saved, dropout=0.0, rng=()):
    lib = _capi.lib()
    x, dy, out = _f32c(x), _f32c(dy), _f32c(out)
    ctx = None if context is None else _f32c(context)
    p, (b, L, N, D, ld) = _attn_params(x, ctx, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads)
    _set_drop(p, dropout, rng)
    m = _mask_u8(mask, b)
    has_ctx = int(ctx is not None)
    need = lib.hn_attn_bwd_workspace_bytes(C.byref(p), has_ctx, ld, b, L, N, D, int(m is not None))
    if need == 0:
        _capi.check(-1, "hn_attn_bwd_workspace_bytes")
    ws = WS.get(x.device, need)
    empty = torch.empty(0, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    g = [_opt_out(empty, t) for t in (norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out)]
    gp = lambda t: t.data_ptr() if t.numel() else None      # noqa: E731
    grads = _capi.AttnGrads(norm_w=gp(g[0]), norm_b=gp(g[1]), ctx_gamma=gp(g[2]), ctx_beta=gp(g[3]), w_q=gp(g[4]), w_kv=gp(g[5]),
                            w_out=gp(g[6]), b_out=gp(g[7]))
    _capi.check(lib.hn_attn_bwd(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), _ptr(ctx), ld, b, L, N, D, _ptr(m),
                                stats.data_ptr(), saved.data_ptr(), dy.data_ptr(), dx.data_ptr(), C.byref(grads), ws.data_ptr(),
                                ws.numel(), _stream_ptr(x.device)), "hn_attn_bwd")
    return [dx] + g


@torch.library.register_fake("healnet_hip::attention_bwd")
def _(dy, x, out, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads, residual, stats, saved, dropout=0.0, rng=()):
    empty = x.new_empty((0,), dtype=torch.float32)
    return [torch.empty_like(x)] + [empty if t is None else torch.empty_like(t)
                                    for t in (norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out)]


def _attn_setup(ctx, inputs, output):
    (x, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads, residual, train, dropout, rng) = inputs
    out, stats, saved = output
    ctx.dropout, ctx.rng = float(dropout), [int(v) for v in rng]
    if not train:
        raise RuntimeError("healnet_hip::attention_fwd was called with train=False on inputs that require grad")
    if context is not None and context.requires_grad:
        raise RuntimeError("healnet_hip::attention: no gradient flows to the context (include/healnet_hip.h hn_attn_bwd); "
                           "detach it -- HEALNet's contexts are encoded modality inputs, not activations")
    ctx.heads, ctx.residual = heads, residual
    ctx.opt = [t is not None for t in (context, mask, norm_w, norm_b, ctx_gamma, ctx_beta)]
    ctx.save_for_backward(x, out, stats, saved, w_q, w_kv, w_out, b_out,
                          *[t for t in (context, mask, norm_w, norm_b, ctx_gamma, ctx_beta) if t is not None])


def _attn_backward(ctx, dout, dstats, dsaved):
    x, out, stats, saved, w_q, w_kv, w_out, b_out, *rest = ctx.saved_tensors
    it = iter(rest)
    context, mask, norm_w, norm_b, ctx_gamma, ctx_beta = [next(it) if have else None for have in ctx.opt]
    g = torch.ops.healnet_hip.attention_bwd(dout.contiguous(), x, out, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv,
                                            w_out, b_out, ctx.heads, ctx.residual, stats, saved, ctx.dropout, ctx.rng)
    opt = lambda t, have: t if have else None      # noqa: E731
    return (g[0], None, None, opt(g[1], ctx.opt[2]), opt(g[2], ctx.opt[3]), opt(g[3], ctx.opt[4]), opt(g[4], ctx.opt[5]),
            g[5], g[6], g[7], g[8], None, None, None, None, None)


torch.library.register_autograd("healnet_hip::attention_fwd", _attn_backward, setup_context=_attn_setup)


def _attention(x, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads, residual):
    """CompositeImplicitAutograd: attention_fwd with the tape only when something requires grad."""
    train = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                            for t in (x, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out))
    return torch.ops.healnet_hip.attention_fwd(x, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads,
                                               residual, train)[0]


# ------------------------------------------------------------------------------------------------
# feed-forward block, head
# ------------------------------------------------------------------------------------------------
def _ff_params(x, norm_w, norm_b, w1, b1, w2, b2, gelu):
    dim = x.shape[-1]
    return _capi.FFParams(dim=dim, gate=1 if gelu else 0, norm_w=_ptr(norm_w), norm_b=_ptr(norm_b), w1=_ptr(w1), b1=_ptr(b1),
                          w2=_ptr(w2), b2=_ptr(b2)), x.numel() // dim


def _feed_forward(x, norm_w, norm_b, w1, b1, w2, b2, gelu, residual, dropout=0.0, rng=()):
    lib = _capi.lib()
    x = _f32c(x)
    p, rows = _ff_params(x, norm_w, norm_b, w1, b1, w2, b2, gelu)
    _set_drop(p, dropout, rng)
    ws = WS.get(x.device, lib.hn_ff_workspace_bytes(C.byref(p), rows))
    out = torch.empty_like(x)
    _capi.check(lib.hn_ff_fwd(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), rows, ws.data_ptr(), ws.numel(),
                              _stream_ptr(x.device)), "hn_ff_fwd")
    return out


@torch.library.register_fake("healnet_hip::feed_forward")
def _(x, norm_w, norm_b, w1, b1, w2, b2, gelu, residual, dropout=0.0, rng=()):
    return x.new_empty(x.shape, dtype=torch.float32)


def _feed_forward_bwd(dy, x, norm_w, norm_b, w1, b1, w2, b2, gelu, residual, dropout=0.0, rng=()):
    lib = _capi.lib()
    x, dy = _f32c(x), _f32c(dy)
    p, rows = _ff_params(x, norm_w, norm_b, w1, b1, w2, b2, gelu)
    _set_drop(p, dropout, rng)
    ws = WS.get(x.device, lib.hn_ff_bwd_workspace_bytes(C.byref(p), rows))
    empty = torch.empty(0, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    g = [_opt_out(empty, t) for t in (norm_w, norm_b, w1, b1, w2, b2)]
    gp = lambda t: t.data_ptr() if t.numel() else None      # noqa: E731
    grads = _capi.FFGrads(norm_w=gp(g[0]), norm_b=gp(g[1]), w1=gp(g[2]), b1=gp(g[3]), w2=gp(g[4]), b2=gp(g[5]))
    _capi.check(lib.hn_ff_bwd(C.byref(p), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), int(residual), rows, C.byref(grads),
                              ws.data_ptr(), ws.numel(), _stream_ptr(x.device)), "hn_ff_bwd")
    return [dx] + g


@torch.library.register_fake("healnet_hip::feed_forward_bwd")
def _(dy, x, norm_w, norm_b, w1, b1, w2, b2, gelu, residual, dropout=0.0, rng=()):
    empty = x.new_empty((0,), dtype=torch.float32)
    return [torch.empty_like(x)] + [empty if t is None else torch.empty_like(t) for t in (norm_w, norm_b, w1, b1, w2, b2)]


def _ff_setup(ctx, inputs, output):
    x, norm_w, norm_b, w1, b1, w2, b2, gelu, residual, dropout, rng = inputs
    ctx.gelu, ctx.residual, ctx.has_norm = gelu, residual, norm_w is not None
    ctx.dropout, ctx.rng = float(dropout), [int(v) for v in rng]
    ctx.save_for_backward(x, w1, b1, w2, b2, *([norm_w, norm_b] if norm_w is not None else []))


def _ff_backward(ctx, dout):
    x, w1, b1, w2, b2, *norm = ctx.saved_tensors
    norm_w, norm_b = norm if ctx.has_norm else (None, None)
    g = torch.ops.healnet_hip.feed_forward_bwd(dout.contiguous(), x, norm_w, norm_b, w1, b1, w2, b2, ctx.gelu, ctx.residual, ctx.dropout, ctx.rng)
    return (g[0], g[1] if ctx.has_norm else None, g[2] if ctx.has_norm else None, g[3], g[4], g[5], g[6], None, None, None, None)


torch.library.register_autograd("healnet_hip::feed_forward", _ff_backward, setup_context=_ff_setup)


# ------------------------------------------------------------------------------------------------
# latent block (self-attention + feed-forward)
# ------------------------------------------------------------------------------------------------
def _lb_params(x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2, gelu, fake=False):
    b, L, d = x.shape
    pf = (lambda t: None if t is None else _FAKE_PTR) if fake else _ptr
    ap = _capi.AttnParams(heads=heads, dim_head=w_q.shape[0] // heads, query_dim=d, norm_w=pf(a_norm_w), norm_b=pf(a_norm_b),
                          w_q=pf(w_q), w_kv=pf(w_kv), w_out=pf(w_out), b_out=pf(b_out))
    fp = _capi.FFParams(dim=d, gate=1 if gelu else 0, norm_w=pf(f_norm_w), norm_b=pf(f_norm_b), w1=pf(w1), b1=pf(b1), w2=pf(w2),
                        b2=pf(b2))
    return ap, fp, int(b), int(L), int(d)


def _latent_block_fwd(x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2, gelu, train, attn_dropout=0.0,
                      ff_dropout=0.0, rng=()):
    lib = _capi.lib()
    x = _f32c(x)
    ap, fp, b, L, d = _lb_params(x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2, gelu)
    if train:
        _set_drop(ap, attn_dropout, rng)
        _set_drop(fp, ff_dropout, rng, 1)
    need = lib.hn_latent_block_workspace_bytes(C.byref(ap), C.byref(fp), b, L)
    if need == 0:
        _capi.check(-1, "hn_latent_block_workspace_bytes")
    ws = WS.get(x.device, need)
    out = torch.empty_like(x)
    stats = torch.empty(b, heads, L, 2, dtype=torch.float32, device=x.device)
    if train:
        x_mid = torch.empty_like(x)
        saved = torch.empty(lib.hn_attn_saved_floats(C.byref(ap), 0, 0, b, L, L, d, 0), dtype=torch.float32, device=x.device)
    else:
        x_mid = saved = torch.empty(0, dtype=torch.float32, device=x.device)
    _capi.check(lib.hn_latent_block_fwd(C.byref(ap), C.byref(fp), x.data_ptr(), out.data_ptr(), b, L,
                                        x_mid.data_ptr() if train else None, stats.data_ptr(), saved.data_ptr() if train else None,
                                        ws.data_ptr(), ws.numel(), _stream_ptr(x.device)), "hn_latent_block_fwd")
    return out, x_mid, stats, saved


@torch.library.register_fake("healnet_hip::latent_block_fwd")
def _(x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2, gelu, train, attn_dropout=0.0, ff_dropout=0.0, rng=()):
    ap, fp, b, L, d = _lb_params(x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2, gelu, fake=True)
    n_saved = _capi.lib().hn_attn_saved_floats(C.byref(ap), 0, 0, b, L, L, d, 0) if train else 0
    e = x.new_empty
    return (e(x.shape, dtype=torch.float32), e(x.shape if train else (0,), dtype=torch.float32), e((b, heads, L, 2), dtype=torch.float32),
            e((n_saved,), dtype=torch.float32))


def _latent_block_bwd(dy, x_mid, stats, saved, x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2,
                      gelu, attn_dropout=0.0, ff_dropout=0.0, rng=()):
    lib = _capi.lib()
    x, dy = _f32c(x), _f32c(dy)
    ap, fp, b, L, d = _lb_params(x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2, gelu)
    _set_drop(ap, attn_dropout, rng)
    _set_drop(fp, ff_dropout, rng, 1)
    need = lib.hn_latent_block_bwd_workspace_bytes(C.byref(ap), C.byref(fp), b, L)
    if need == 0:
        _capi.check(-1, "hn_latent_block_bwd_workspace_bytes")
    ws = WS.get(x.device, need)
    empty = torch.empty(0, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    ga = [_opt_out(empty, t) for t in (a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out)]
    gf = [_opt_out(empty, t) for t in (f_norm_w, f_norm_b, w1, b1, w2, b2)]
    gp = lambda t: t.data_ptr() if t.numel() else None      # noqa: E731
    agr = _capi.AttnGrads(norm_w=gp(ga[0]), norm_b=gp(ga[1]), w_q=gp(ga[2]), w_kv=gp(ga[3]), w_out=gp(ga[4]), b_out=gp(ga[5]))
    fgr = _capi.FFGrads(norm_w=gp(gf[0]), norm_b=gp(gf[1]), w1=gp(gf[2]), b1=gp(gf[3]), w2=gp(gf[4]), b2=gp(gf[5]))
    _capi.check(lib.hn_latent_block_bwd(C.byref(ap), C.byref(fp), x.data_ptr(), x_mid.data_ptr(), b, L, stats.data_ptr(),
                                        saved.data_ptr(), dy.data_ptr(), dx.data_ptr(), C.byref(agr), C.byref(fgr), ws.data_ptr(),
                                        ws.numel(), _stream_ptr(x.device)), "hn_latent_block_bwd")
    return [dx] + ga + gf


@torch.library.register_fake("healnet_hip::latent_block_bwd")
def _(dy, x_mid, stats, saved, x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2, gelu, attn_dropout=0.0,
      ff_dropout=0.0, rng=()):
    empty = x.new_empty((0,), dtype=torch.float32)
    return [torch.empty_like(x)] + [empty if t is None else torch.empty_like(t)
                                    for t in (a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, f_norm_w, f_norm_b, w1, b1, w2, b2)]


def _lb_setup(ctx, inputs, output):
    (x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2, gelu, train, attn_dropout, ff_dropout, rng) = inputs
    out, x_mid, stats, saved = output
    ctx.drop = (float(attn_dropout), float(ff_dropout), [int(v) for v in rng])
    if not train:
        raise RuntimeError("healnet_hip::latent_block_fwd was called with train=False on inputs that require grad")
    ctx.heads, ctx.gelu = heads, gelu
    ctx.opt = [t is not None for t in (a_norm_w, a_norm_b, f_norm_w, f_norm_b)]
    ctx.save_for_backward(x, x_mid, stats, saved, w_q, w_kv, w_out, b_out, w1, b1, w2, b2,
                          *[t for t in (a_norm_w, a_norm_b, f_norm_w, f_norm_b) if t is not None])


def _lb_backward(ctx, dout, dmid, dstats, dsaved):
    x, x_mid, stats, saved, w_q, w_kv, w_out, b_out, w1, b1, w2, b2, *rest = ctx.saved_tensors
    it = iter(rest)
    a_nw, a_nb, f_nw, f_nb = [next(it) if have else None for have in ctx.opt]
    g = torch.ops.healnet_hip.latent_block_bwd(dout.contiguous(), x_mid, stats, saved, x, a_nw, a_nb, w_q, w_kv, w_out, b_out, ctx.heads,
                                               f_nw, f_nb, w1, b1, w2, b2, ctx.gelu, *ctx.drop)
    o = ctx.opt
    return (g[0], g[1] if o[0] else None, g[2] if o[1] else None, g[3], g[4], g[5], g[6], None,
            g[7] if o[2] else None, g[8] if o[3] else None, g[9], g[10], g[11], g[12], None, None, None, None, None)


torch.library.register_autograd("healnet_hip::latent_block_fwd", _lb_backward, setup_context=_lb_setup)


def _latent_block(x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2, gelu):
    ts = (x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, f_norm_w, f_norm_b, w1, b1, w2, b2)
    train = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)
    return torch.ops.healnet_hip.latent_block_fwd(x, a_norm_w, a_norm_b, w_q, w_kv, w_out, b_out, heads, f_norm_w, f_norm_b, w1, b1, w2, b2,
                                                  gelu, train)[0]


def _head(x, norm_w, norm_b, w, bias):
    x = _f32c(x)
    b, L, d = x.shape
    out = torch.empty(b, w.shape[0], dtype=torch.float32, device=x.device)
    _capi.check(_capi.lib().hn_head_fwd(x.data_ptr(), b, L, d, _ptr(norm_w), _ptr(norm_b), _ptr(w), _ptr(bias), w.shape[0],
                                        out.data_ptr(), _stream_ptr(x.device)), "hn_head_fwd")
    return out


@torch.library.register_fake("healnet_hip::head")
def _(x, norm_w, norm_b, w, bias):
    return x.new_empty((x.shape[0], w.shape[0]), dtype=torch.float32)


def _head_bwd(dlogits, x, norm_w, norm_b, w):
    lib = _capi.lib()
    x, dl = _f32c(x), _f32c(dlogits)
    b, L, d = x.shape
    out_dims = w.shape[0]
    dx = torch.empty_like(x)
    g = [torch.zeros_like(norm_w), torch.zeros_like(norm_b), torch.zeros_like(w), torch.zeros(out_dims, dtype=torch.float32, device=x.device)]
    ws = WS.get(x.device, lib.hn_head_bwd_workspace_bytes(b, d, out_dims))
    _capi.check(lib.hn_head_bwd(x.data_ptr(), b, L, d, _ptr(norm_w), _ptr(norm_b), _ptr(w), out_dims, dl.data_ptr(), dx.data_ptr(),
                                g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), ws.data_ptr(), ws.numel(),
                                _stream_ptr(x.device)), "hn_head_bwd")
    return [dx] + g


@torch.library.register_fake("healnet_hip::head_bwd")
def _(dlogits, x, norm_w, norm_b, w):
    return [torch.empty_like(x), torch.empty_like(norm_w), torch.empty_like(norm_b), torch.empty_like(w),
            x.new_empty((w.shape[0],), dtype=torch.float32)]


def _head_setup(ctx, inputs, output):
    x, norm_w, norm_b, w, bias = inputs
    ctx.save_for_backward(x, norm_w, norm_b, w)


def _head_backward(ctx, dout):
    g = torch.ops.healnet_hip.head_bwd(dout.contiguous(), *ctx.saved_tensors)
    return tuple(g)


torch.library.register_autograd("healnet_hip::head", _head_backward, setup_context=_head_setup)


# ------------------------------------------------------------------------------------------------
# whole fusion stack
# ------------------------------------------------------------------------------------------------
def _guarded(call, what: str, device: torch.device, rerun: bool) -> None:
    """Run a fused entry point; on HN_E_CORESIDENCY (a cluster-mode chain launched EARLIER lost an exchange: include/healnet_hip.h)
    drain the device, consume the report, warn once -- and either run the call again (``rerun``: its own inputs are clean, and
    cluster mode is off from here on) or pass the error on (a backward whose tape may already hold the NaN rows: the step has to be
    repeated from its forward, healnet_amd.train.retry_step does that)."""
    try:
        _capi.check(call(), what)
    except _capi.CoresidencyLost as err:
        if torch.cuda.is_current_stream_capturing():
            raise
        _capi.note_coresidency(err, device.index if device.index is not None else torch.cuda.current_device(),
                               lambda: torch.cuda.synchronize(device))
        if not rerun:
            raise
        _capi.check(call(), what)


def _out_shape(spec: Spec, b: int, embeddings: bool):
    return (b, spec.l_c, spec.l_d) if (embeddings or not spec.head) else (b, spec.out_dims)


def _fusion_forward(tensors, mask, params, spec, skip_self, embeddings, keep_stats):
    lib = _capi.lib()
    sp = spec_of(spec)
    device = params[0].device
    with torch.cuda.device(device):        # kernels are launched on the CURRENT device of the calling thread
        model, keep = sp.model_cached(params)
        inp, held, b = sp.inputs(tensors)
        shape_key = (b, tuple(None if t is None else (tuple(t.shape), t.dtype) for t in held))
        need = sp.sizes_cached("fwd", shape_key, lambda: lib.hn_fusion_workspace_bytes(C.byref(model), inp, b))
        if need == 0:
            _capi.check(-1, "hn_fusion_workspace_bytes")
        ws = WS.get(device, need)
        out = torch.empty(_out_shape(sp, b, embeddings), dtype=torch.float32, device=device)
        stats_ptrs = x_ptrs = None
        if keep_stats:
            # every attention block's softmax statistics and input, one row per slot (layer-major: cross_0..cross_{M-1}, self);
            # the blocks write them in place (hn_fusion_forward chains the latent array through the trace slots: no copies)
            stats = torch.empty(sp.n_slots, b * sp.max_heads * sp.l_c * 2, dtype=torch.float32, device=device)
            trace = torch.empty(sp.n_slots, b, sp.l_c, sp.l_d, dtype=torch.float32, device=device)
            stats_ptrs = (C.c_void_p * sp.n_slots)()
            x_ptrs = (C.c_void_p * sp.n_slots)()
            for slot in range(sp.n_slots):
                j = slot % (sp.M + 1)
                if (j < sp.M and held[j] is not None) or (j == sp.M and sp.spca > 0):
                    stats_ptrs[slot] = stats[slot].data_ptr()
                    x_ptrs[slot] = trace[slot].data_ptr()
        else:
            stats = torch.empty(0, dtype=torch.float32, device=device)
            trace = torch.empty(0, dtype=torch.float32, device=device)
        profile = getattr(_tls, "profile", None)
        _tls.profile = None
        _guarded(lambda: lib.hn_fusion_forward(C.byref(model), inp, b, _ptr(mask), int(skip_self), int(embeddings), out.data_ptr(),
                                               stats_ptrs, x_ptrs, ws.data_ptr(), ws.numel(), _stream_ptr(device), profile),
                 "hn_fusion_forward", device, rerun=True)
    return out, stats, trace


def _batch_of(tensors):
    for t in tensors:
        if t is not None:
            return int(t.shape[0])
    raise ValueError("at least one modality must be present")


@torch.library.register_fake("healnet_hip::fusion_forward")
def _(tensors, mask, params, spec, skip_self, embeddings, keep_stats):
    sp = spec_of(spec)
    b = _batch_of(tensors)
    like = params[0]
    out = like.new_empty(_out_shape(sp, b, embeddings), dtype=torch.float32)
    if keep_stats:
        return (out, like.new_empty((sp.n_slots, b * sp.max_heads * sp.l_c * 2), dtype=torch.float32),
                like.new_empty((sp.n_slots, b, sp.l_c, sp.l_d), dtype=torch.float32))
    return out, like.new_empty((0,), dtype=torch.float32), like.new_empty((0,), dtype=torch.float32)


def _fusion_forward_train(tensors, mask, params, spec, skip_self, embeddings, rng, grad_buffer, grad_offsets):
    lib = _capi.lib()
    sp = spec_of(spec)
    device = params[0].device
    with torch.cuda.device(device):
        model, keep = sp.model_train_cached(params, rng)
        inp, held, b = sp.inputs(tensors)
        masked = int(mask is not None)
        shape_key = (b, masked, int(skip_self), rng is not None, tuple(None if t is None else (tuple(t.shape), t.dtype) for t in held))
        tape_bytes, need = sp.sizes_cached("train", shape_key, lambda: (
            lib.hn_fusion_tape_bytes(C.byref(model), inp, b, masked, int(skip_self)), lib.hn_fusion_workspace_bytes(C.byref(model), inp, b)))
        if tape_bytes == 0 or need == 0:
            _capi.check(-1, "hn_fusion_tape_bytes")
        tape = torch.empty(tape_bytes, dtype=torch.uint8, device=device)
        ws = WS.get(device, need)
        out = torch.empty(_out_shape(sp, b, embeddings), dtype=torch.float32, device=device)
        _guarded(lambda: lib.hn_fusion_forward_train(C.byref(model), inp, b, _ptr(mask), int(skip_self), int(embeddings), out.data_ptr(),
                                                     None, None, tape.data_ptr(), tape.numel(), ws.data_ptr(), ws.numel(),
                                                     _stream_ptr(device)), "hn_fusion_forward_train", device, rerun=True)
        # where the tape keeps every attention block's softmax statistics / input (float offsets; -1 = block not executed):
        # the host views them in place for Attention.attn_weights -- laid out with THIS call's descriptor (dropout included)
        def _layout():
            so, xo = (C.c_size_t * sp.n_slots)(), (C.c_size_t * sp.n_slots)()
            _capi.check(lib.hn_fusion_tape_layout(C.byref(model), inp, b, masked, int(skip_self), so, xo), "hn_fusion_tape_layout")
            none = C.c_size_t(-1).value
            return torch.tensor([-1 if v == none else int(v) for v in list(so) + list(xo)], dtype=torch.int64)
        layout = sp.sizes_cached("layout", shape_key, _layout).clone()      # (a fresh tensor per call: autograd owns the outputs)
    return out, tape, layout


@torch.library.register_fake("healnet_hip::fusion_forward_train")
def _(tensors, mask, params, spec, skip_self, embeddings, rng, grad_buffer, grad_offsets):
    sp = spec_of(spec)
    b = _batch_of(tensors)
    rng_fake = None if rng is None else torch.zeros(2, dtype=torch.int64)   # sizes depend on WHETHER blocks drop, not on the seed
    model, keep = sp.model(None, rng_fake)
    inp, _, _ = sp.inputs(tensors, fake=True)
    tape_bytes = _capi.lib().hn_fusion_tape_bytes(C.byref(model), inp, b, int(mask is not None), int(skip_self))
    like = params[0]
    return (like.new_empty(_out_shape(sp, b, embeddings), dtype=torch.float32), like.new_empty((tape_bytes,), dtype=torch.uint8),
            torch.empty((2 * sp.n_slots,), dtype=torch.int64, device="cpu"))


def _run_fusion_backward(dout, tape, tensors, mask, params, spec, skip_self, embeddings, rng, gptr, hook):
    lib = _capi.lib()
    sp = spec_of(spec)
    device = params[0].device
    model, keep = sp.model_train_cached(params, rng)
    inp, held, b = sp.inputs(tensors)
    grads, keep_g = sp.grads_cached(gptr)
    masked = int(mask is not None)
    shape_key = (b, masked, rng is not None, tuple(None if t is None else (tuple(t.shape), t.dtype) for t in held))
    need = sp.sizes_cached("bwd", shape_key, lambda: lib.hn_fusion_backward_workspace_bytes(C.byref(model), inp, b, masked))
    if need == 0:
        _capi.check(-1, "hn_fusion_backward_workspace_bytes")
    ws = WS.get(device, need)
    dout = dout.contiguous().float()
    ready = None
    if hook is not None:
        hook.begin(_stream_ptr(device))
        ready = C.byref(hook.ready)
    _guarded(lambda: lib.hn_fusion_backward(C.byref(model), inp, b, _ptr(mask), int(skip_self), int(embeddings), dout.data_ptr(),
                                            tape.data_ptr(), C.byref(grads), ws.data_ptr(), ws.numel(), _stream_ptr(device), ready),
             "hn_fusion_backward", device, rerun=False)
    if hook is not None:
        hook.end(_stream_ptr(device))


def _fusion_backward(dout, tape, tensors, mask, params, spec, skip_self, embeddings, rng, needs_grad):
    """Gradients as fresh tensors (zero-initialised, the kernels accumulate), an empty tensor for parameters without one."""
    device = params[0].device
    with torch.cuda.device(device):
        empty = torch.empty(0, dtype=torch.float32, device=device)
        out = [torch.zeros_like(p, dtype=torch.float32) if need else empty for p, need in zip(params, needs_grad)]
        gptr = [g.data_ptr() if need else None for g, need in zip(out, needs_grad)]
        _run_fusion_backward(dout, tape, tensors, mask, params, spec, skip_self, embeddings, rng, gptr, None)
    return out


@torch.library.register_fake("healnet_hip::fusion_backward")
def _(dout, tape, tensors, mask, params, spec, skip_self, embeddings, rng, needs_grad):
    return [torch.empty_like(p) if need else p.new_empty((0,)) for p, need in zip(params, needs_grad)]


def _fusion_backward_into(dout, tape, tensors, mask, params, spec, skip_self, embeddings, rng, grad_buffer, grad_offsets):
    """The kernels accumulate straight into ``grad_buffer[grad_offsets[i] : ...]`` (float offsets, -1 = no gradient for that
    parameter): no per-parameter zero tensors, no AccumulateGrad pass.  A gradient-readiness hook registered for the buffer
    (healnet_amd.dist.GradReadyAllReduce) is driven from inside hn_fusion_backward."""
    device = params[0].device
    with torch.cuda.device(device):
        base = grad_buffer.data_ptr()
        gptr = [base + 4 * off if off >= 0 else None for off in grad_offsets]
        _run_fusion_backward(dout, tape, tensors, mask, params, spec, skip_self, embeddings, rng, gptr, _BACKWARD_HOOKS.get(base))


@torch.library.register_fake("healnet_hip::fusion_backward_into")
def _(dout, tape, tensors, mask, params, spec, skip_self, embeddings, rng, grad_buffer, grad_offsets):
    return None


def _fusion_setup(ctx, inputs, output):
    tensors, mask, params, spec, skip_self, embeddings, rng, grad_buffer, grad_offsets = inputs
    out, tape, layout = output
    ctx.spec, ctx.skip_self, ctx.embeddings, ctx.grad_offsets = spec, skip_self, embeddings, list(grad_offsets)
    # no zero "gradients" for the outputs nobody differentiates: left at its default autograd materialised zeros_like(tape) in every
    # backward -- a 450 MB byte fill per cfg4 step (58 us on the GPU, plus the allocation), found in the step's kernel list
    ctx.set_materialize_grads(False)
    ctx.present = [t is not None for t in tensors]
    ctx.n_params = len(params)
    ctx.flags = (mask is not None, rng is not None)
    ctx.needs = [bool(p.requires_grad) for p in params]
    # the flat gradient buffer is WRITTEN by every backward (and by the side-stream all-reduce on its views): kept as a plain
    # attribute, not through save_for_backward -- with two forwards in front of one backward (two model calls summed into one
    # loss) the first backward bumps its version counter and the second would fail autograd's saved-tensor check
    ctx.grad_buffer = grad_buffer
    extra = [t for t in (mask, rng) if t is not None]
    ctx.save_for_backward(tape, *params, *[t for t in tensors if t is not None], *extra)


def _fusion_backward_formula(ctx, dout, dtape, dlayout):
    if dout is None:                                         # the output itself took no gradient (only reachable through tape / layout)
        d_tensors = [None] * len(ctx.present) if all(ctx.present) else None
        return d_tensors, None, [None] * ctx.n_params, None, None, None, None, None, ([] if not ctx.grad_offsets else None)
    saved = list(ctx.saved_tensors)
    tape, params = saved[0], saved[1:1 + ctx.n_params]
    it = iter(saved[1 + ctx.n_params:])
    tensors = [next(it) if have else None for have in ctx.present]
    mask, rng = [next(it) if have else None for have in ctx.flags]
    grad_buffer = ctx.grad_buffer
    if grad_buffer is not None:
        torch.ops.healnet_hip.fusion_backward_into(dout.contiguous(), tape, tensors, mask, params, ctx.spec, ctx.skip_self,
                                                   ctx.embeddings, rng, grad_buffer, ctx.grad_offsets)
        grads = [None] * len(params)
    else:
        g = torch.ops.healnet_hip.fusion_backward(dout.contiguous(), tape, tensors, mask, params, ctx.spec, ctx.skip_self,
                                                  ctx.embeddings, rng, ctx.needs)
        grads = [gi if need else None for gi, need in zip(g, ctx.needs)]
    # torch.library wants the structure of the inputs back: a list that held only tensors is a list of (optional) gradients,
    # a list with a None entry / of ints is a single leaf
    d_tensors = [None] * len(ctx.present) if all(ctx.present) else None
    return d_tensors, None, grads, None, None, None, None, None, ([] if not ctx.grad_offsets else None)


torch.library.register_autograd("healnet_hip::fusion_forward_train", _fusion_backward_formula, setup_context=_fusion_setup)


class FusionTrainFn(torch.autograd.Function):
    """The eager route of ``HealNet.forward`` in training: the SAME two implementations as ``torch.ops.healnet_hip.fusion_forward_train``
    and its registered backward (``_fusion_forward_train`` / ``_run_fusion_backward``: one C-ABI call each), behind a plain
    ``autograd.Function`` instead of the dispatcher.  Why: at the reference's tuned TCGA shapes a step is ~90 launches that the C side
    enqueues in ~0.2 ms, while the operator route added ~0.7 ms of host time around them (cProfile on the GPU box, round 5: pytree
    flatten / unflatten of the 40-tensor parameter list on every op call, ``torch.library``'s autograd wrapper, two redispatches) --
    the eager step was host-bound at 1.2 ms against 0.73 ms of GPU work.  The registered operator stays what ``torch.compile`` traces
    (``HealNet.forward`` takes it under ``torch.compiler.is_compiling()``; ``HN_FORCE_TORCH_OPS=1`` forces it everywhere).

    apply(meta, *flat): ``meta`` = (n_tensors, mask?, rng?, spec, skip_self, embeddings, grad_buffer, grad_offsets) and ``flat`` = the
    modality tensors (None for a missing one) followed by the parameters -- parameters must be positional for autograd to see them."""

    @staticmethod
    def forward(ctx, meta, *flat):
        n_t, mask, rng, spec, skip_self, embeddings, grad_buffer, grad_offsets = meta
        tensors, params = list(flat[:n_t]), list(flat[n_t:])
        out, tape, layout = _fusion_forward_train(tensors, mask, params, spec, skip_self, embeddings, rng, grad_buffer, grad_offsets)
        ctx.meta = meta
        ctx.tensors, ctx.params = tensors, params           # (inputs and parameters are alive anyway; the tape is what this step owns)
        ctx.tape = tape
        # what save_for_backward would have checked (not used: it costs the host time this route exists to avoid): an
        # in-place update of a parameter between this forward and its backward -- opt.step() before a delayed backward -- must raise,
        # not differentiate against the new weights
        ctx.versions = [p._version for p in params]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(tape, layout)
        return out, tape, layout

    @staticmethod
    def backward(ctx, dout, _dtape, _dlayout):
        n_t, mask, rng, spec, skip_self, embeddings, grad_buffer, grad_offsets = ctx.meta
        n = 1 + n_t + len(ctx.params)
        if dout is None:
            return (None,) * n
        params, tensors, tape = ctx.params, ctx.tensors, ctx.tape
        if tape is None:
            raise RuntimeError("healnet_amd: backward through the fused forward a second time (its tape was released after the first "
                               "backward; run the forward again)")
        for p, v in zip(params, ctx.versions):
            if p._version != v:
                raise RuntimeError("healnet_amd: a parameter of the model was modified in place between the forward and its backward "
                                   f"(version {p._version}, expected {v}): the tape was recorded against the old weights")
        ctx.tape = None                                     # the step's tape (up to ~450 MB) is not kept alive by a lingering grad_fn
        device = params[0].device
        with torch.cuda.device(device):
            if grad_buffer is not None:
                base = grad_buffer.data_ptr()
                gptr = [base + 4 * off if off >= 0 else None for off in grad_offsets]
                _run_fusion_backward(dout.contiguous(), tape, tensors, mask, params, spec, skip_self, embeddings, rng, gptr,
                                     _BACKWARD_HOOKS.get(base))
                grads = [None] * len(params)
            else:
                needs = ctx.needs_input_grad[1 + n_t:]
                empty = None
                outg = [torch.zeros_like(p, dtype=torch.float32) if need else empty for p, need in zip(params, needs)]
                gptr = [g.data_ptr() if g is not None else None for g in outg]
                _run_fusion_backward(dout.contiguous(), tape, tensors, mask, params, spec, skip_self, embeddings, rng, gptr, None)
                grads = outg
        return (None,) + (None,) * n_t + tuple(grads)


FORCE_TORCH_OPS = os.environ.get("HN_FORCE_TORCH_OPS", "0") == "1"


for _name, _fn in (("fourier_encode_concat", _fourier_encode_concat), ("encode_norm", _encode_norm), ("attention_fwd", _attention_fwd),
                   ("attention_bwd", _attention_bwd), ("feed_forward", _feed_forward), ("feed_forward_bwd", _feed_forward_bwd),
                   ("head", _head), ("head_bwd", _head_bwd), ("temperature_softmax", _temperature_softmax),
                   ("latent_block_fwd", _latent_block_fwd), ("latent_block_bwd", _latent_block_bwd),
                   ("fusion_forward", _fusion_forward), ("fusion_forward_train", _fusion_forward_train),
                   ("fusion_backward", _fusion_backward), ("fusion_backward_into", _fusion_backward_into),
                   ("encode_norm_slab", _encode_norm_slab), ("attention_partial", _attention_partial),
                   ("attention_merge", _attention_merge)):
    _lib.impl(_name, _fn, "CUDA")
_lib.impl("attention", _attention, "CompositeImplicitAutograd")
_lib.impl("latent_block", _latent_block, "CompositeImplicitAutograd")
