"""``torch.ops.healnet_hip.*`` -- the C-ABI entry points registered as PyTorch operators (CUDA/HIP dispatch key
only: there is no CPU kernel, calling them with CPU tensors raises NotImplementedError from the dispatcher).

    torch.ops.healnet_hip.fourier_encode_concat(data, num_freq_bands, max_freq, fourier_encode_data) -> (b, N, D)
    torch.ops.healnet_hip.encode_norm(data, num_freq_bands, max_freq, fourier_encode_data, pitch)    -> (b, N, pitch)
    torch.ops.healnet_hip.attention(x, context?, mask?, norm_w?, norm_b?, ctx_gamma?, ctx_beta?, w_q, w_kv, w_out, b_out,
                                    heads, residual) -> (b, L, query_dim)
    torch.ops.healnet_hip.feed_forward(x, norm_w?, norm_b?, w1, b1, w2, b2, gelu, residual) -> like x
    torch.ops.healnet_hip.head(x, norm_w, norm_b, w, bias) -> (b, out_dims)
    torch.ops.healnet_hip.temperature_softmax(logits, temperature) -> like logits (softmax over the last dim)
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _capi
from .healnet import _WS, _f32c, _ptr, _stream_ptr, fourier_encode_concat as _encode, temperature_softmax as _tsoftmax

_lib = torch.library.Library("healnet_hip", "DEF")
_lib.define("fourier_encode_concat(Tensor data, int num_freq_bands, float max_freq, bool fourier_encode_data) -> Tensor")
_lib.define("encode_norm(Tensor data, int num_freq_bands, float max_freq, bool fourier_encode_data, int pitch) -> Tensor")
_lib.define("attention(Tensor x, Tensor? context, Tensor? mask, Tensor? norm_w, Tensor? norm_b, Tensor? ctx_gamma, "
            "Tensor? ctx_beta, Tensor w_q, Tensor w_kv, Tensor w_out, Tensor b_out, int heads, bool residual) -> Tensor")
_lib.define("feed_forward(Tensor x, Tensor? norm_w, Tensor? norm_b, Tensor w1, Tensor b1, Tensor w2, Tensor b2, bool gelu, "
            "bool residual) -> Tensor")
_lib.define("head(Tensor x, Tensor norm_w, Tensor norm_b, Tensor w, Tensor bias) -> Tensor")
_lib.define("temperature_softmax(Tensor logits, float temperature) -> Tensor")


def _fourier_encode_concat(data, num_freq_bands, max_freq, fourier_encode_data):
    return _encode(data, num_freq_bands, max_freq, fourier_encode_data)


def _encode_norm(data, num_freq_bands, max_freq, fourier_encode_data, pitch):
    x = _f32c(data)
    b, spatial, ch = x.shape[0], list(x.shape[1:-1]), x.shape[-1]
    n = 1
    for s in spatial:
        n *= s
    z = torch.empty(b, n, pitch, dtype=torch.float32, device=x.device)
    sp = (C.c_int * len(spatial))(*spatial)
    _capi.check(_capi.lib().hn_encode_norm(x.data_ptr(), b, len(spatial), sp, ch, num_freq_bands, float(max_freq),
                                           int(fourier_encode_data), 1e-5, z.data_ptr(), pitch, _stream_ptr(x.device)),
                "hn_encode_norm")
    return z


def _attention(x, context, mask, norm_w, norm_b, ctx_gamma, ctx_beta, w_q, w_kv, w_out, b_out, heads, residual):
    lib = _capi.lib()
    x = _f32c(x)
    b, L, qd = x.shape
    inner = w_q.shape[0]
    ctx = None if context is None else _f32c(context)
    N, D, ld = (L, qd, 0) if ctx is None else (ctx.shape[1], w_kv.shape[1], ctx.shape[2])
    p = _capi.AttnParams(heads=heads, dim_head=inner // heads, query_dim=qd, norm_w=_ptr(norm_w), norm_b=_ptr(norm_b),
                         ctx_gamma=_ptr(ctx_gamma), ctx_beta=_ptr(ctx_beta), w_q=_ptr(w_q), w_kv=_ptr(w_kv),
                         w_out=_ptr(w_out), b_out=_ptr(b_out))
    m = None if mask is None else mask.reshape(b, -1).to(torch.uint8).contiguous()
    need = lib.hn_attn_workspace_bytes(C.byref(p), int(ctx is not None), ld, b, L, N, D)
    if need == 0:
        _capi.check(-1, "hn_attn_workspace_bytes")
    ws = _WS.get(x.device, need)
    out = torch.empty_like(x)
    _capi.check(lib.hn_attn_fwd(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), _ptr(ctx), ld, b, L, N, D, _ptr(m),
                                None, ws.data_ptr(), ws.numel(), _stream_ptr(x.device)), "hn_attn_fwd")
    return out


def _feed_forward(x, norm_w, norm_b, w1, b1, w2, b2, gelu, residual):
    lib = _capi.lib()
    x = _f32c(x)
    dim = x.shape[-1]
    rows = x.numel() // dim
    p = _capi.FFParams(dim=dim, gate=1 if gelu else 0, norm_w=_ptr(norm_w), norm_b=_ptr(norm_b), w1=_ptr(w1), b1=_ptr(b1),
                       w2=_ptr(w2), b2=_ptr(b2))
    ws = _WS.get(x.device, lib.hn_ff_workspace_bytes(C.byref(p), rows))
    out = torch.empty_like(x)
    _capi.check(lib.hn_ff_fwd(C.byref(p), x.data_ptr(), out.data_ptr(), int(residual), rows, ws.data_ptr(), ws.numel(),
                              _stream_ptr(x.device)), "hn_ff_fwd")
    return out


def _head(x, norm_w, norm_b, w, bias):
    x = _f32c(x)
    b, L, d = x.shape
    out = torch.empty(b, w.shape[0], dtype=torch.float32, device=x.device)
    _capi.check(_capi.lib().hn_head_fwd(x.data_ptr(), b, L, d, _ptr(norm_w), _ptr(norm_b), _ptr(w), _ptr(bias), w.shape[0],
                                        out.data_ptr(), _stream_ptr(x.device)), "hn_head_fwd")
    return out


for _name, _fn in (("fourier_encode_concat", _fourier_encode_concat), ("encode_norm", _encode_norm), ("attention", _attention),
                   ("feed_forward", _feed_forward), ("head", _head),
                   ("temperature_softmax", lambda logits, temperature: _tsoftmax(logits, temperature, -1))):
    _lib.impl(_name, _fn, "CUDA")
