"""GPU: backward kernels (through the C ABI) against autograd of the CPU oracle."""
import ctypes as C

import pytest
import torch

from conftest import assert_close
from oracle import healnet_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def capi():
    from healnet_amd import _capi
    return _capi


def _ws(nbytes):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=DEV)


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("dim,rows,snn,norm,residual", [(128, 512, True, True, True), (16, 37, False, True, True),
                                                        (119, 75, True, True, False), (32, 64, True, False, True)])
def test_ff_backward(capi, dim, rows, snn, norm, residual):
    gen = torch.Generator().manual_seed(dim + rows)
    p = {k: (torch.randn(*s, generator=gen) * sc).requires_grad_(True) for k, (s, sc) in dict(
        nw=((dim,), 0.3), nb=((dim,), 0.3), w1=((8 * dim, dim), dim ** -0.5), b1=((8 * dim,), 0.2),
        w2=((dim, 4 * dim), (4 * dim) ** -0.5), b2=((dim,), 0.2)).items()}
    with torch.no_grad():
        p["nw"].add_(1.0)
    x = (torch.randn(1, rows, dim, generator=gen) * 1.5).requires_grad_(True)
    dy = torch.randn(1, rows, dim, generator=gen)
    xin = O.layer_norm(x, p["nw"], p["nb"]) if norm else x
    y = O.feed_forward(xin, p["w1"], p["b1"], p["w2"], p["b2"], snn) + (x if residual else 0)
    y.backward(dy)
    d = {k: v.detach().to(DEV).contiguous() for k, v in p.items()}
    gr = {k: torch.zeros_like(v) for k, v in d.items()}
    params = capi.FFParams(dim=dim, gate=0 if snn else 1, norm_w=d["nw"].data_ptr() if norm else None,
                           norm_b=d["nb"].data_ptr() if norm else None, w1=d["w1"].data_ptr(), b1=d["b1"].data_ptr(),
                           w2=d["w2"].data_ptr(), b2=d["b2"].data_ptr())
    grads = capi.FFGrads(norm_w=gr["nw"].data_ptr() if norm else None, norm_b=gr["nb"].data_ptr() if norm else None,
                         w1=gr["w1"].data_ptr(), b1=gr["b1"].data_ptr(), w2=gr["w2"].data_ptr(), b2=gr["b2"].data_ptr())
    lib = capi.lib()
    ws = _ws(lib.hn_ff_bwd_workspace_bytes(C.byref(params), rows))
    xd, dyd = x.detach().to(DEV).contiguous(), dy.to(DEV).contiguous()
    dx = torch.empty_like(xd)
    capi.check(lib.hn_ff_bwd(C.byref(params), xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), int(residual), rows, C.byref(grads),
                             ws.data_ptr(), ws.numel(), _stream()), "hn_ff_bwd")
    assert_close(dx.cpu(), x.grad, rel=2e-4, what="ff.dx")
    for k in (["nw", "nb"] if norm else []) + ["w1", "b1", "w2", "b2"]:
        assert_close(gr[k].cpu(), p[k].grad, rel=2e-4, what="ff.d" + k)
    # accumulation semantics: a second call doubles the parameter gradients
    capi.check(lib.hn_ff_bwd(C.byref(params), xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), int(residual), rows, C.byref(grads),
                             ws.data_ptr(), ws.numel(), _stream()), "hn_ff_bwd")
    assert_close(gr["w2"].cpu(), 2 * p["w2"].grad, rel=2e-4, what="ff.dw2 accumulate")


@pytest.mark.parametrize("b,L,d,out", [(32, 128, 128, 4), (3, 25, 119, 5)])
def test_head_backward(capi, b, L, d, out):
    gen = torch.Generator().manual_seed(b + d)
    nw = (1 + 0.3 * torch.randn(d, generator=gen)).requires_grad_(True)
    nb = (0.3 * torch.randn(d, generator=gen)).requires_grad_(True)
    w = (torch.randn(out, d, generator=gen) * d ** -0.5).requires_grad_(True)
    bias = (0.2 * torch.randn(out, generator=gen)).requires_grad_(True)
    x = (torch.randn(b, L, d, generator=gen)).requires_grad_(True)
    dl = torch.randn(b, out, generator=gen)
    (O.layer_norm(x.mean(1), nw, nb) @ w.t() + bias).backward(dl)
    dev = [t.detach().to(DEV).contiguous() for t in (x, nw, nb, w, bias, dl)]
    dx = torch.empty_like(dev[0])
    g = [torch.zeros_like(t) for t in dev[1:5]]
    lib = capi.lib()
    ws = _ws(lib.hn_head_bwd_workspace_bytes(b, d, out))
    capi.check(lib.hn_head_bwd(dev[0].data_ptr(), b, L, d, dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), out,
                               dev[5].data_ptr(), dx.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                               g[3].data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "hn_head_bwd")
    assert_close(dx.cpu(), x.grad, rel=2e-4, what="head.dx")
    for got, want, nm in zip(g, (nw, nb, w, bias), ("dnw", "dnb", "dw", "dbias")):
        assert_close(got.cpu(), want.grad, rel=2e-4, what="head." + nm)
