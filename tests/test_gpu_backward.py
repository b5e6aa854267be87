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
    # all-NaN scratch: a kernel that reads workspace it has not written fails deterministically instead of once in a while
    return torch.full((max(int(nbytes), 256),), 0xFF, dtype=torch.uint8, device=DEV)


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
    dx = torch.full_like(xd, float("nan"))
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
    dx = torch.full_like(dev[0], float("nan"))
    g = [torch.zeros_like(t) for t in dev[1:5]]
    lib = capi.lib()
    ws = _ws(lib.hn_head_bwd_workspace_bytes(b, d, out))
    capi.check(lib.hn_head_bwd(dev[0].data_ptr(), b, L, d, dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), out,
                               dev[5].data_ptr(), dx.data_ptr(), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                               g[3].data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "hn_head_bwd")
    assert_close(dx.cpu(), x.grad, rel=2e-4, what="head.dx")
    for got, want, nm in zip(g, (nw, nb, w, bias), ("dnw", "dnb", "dw", "dbias")):
        assert_close(got.cpu(), want.grad, rel=2e-4, what="head." + nm)


def _attn_case(capi, b, L, N, D, heads, dh, qd, self_attn=False, masked=False, norm=True, residual=True, seed=0, _depth=0, nan_pad=False):
    import healnet_amd.healnet as H
    gen = torch.Generator().manual_seed(seed)
    inner = heads * dh
    kdim = qd if self_attn else D
    P = {"nw": 1 + 0.3 * torch.randn(qd, generator=gen), "nb": 0.3 * torch.randn(qd, generator=gen),
         "cg": 1 + 0.3 * torch.randn(kdim, generator=gen), "cb": 0.3 * torch.randn(kdim, generator=gen),
         "wq": 2.0 * torch.randn(inner, qd, generator=gen) * qd ** -0.5, "wkv": 2.0 * torch.randn(2 * inner, kdim, generator=gen) * kdim ** -0.5,
         "wo": torch.randn(qd, inner, generator=gen) * inner ** -0.5, "bo": 0.2 * torch.randn(qd, generator=gen)}
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    x = (torch.randn(b, L, qd, generator=gen) * 1.3).requires_grad_(True)
    ctx = None if self_attn else torch.rand(b, N, D, generator=gen) * 2
    mask = None
    if masked:
        mask = torch.rand(b, N, generator=gen) > 0.3
        mask[:, 0] = True
    dy = torch.randn(b, L, qd, generator=gen)
    xn = O.layer_norm(x, P["nw"], P["nb"]) if norm else x
    cn = None if self_attn else (O.layer_norm(ctx, P["cg"], P["cb"]) if norm else ctx)
    act = O.attention(xn, cn, P["wq"], P["wkv"], P["wo"], P["bo"], heads, mask)
    # |pre| of the LeakyReLU from its output: act = pre (pre > 0) or 0.01 pre (pre < 0)
    margin = float(torch.where(act > 0, act, act * 100).abs().min())
    if margin < 1e-5 and _depth < 20:
        return _attn_case(capi, b, L, N, D, heads, dh, qd, self_attn, masked, norm, residual, seed + 1, _depth + 1, nan_pad)
    y = act + (x if residual else 0)
    y.backward(dy)

    d = {k: v.detach().to(DEV).contiguous() for k, v in P.items()}
    gr = {k: torch.zeros_like(v) for k, v in d.items()}
    lib = capi.lib()
    params = capi.AttnParams(heads=heads, dim_head=dh, query_dim=qd, norm_w=d["nw"].data_ptr() if norm else None,
                             norm_b=d["nb"].data_ptr() if norm else None,
                             ctx_gamma=d["cg"].data_ptr() if (norm and not self_attn) else None,
                             ctx_beta=d["cb"].data_ptr() if (norm and not self_attn) else None, w_q=d["wq"].data_ptr(),
                             w_kv=d["wkv"].data_ptr(), w_out=d["wo"].data_ptr(), b_out=d["bo"].data_ptr())
    grads = capi.AttnGrads(norm_w=gr["nw"].data_ptr() if norm else None, norm_b=gr["nb"].data_ptr() if norm else None,
                           ctx_gamma=gr["cg"].data_ptr() if (norm and not self_attn) else None,
                           ctx_beta=gr["cb"].data_ptr() if (norm and not self_attn) else None, w_q=gr["wq"].data_ptr(),
                           w_kv=gr["wkv"].data_ptr(), w_out=gr["wo"].data_ptr(), b_out=gr["bo"].data_ptr())
    xd, dyd = x.detach().to(DEV).contiguous(), dy.to(DEV).contiguous()
    z, ld, has_ctx = None, 0, 0
    if not self_attn:
        has_ctx = 1
        if norm:
            ld = lib.hn_context_pitch(D, dh)
            z = H._normalise_context(ctx.to(DEV).contiguous(), ld)
            if nan_pad:      # the pad columns D .. ld-1 of a context row belong to nobody: whatever they hold must not reach a result
                assert ld > D
                z.view(b, N, ld)[..., D:] = float("nan")
        else:
            z, ld = ctx.to(DEV).contiguous(), D
    m8 = None if mask is None else mask.to(DEV).to(torch.uint8).contiguous()
    Nn, Dd = (L, qd) if self_attn else (N, D)
    nsaved = lib.hn_attn_saved_floats(C.byref(params), has_ctx, ld, b, L, Nn, Dd, int(masked))
    saved = torch.full((nsaved,), float("nan"), dtype=torch.float32, device=DEV)
    stats = torch.full((b, heads, L, 2), float("nan"), dtype=torch.float32, device=DEV)
    ws = _ws(max(lib.hn_attn_workspace_bytes(C.byref(params), has_ctx, ld, b, L, Nn, Dd),
                 lib.hn_attn_bwd_workspace_bytes(C.byref(params), has_ctx, ld, b, L, Nn, Dd, int(masked))))
    xo = torch.full_like(xd, float("nan"))
    zp = None if z is None else z.data_ptr()
    mp = None if m8 is None else m8.data_ptr()
    capi.check(lib.hn_attn_fwd_train(C.byref(params), xd.data_ptr(), xo.data_ptr(), int(residual), zp, ld, b, L, Nn, Dd, mp,
                                     stats.data_ptr(), saved.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "fwd_train")
    assert_close(xo.cpu(), y.detach(), rel=2e-4, what="attn.fwd_train")
    dx = torch.full_like(xd, float("nan"))
    capi.check(lib.hn_attn_bwd(C.byref(params), xd.data_ptr(), xo.data_ptr(), int(residual), zp, ld, b, L, Nn, Dd, mp,
                               stats.data_ptr(), saved.data_ptr(), dyd.data_ptr(), dx.data_ptr(), C.byref(grads), ws.data_ptr(),
                               ws.numel(), _stream()), "hn_attn_bwd")
    assert_close(dx.cpu(), x.grad, rel=5e-4, what="attn.dx")
    names = ["wq", "wkv", "wo", "bo"] + (["nw", "nb"] if norm else []) + (["cg", "cb"] if (norm and not self_attn) else [])
    for k in names:
        want = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
        assert_close(gr[k].cpu(), want, rel=5e-4, floor=2e-4, what="attn.d" + k)


ATTN_CASES = [
    dict(b=2, L=128, N=600, D=13, heads=8, dh=64, qd=128),                       # rank-D (image-like), several splits
    dict(b=2, L=25, N=77, D=18, heads=2, dh=63, qd=32),                          # rank-D dp=32, odd sizes
    dict(b=2, L=16, N=300, D=13, heads=2, dh=64, qd=32, masked=True),            # rank-D with a mask
    dict(b=2, L=128, N=512, D=96, heads=8, dh=64, qd=128),                       # explicit cross
    dict(b=3, L=17, N=65, D=40, heads=4, dh=27, qd=32, masked=True),             # explicit cross, padded head dim, mask
    dict(b=2, L=128, N=0, D=0, heads=8, dh=64, qd=128, self_attn=True),          # latent self-attention
    dict(b=2, L=24, N=0, D=0, heads=2, dh=16, qd=32, self_attn=True, norm=False, residual=False),
    dict(b=4, L=128, N=1, D=2005, heads=8, dh=64, qd=128),                       # one-token (tabular) context
    dict(b=2, L=16, N=50, D=13, heads=2, dh=16, qd=32, norm=False),              # bare Attention, raw 13-wide context
    dict(b=2, L=128, N=9000, D=200, heads=8, dh=64, qd=128),                     # explicit cross, 18 000-row contraction: LDS-staged
                                                                                 # weight-gradient GEMM with a ragged column tile
    # one narrow head over a long bag (the reference's tuned shapes): gemm_tall_narrow_kernel<2 / 4 / 8> for the K/V projection
    # (3000 rows: ragged against its 64-row tiles; K = 773 / 131: unaligned weight rows, a partial last quad; head padding 27 -> 32
    # through the column-group store; with and without the context affine), > 32 k-slices + the wide reduce for G = dKV^T z
    dict(b=3, L=25, N=1000, D=773, heads=1, dh=16, qd=32),
    dict(b=2, L=16, N=1100, D=131, heads=1, dh=27, qd=32),
    dict(b=2, L=16, N=1200, D=200, heads=2, dh=32, qd=32, norm=False),
    # LDS-DMA GEMMs (gemm_nt.hip) away from cfg4's shape: odd D with NaN-filled pad columns (the masked last k-step), a ragged last row
    # tile, 2 * inner = 256 columns; 4600-row contraction over 48 slices of 3 k-tiles for G = dKV^T z, 141 = 112 + 29 columns
    dict(b=2, L=128, N=2300, D=141, heads=2, dh=64, qd=128, nan_pad=True),
    dict(b=1, L=32, N=4200, D=773, heads=8, dh=64, qd=128, nan_pad=True),       # one bag of cfg4's width, ragged rows (4200 = 32 * 128 + 104)
    dict(b=2, L=16, N=2100, D=96, heads=4, dh=64, qd=32, norm=False),           # no context LayerNorm: the weight is staged without the affine
    # narrow-output LDS-DMA projections (gemm_nt variants 20-22) with the head width re-pitched by the weight staging: the
    # reference's tuned one-head TCGA shapes (dim_head 63 / 27 / 103 / 16)
    dict(b=2, L=17, N=1500, D=773, heads=1, dh=63, qd=19, nan_pad=True),
    dict(b=2, L=24, N=1100, D=131, heads=1, dh=103, qd=33, nan_pad=True),
    dict(b=1, L=16, N=2050, D=65, heads=2, dh=27, qd=32, masked=True),
    dict(b=2, L=20, N=1030, D=773, heads=1, dh=16, qd=26, nan_pad=True),
]
CASE_INDEX = {id(c): k for k, c in enumerate(ATTN_CASES)}


@pytest.mark.parametrize("kw", ATTN_CASES)
def test_attention_backward(capi, kw):
    # Fixed seed per case (NOT hash(): str hashes are randomised per process, which made the inputs -- and, at the kink of
    # LeakyReLU, the outcome -- differ from run to run).  LeakyReLU' jumps from 0.01 to 1 at pre = 0: when some
    # pre-activation is within rounding distance of 0, two correct fp32 forwards can disagree on its sign and their
    # gradients then differ by O(1) in that element (1-2 % of random inputs at these sizes have such an element:
    # tools/seed_scan.py).  _attn_case therefore advances the seed until every |pre| clears 1e-5.
    _attn_case(capi, seed=1000 + 17 * CASE_INDEX[id(kw)], **kw)


# ------------------------------------------------------------------------------------------------
# whole-model gradients: reference-generated fixtures (tiny models) and oracle autograd (larger ones)
# ------------------------------------------------------------------------------------------------
def test_model_gradients_match_reference_fixtures(manifest_):
    import healnet_amd as hn
    from conftest import load_golden
    for name in ["m1_d1", "m2_d3", "m3_d3", "m2_d3_tied", "m2_d2_noself", "m2_d2_nofourier", "m2_d2_gelu", "m2_d2_nohead",
                 "m2_d2_bands4"]:
        g = load_golden("g5_" + name)
        kw = manifest_["g5_" + name]["kwargs"]
        model = hn.HealNet(**kw).train()
        model.load_state_dict({k[4:]: v for k, v in g.items() if k.startswith("sd::")}, strict=True)
        model.to(DEV)
        ins = [g[f"in{i}"].to(DEV) for i in range(kw["n_modalities"])]
        out = model(list(ins))
        assert_close(out.detach().cpu(), g["logits"] if "logits" in g else out.detach().cpu(), rel=2e-4, what=name + ".fwd_train")
        (out * O.filler_input(out.shape, 77).to(DEV)).sum().backward()
        for k, p in model.named_parameters():
            want = g["grad::" + k]
            assert p.grad is not None, k
            assert_close(p.grad.cpu(), want, rel=1e-3, floor=5e-4, what=f"{name}.grad[{k}]")


@pytest.fixture(scope="module")
def manifest_():
    import json, os
    from conftest import GOLD
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


PACKED = {   # packed shared-context layouts of the training path: (channels, spatial shape, cross_dim_head) -> D, k-steps
    "packed_ks1": (4, (9,), 16, dict(fourier_encode_data=False)),   # D = 4 -> ks = 1
    "packed_ks2": (2, (40,), 64),         # D = 7  -> ks = 2
    "packed_ks5": (3, (3, 4, 5), 32),     # D = 18 -> dp = 32, ks = 5 (the volume modality)
    "packed_ks6": (9, (4, 3, 2), 32),     # D = 24 -> ks = 6
    "packed_ks7": (22, (11,), 32),        # D = 27 -> ks = 7
    "natural_d16": (11, (6,), 16),        # D = 16 == dp: no free column, natural layout, synthetic ones column
    "ks4_dp32": (11, (6,), 32),           # D = 16 on a 32-column row: packs into exactly 4 k-steps at inference, natural in training
    "ks4_dp32_d17": (12, (6,), 32),       # D = 17: same
    "packed_masked": (3, (6, 5), 64, dict(), True),                 # image-like D = 13 (ks = 3) with a key mask
}


@pytest.mark.parametrize("cfg", ["cfg1_b2", "cfg4_b2", "missing"] + sorted(PACKED))
def test_model_gradients_vs_oracle_autograd(cfg):
    import healnet_amd as hn
    gen = torch.Generator().manual_seed(31)
    if cfg in PACKED:
        chan, shape, dh = PACKED[cfg][:3]
        kw = dict(n_modalities=2, channel_dims=[5, chan], num_spatial_axes=[1, len(shape)], out_dims=3, depth=2, l_c=24, l_d=32,
                  x_heads=2, l_heads=2, cross_dim_head=dh, latent_dim_head=8, **(PACKED[cfg][3] if len(PACKED[cfg]) > 3 else {}))
        ins = [torch.rand(3, 2, 5, generator=gen), torch.rand(3, *shape, chan, generator=gen)]
        if len(PACKED[cfg]) > 4:      # one mask for every modality (Appendix B-5): both need the same token count
            n = 1
            for s_ in shape:
                n *= s_
            ins[0] = torch.rand(3, n, 5, generator=gen)
            mask = torch.rand(3, n, generator=gen) > 0.3
            mask[:, 0] = True
    elif cfg == "cfg1_b2":
        kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2)
        ins = [torch.rand(2, 1, 2000, generator=gen), torch.rand(2, 48, 40, 3, generator=gen)]
    elif cfg == "cfg4_b2":
        kw = dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4, depth=2)
        ins = [torch.rand(2, 1, 2000, generator=gen), torch.rand(2, 300, 768, generator=gen)]
    else:
        kw = dict(n_modalities=3, channel_dims=[50, 3, 64], num_spatial_axes=[1, 2, 1], out_dims=3, depth=2, l_c=32, l_d=64,
                  x_heads=4, l_heads=4, cross_dim_head=32, latent_dim_head=16)
        ins = [torch.rand(3, 1, 50, generator=gen), None, torch.rand(3, 90, 64, generator=gen)]
    torch.manual_seed(11)
    model = hn.HealNet(**kw).train()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    mask = locals().get("mask")
    want = O.fusion_forward(sd, O.FusionConfig(**kw), ins, mask=mask)
    dl = torch.randn(want.shape, generator=gen)
    (want * dl).sum().backward()
    model.to(DEV)
    got = model([None if t is None else t.to(DEV) for t in ins], mask=None if mask is None else mask.to(DEV))
    assert_close(got.detach().cpu(), want.detach(), rel=1e-3, what=cfg + ".fwd_train")
    (got * dl.to(DEV)).sum().backward()
    for k, p in model.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        assert_close(p.grad.cpu(), ref, rel=2e-3, floor=1e-3, what=f"{cfg}.grad[{k}]")
