"""GPU: dropout inside the path.  The masks are counter-based (Philox), so a test can export exactly the
masks a training forward used (hn_dropout_mask) and hand them to the CPU oracle, which restates nn.Dropout for a GIVEN mask:
forward and every parameter gradient are then compared like in the dropout-free tests; the statistics of the masks
themselves (keep rate, independence across blocks / calls) are checked separately."""
import ctypes as C

import pytest
import torch

from conftest import assert_close, rel_err
from oracle import healnet_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


def _mask(hn, p, seed, offset, stream, is_ff, rows, cols):
    from healnet_amd import _capi
    out = torch.empty(rows, cols, dtype=torch.uint8, device=DEV)
    rng = _capi.Rng(seed=seed & 0xFFFFFFFFFFFFFFFF, offset=offset, stream=stream)
    _capi.check(_capi.lib().hn_dropout_mask(float(p), rng, int(is_ff), rows, cols, out.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "hn_dropout_mask")
    return out


def _oracle_masks(hn, model, kw, b, n_tokens, seed, offset, present):
    """multipliers per executed block, numbered like build_schedule (api_train.hip) / oracle.fusion_forward"""
    M, L, d = kw["n_modalities"], kw["l_c"], kw["l_d"]
    pa, pf = kw.get("attn_dropout", 0.0), kw.get("ff_dropout", 0.0)
    drop, k = {}, 0
    for layer in range(kw["depth"]):
        for m in range(M):
            if present[m]:
                h = kw["x_heads"]
                if pa > 0:
                    drop[k] = (_mask(hn, pa, seed, offset, k, False, b * h * L, n_tokens[m]).float() / (1 - pa)).reshape(b * h, L, n_tokens[m]).cpu()
                k += 1
                if pf > 0:
                    drop[k] = (_mask(hn, pf, seed, offset, k, True, b * L, d).float() / (1 - pf)).reshape(b, L, d).cpu()
                k += 1
            if kw.get("self_per_cross_attn", 1) > 0:
                h = kw["l_heads"]
                if pa > 0:
                    drop[k] = (_mask(hn, pa, seed, offset, k, False, b * h * L, L).float() / (1 - pa)).reshape(b * h, L, L).cpu()
                k += 1
                if pf > 0:
                    drop[k] = (_mask(hn, pf, seed, offset, k, True, b * L, d).float() / (1 - pf)).reshape(b, L, d).cpu()
                k += 1
    return drop


CASES = [
    # tab (explicit binding, several tokens) + image (rank-D binding), both dropouts
    dict(kw=dict(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2, l_c=16, l_d=32, x_heads=2,
                 l_heads=2, cross_dim_head=16, latent_dim_head=16, attn_dropout=0.25, ff_dropout=0.1),
         shapes=[(9, 40), (11, 13, 3)]),
    # one-token tabular context (the shortcut path must be bypassed when dropping) + attention dropout only, GELU gate
    dict(kw=dict(n_modalities=2, channel_dims=[50, 3], num_spatial_axes=[1, 2], out_dims=3, depth=1, l_c=8, l_d=16, x_heads=2,
                 l_heads=2, cross_dim_head=8, latent_dim_head=8, attn_dropout=0.4, ff_dropout=0.0, snn=False),
         shapes=[(1, 50), (6, 7, 3)]),
    # D = 16 exactly on a 16-column row (no spare column for the thinned row sum): the modality takes the explicit binding
    dict(kw=dict(n_modalities=1, channel_dims=[11], num_spatial_axes=[1], out_dims=3, depth=2, l_c=8, l_d=16, x_heads=2,
                 l_heads=2, cross_dim_head=16, latent_dim_head=8, attn_dropout=0.3, ff_dropout=0.1),
         shapes=[(6, 11)]),
    # feed-forward dropout only, no latent self blocks
    dict(kw=dict(n_modalities=1, channel_dims=[3], num_spatial_axes=[2], out_dims=2, depth=2, l_c=8, l_d=16, x_heads=2,
                 l_heads=2, cross_dim_head=8, latent_dim_head=8, attn_dropout=0.0, ff_dropout=0.3, self_per_cross_attn=0),
         shapes=[(10, 9, 3)]),
    # the reference's tuned TCGA shape class (config/best_hyperparams.yml: odd latent width, ONE cross head of an odd dim, no
    # latent self-attention, both dropouts): l_d = 30 is not a multiple of 4 -> the feed-forward mask's last column quad is partial
    dict(kw=dict(n_modalities=2, channel_dims=[37, 21], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=17, l_d=30, x_heads=1,
                 l_heads=8, cross_dim_head=27, latent_dim_head=11, attn_dropout=0.3, ff_dropout=0.25, self_per_cross_attn=0),
         shapes=[(1, 37), (50, 21)]),
    dict(kw=dict(n_modalities=1, channel_dims=[21], num_spatial_axes=[1], out_dims=4, depth=1, l_c=16, l_d=65, x_heads=1,
                 l_heads=8, cross_dim_head=103, latent_dim_head=51, attn_dropout=0.25, ff_dropout=0.06, self_per_cross_attn=0),
         shapes=[(33, 21)]),
    # l_c = 64: the (b, h) blocks start on multiples of 64 rows -> the cores' four query tiles share ONE generator call
    # (common.h drop_rows4: windows 0 .. 3); both bindings (tabular explicit, image rank-D)
    dict(kw=dict(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=4, depth=1, l_c=64, l_d=32, x_heads=2,
                 l_heads=2, cross_dim_head=16, latent_dim_head=16, attn_dropout=0.3, ff_dropout=0.1),
         shapes=[(9, 40), (11, 13, 3)]),
    # l_c = 32: every other block starts at 32 mod 64 -> tile pairs on windows (2, 3) of their group's call (drop_pair)
    dict(kw=dict(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=4, depth=1, l_c=32, l_d=32, x_heads=3,
                 l_heads=2, cross_dim_head=16, latent_dim_head=16, attn_dropout=0.2, ff_dropout=0.0),
         shapes=[(9, 40), (11, 13, 3)]),
]


@pytest.mark.parametrize("case", CASES)
def test_training_forward_and_gradients_under_the_exported_masks(hn, case):
    kw, shapes = case["kw"], case["shapes"]
    torch.manual_seed(21)
    model = hn.HealNet(**kw).train()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = O.FusionConfig(**{k: v for k, v in kw.items()})
    gen = torch.Generator().manual_seed(8)
    b = 3
    ins = [torch.rand(b, *s, generator=gen) for s in shapes]
    n_tokens = [int(torch.tensor(s[:-1]).prod()) for s in shapes]
    target = torch.randn(b, kw["out_dims"], generator=gen)
    model.to(DEV)
    y = model([t.to(DEV) for t in ins])
    seed, offset = model._last_rng
    loss = ((y - target.to(DEV)) ** 2).sum()
    loss.backward()

    drop = _oracle_masks(hn, model, kw, b, n_tokens, seed, offset, [True] * kw["n_modalities"])
    cpu = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yc = O.fusion_forward(cpu, cfg, ins, drop=drop)
    assert_close(y.detach().cpu(), yc.detach(), rel=2e-4, what="logits under dropout")
    ((yc - target) ** 2).sum().backward()
    named = dict(model.named_parameters())
    for k, v in cpu.items():
        if v.grad is None:
            continue
        g = named[k].grad
        assert g is not None, k
        if float(v.grad.abs().max()) < 1e-12:       # exactly zero in exact arithmetic (queries / keys of a one-token softmax)
            assert float(g.abs().max()) < 1e-5, k
            continue
        assert rel_err(g.cpu(), v.grad) <= 5e-4, (k, rel_err(g.cpu(), v.grad))

    # a second forward draws fresh masks (offset advances); eval mode draws none and equals the dropout-free model
    y2 = model([t.to(DEV) for t in ins])
    assert model._last_rng[1] == offset + 1 and not torch.equal(y2, y)
    model.eval()
    with torch.no_grad():
        ye = model([t.to(DEV) for t in ins])
    assert_close(ye.cpu(), O.fusion_forward(sd, cfg, ins), rel=2e-4, what="eval mode: no dropout")


def test_dropout_in_train_mode_applies_under_no_grad_too(hn):
    kw = CASES[0]["kw"]
    torch.manual_seed(3)
    model = hn.HealNet(**kw).train().to(DEV)
    ins = [torch.rand(2, *s, device=DEV) for s in CASES[0]["shapes"]]
    with torch.no_grad():
        a, b_ = model(list(ins)), model(list(ins))
    assert not torch.equal(a, b_)                     # nn.Dropout is active in training mode whatever the grad mode


def test_mask_statistics(hn):
    p = 0.3
    m0 = _mask(hn, p, 1234, 1, 5, False, 4096, 1000).float()
    assert abs(float(m0.mean()) - (1 - p)) < 2e-3
    # different stream / offset / seed / ff-bit -> different, uncorrelated masks
    for (seed, off, sid, ff) in [(1234, 1, 6, False), (1234, 2, 5, False), (99, 1, 5, False), (1234, 1, 5, True)]:
        m1 = _mask(hn, p, seed, off, sid, ff, 4096, 1000).float()
        agree = float((m0 == m1).float().mean())
        assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 3e-3, (seed, off, sid, ff, agree)
    # neighbouring rows / columns are uncorrelated
    c = float(((m0[:, 1:] - (1 - p)) * (m0[:, :-1] - (1 - p))).mean())
    r = float(((m0[1:] - (1 - p)) * (m0[:-1] - (1 - p))).mean())
    assert abs(c) < 2e-3 and abs(r) < 2e-3
    # attention masks: rows 16 / 32 / 48 apart share a generator call (16 decisions of 16 bits from overlapping windows of the
    # call's 128 bits, common.h): exact marginals, covariance of window neighbours <= 2^-10 by construction
    for d in (16, 32, 48, 64):
        r16 = float(((m0[d:] - (1 - p)) * (m0[:-d] - (1 - p))).mean())
        assert abs(r16) < 2e-3, (d, r16)
    # the rate keeps its 2^-16 resolution: p = 0.3 + 1/512 is told apart from 0.3 (8-bit decisions could not)
    for pq in (0.3 + 1.0 / 512, 0.05, 0.7):
        mq = _mask(hn, pq, 77, 3, 9, False, 4096, 2048).float()
        assert abs(float(mq.mean()) - (1 - pq)) < 7e-4, (pq, float(mq.mean()))
        for k in range(4):          # every window position on its own
            rows = torch.arange(4096).div(16, rounding_mode="floor").remainder(4) == k
            assert abs(float(mq[rows].mean()) - (1 - pq)) < 1.2e-3, (pq, k)
    assert float(_mask(hn, 0.0, 1, 1, 1, False, 8, 8).float().mean()) == 1.0


def test_expected_output_matches_the_dropout_free_model(hn):
    """E[dropout(x)] = x: averaging training-mode outputs of a LINEAR probe of the masks (attention output before the
    non-linearity is not observable through the ABI, so use a large sample and a loose bound on the logits)."""
    kw = dict(n_modalities=1, channel_dims=[3], num_spatial_axes=[2], out_dims=4, depth=1, l_c=8, l_d=16, x_heads=2, l_heads=2,
              cross_dim_head=8, latent_dim_head=8, attn_dropout=0.2, ff_dropout=0.0, self_per_cross_attn=0)
    torch.manual_seed(5)
    model = hn.HealNet(**kw).train().to(DEV)
    img = torch.rand(4, 16, 16, 3, device=DEV)
    with torch.no_grad():
        acc = torch.zeros(4, 4, device=DEV)
        n = 300
        for _ in range(n):
            acc += model([img])
        model.eval()
        ref = model([img])
    assert rel_err(acc / n, ref) < 5e-2


def test_general_dropout_core_route_still_matches(hn):
    """HN_NO_DROP_BOUND=1 (subprocess): dropping modalities keep the natural context layout and the general softmax path of the
    core -- the route every rank-D dropout block took before the bounded / packed variants; same parity cases, same bounds."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HN_NO_DROP_BOUND="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_dropout.py"), "-q", "-x", "-m", "gpu",
                        "-k", "exported_masks"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------
# stand-alone modules: Attention(query_dim, context_dim, heads, dim_head, dropout) / FeedForward(dim, dropout=)
# are part of the boundary (healnet.py:369-381, :339-347); in training mode they drop like the reference's nn.Dropout
# ------------------------------------------------------------------------------------------------
def _mult(hn, mod, p, is_ff, rows, cols):
    seed, offset, stream = mod._last_rng
    return _mask(hn, p, seed, offset, stream, is_ff, rows, cols).float().cpu() / (1 - p)


def _grads_close(named, cpu, what):
    for k, v in cpu.items():
        if v.grad is None:
            continue
        g = named[k].grad
        assert g is not None, (what, k)
        assert rel_err(g.cpu(), v.grad) <= 5e-4, (what, k, rel_err(g.cpu(), v.grad))


@pytest.mark.parametrize("case", [
    dict(qd=32, cd=21, heads=2, dh=16, L=16, N=37, b=3, p=0.25, norm_ctx=True),       # cross block, LayerNorm-ed context (rank-D binding refused: D > 15 -> explicit)
    dict(qd=32, cd=13, heads=4, dh=16, L=16, N=50, b=2, p=0.3, norm_ctx=True),        # narrow context: the shared-context (rank-D) binding
    dict(qd=48, cd=None, heads=3, dh=16, L=24, N=None, b=2, p=0.2, norm_ctx=False),   # self-attention
])
def test_standalone_attention_drops_with_the_exported_mask(hn, case):
    c = case
    torch.manual_seed(31)
    blk = hn.PreNorm(c["qd"], hn.Attention(c["qd"], c["cd"], heads=c["heads"], dim_head=c["dh"], dropout=c["p"]),
                     context_dim=c["cd"] if c["norm_ctx"] else None).train()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
    gen = torch.Generator().manual_seed(32)
    x = torch.randn(c["b"], c["L"], c["qd"], generator=gen)
    ctx = None if c["cd"] is None else torch.randn(c["b"], c["N"], c["cd"], generator=gen)
    target = torch.randn(c["b"], c["L"], c["qd"], generator=gen)
    blk.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = blk(xg, context=None if ctx is None else ctx.to(DEV))
    ((y - target.to(DEV)) ** 2).sum().backward()
    n = c["L"] if ctx is None else c["N"]
    dm = _mult(hn, blk.fn, c["p"], False, c["b"] * c["heads"] * c["L"], n).reshape(c["b"] * c["heads"], c["L"], n)
    xc = x.clone().requires_grad_(True)
    xn = O.layer_norm(xc, sd["norm.weight"], sd["norm.bias"])
    cn = None if ctx is None else (O.layer_norm(ctx, sd["norm_context.weight"], sd["norm_context.bias"]) if c["norm_ctx"] else ctx)
    yc = O.attention(xn, cn, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"], c["heads"],
                     drop_mult=dm)
    assert_close(y.detach().cpu(), yc.detach(), rel=2e-4, what="stand-alone attention under dropout")
    ((yc - target) ** 2).sum().backward()
    assert rel_err(xg.grad.cpu(), xc.grad) <= 5e-4
    _grads_close(dict(blk.named_parameters()), sd, "attention")
    # the probabilities the module reports are the UN-thinned ones (healnet.py:420 stores attn before the dropout)
    pw = blk.fn.attn_weights
    assert float((pw.sum(-1) - 1).abs().max()) < 1e-5
    # a second call draws a fresh mask; eval mode draws none
    y2 = blk(xg.detach(), context=None if ctx is None else ctx.to(DEV))
    assert not torch.equal(y2, y.detach())
    blk.eval()
    with torch.no_grad():
        ye = blk(xg.detach(), context=None if ctx is None else ctx.to(DEV))
        want = O.attention(xn, cn, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"], c["heads"])
    assert_close(ye.cpu(), want.detach(), rel=2e-4, what="stand-alone attention, eval mode")


@pytest.mark.parametrize("dim,snn,p", [(32, True, 0.3), (30, False, 0.15)])
def test_standalone_feed_forward_drops_with_the_exported_mask(hn, dim, snn, p):
    torch.manual_seed(41)
    blk = hn.PreNorm(dim, hn.FeedForward(dim, dropout=p, snn=snn)).train()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
    gen = torch.Generator().manual_seed(42)
    x = torch.randn(3, 10, dim, generator=gen)
    target = torch.randn(3, 10, dim, generator=gen)
    blk.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = blk(xg)
    ((y - target.to(DEV)) ** 2).sum().backward()
    dm = _mult(hn, blk.fn, p, True, 30, dim).reshape(3, 10, dim)
    xc = x.clone().requires_grad_(True)
    yc = O.feed_forward(O.layer_norm(xc, sd["norm.weight"], sd["norm.bias"]), sd["fn.net.0.weight"], sd["fn.net.0.bias"], sd["fn.net.2.weight"],
                        sd["fn.net.2.bias"], snn=snn, drop_mult=dm)
    assert_close(y.detach().cpu(), yc.detach(), rel=2e-4, what="stand-alone feed-forward under dropout")
    ((yc - target) ** 2).sum().backward()
    assert rel_err(xg.grad.cpu(), xc.grad) <= 5e-4
    _grads_close(dict(blk.named_parameters()), sd, "feed_forward")
    blk.eval()
    with torch.no_grad():
        ye = blk(xg.detach())
    assert_close(ye.cpu(), O.feed_forward(O.layer_norm(x, sd["norm.weight"], sd["norm.bias"]), sd["fn.net.0.weight"], sd["fn.net.0.bias"],
                                          sd["fn.net.2.weight"], sd["fn.net.2.bias"], snn=snn).detach(), rel=2e-4, what="feed-forward, eval mode")


def test_latent_block_drops_with_the_exported_masks(hn):
    d, heads, dh, b, L, pa, pf = 128, 8, 64, 2, 32, 0.2, 0.1
    torch.manual_seed(51)
    att = hn.PreNorm(d, hn.Attention(d, heads=heads, dim_head=dh, dropout=pa)).train()
    ff = hn.PreNorm(d, hn.FeedForward(d, dropout=pf, snn=True)).train()
    sa = {k: v.detach().clone().requires_grad_(True) for k, v in att.state_dict().items()}
    sf = {k: v.detach().clone().requires_grad_(True) for k, v in ff.state_dict().items()}
    gen = torch.Generator().manual_seed(52)
    x = torch.randn(b, L, d, generator=gen)
    target = torch.randn(b, L, d, generator=gen)
    att.to(DEV), ff.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    y = hn.latent_block(att, ff, xg)
    ((y - target.to(DEV)) ** 2).sum().backward()
    dma = _mult(hn, att.fn, pa, False, b * heads * L, L).reshape(b * heads, L, L)
    dmf = _mult(hn, ff.fn, pf, True, b * L, d).reshape(b, L, d)
    xc = x.clone().requires_grad_(True)
    x1 = O.attention(O.layer_norm(xc, sa["norm.weight"], sa["norm.bias"]), None, sa["fn.to_q.weight"], sa["fn.to_kv.weight"], sa["fn.to_out.0.weight"],
                     sa["fn.to_out.0.bias"], heads, drop_mult=dma) + xc
    yc = O.feed_forward(O.layer_norm(x1, sf["norm.weight"], sf["norm.bias"]), sf["fn.net.0.weight"], sf["fn.net.0.bias"], sf["fn.net.2.weight"],
                        sf["fn.net.2.bias"], snn=True, drop_mult=dmf) + x1
    assert_close(y.detach().cpu(), yc.detach(), rel=2e-4, what="latent block under dropout")
    ((yc - target) ** 2).sum().backward()
    assert rel_err(xg.grad.cpu(), xc.grad) <= 5e-4
    _grads_close(dict(att.named_parameters()), sa, "latent block attention")
    _grads_close(dict(ff.named_parameters()), sf, "latent block feed-forward")
