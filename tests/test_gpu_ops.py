"""GPU: per-op parity of the HIP kernels (through the C ABI) against the committed golden fixtures
(generated from the reference) and against the CPU oracle on seeded inputs."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close, load_golden, rel_err
from oracle import healnet_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


# ---------------------------------------------------------------------------------------------- K1
def test_encode_matches_reference_fixtures(hn):
    g = load_golden("g2_preprocess")
    for name in ("tab", "img", "vol", "one"):
        got = hn.fourier_encode_concat(g[name + "_in"].to(DEV)).cpu()
        assert_close(got, g[name + "_enc"], rel=2e-6, what=f"encode[{name}]")


@pytest.mark.parametrize("shape,bands,maxf", [((2, 224, 224, 3), 2, 10.0), ((3, 1, 2000), 2, 10.0),
                                              ((2, 257, 768), 2, 10.0), ((1, 5, 7, 9, 2), 4, 2.0),
                                              ((2, 3, 4, 5, 6, 2), 2, 10.0), ((1, 1, 1), 1, 10.0)])
def test_encode_vs_oracle(hn, shape, bands, maxf):
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(3))
    want = O.encode_modality(x, bands, maxf, True)
    got = hn.fourier_encode_concat(x.to(DEV), bands, maxf).cpu()
    assert_close(got, want, rel=3e-6, what=f"encode{shape}")
    got2 = hn.fourier_encode_concat(x.to(DEV), bands, maxf, fourier_encode_data=False).cpu()
    assert torch.equal(got2, x.reshape(shape[0], -1, shape[-1]))


def test_encode_norm_is_affine_free_layernorm(hn):
    from healnet_amd.healnet import _normalise_context
    for shape in [(2, 100, 13), (2, 7, 773), (3, 1, 2005), (1, 33, 18)]:
        ctx = torch.randn(*shape, generator=torch.Generator().manual_seed(1)) * 3 + 1
        pitch = {13: 16, 18: 32, 773: 776, 2005: 2008}[shape[-1]]
        z = _normalise_context(ctx.to(DEV), pitch).cpu()
        want = F.layer_norm(ctx, (shape[-1],))
        assert_close(z[..., : shape[-1]], want, rel=2e-5, what=f"norm{shape}")
        assert (z[..., shape[-1]:] == 0).all()


# ------------------------------------------------------------------------------------------ attention
def _attention_from_golden(hn, g, name, c):
    att = hn.Attention(c["qd"], c["cd"], heads=c["heads"], dim_head=c["dh"]).to(DEV)
    with torch.no_grad():
        att.to_q.weight.copy_(g[name + "_wq"])
        att.to_kv.weight.copy_(g[name + "_wkv"])
        att.to_out[0].weight.copy_(g[name + "_wo"])
        att.to_out[0].bias.copy_(g[name + "_bo"])
    return att


def test_attention_matches_reference_fixtures(hn, manifest):
    g = load_golden("g3_attention")
    for name, c in manifest["g3_attention"]["cases"].items():
        att = _attention_from_golden(hn, g, name, c)
        ctx = g.get(name + "_ctx")
        mask = g.get(name + "_mask")
        y = att(g[name + "_x"].to(DEV), context=None if ctx is None else ctx.to(DEV),
                mask=None if mask is None else mask.to(DEV))
        assert_close(y.cpu(), g[name + "_y"], rel=1e-4, what=f"attention[{name}].y")
        p = att.attn_weights
        assert p.shape == g[name + "_p"].shape
        assert_close(p.cpu(), g[name + "_p"], rel=1e-4, what=f"attention[{name}].p")


@pytest.mark.parametrize("b,L,N,D,heads,dh", [(2, 128, 5000, 13, 8, 64), (1, 128, 777, 18, 8, 64), (2, 25, 300, 96, 1, 63),
                                              (2, 17, 65, 40, 4, 27), (1, 128, 4096, 773, 8, 64), (3, 128, 1, 2005, 8, 64),
                                              (2, 16, 33, 12, 2, 103), (2, 256, 1, 2189, 8, 64),
                                              (1, 25, 9000, 13, 1, 63), (1, 25, 9100, 96, 1, 63)])   # > 64 token splits
def test_prenorm_cross_attention_vs_oracle(hn, b, L, N, D, heads, dh):
    gen = torch.Generator().manual_seed(b * 1000 + N)
    qd = 128 if L == 128 else 32
    blk = hn.PreNorm(qd, hn.Attention(qd, D, heads=heads, dim_head=dh), context_dim=D).to(DEV)
    with torch.no_grad():
        for p_ in blk.parameters():
            if p_.dim() == 1:
                p_.add_(0.2 * torch.randn(p_.shape, generator=gen).to(DEV))
        blk.fn.to_q.weight.mul_(2.0)
        blk.fn.to_kv.weight.mul_(2.0)
    x = torch.randn(b, L, qd, generator=gen)
    ctx = torch.rand(b, N, D, generator=gen) * 2
    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    xn = O.layer_norm(x, sd["norm.weight"], sd["norm.bias"])
    cn = O.layer_norm(ctx, sd["norm_context.weight"], sd["norm_context.bias"])
    want, pw = O.attention(xn, cn, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"],
                           sd["fn.to_out.0.bias"], heads, return_weights=True)
    got = blk(x.to(DEV), context=ctx.to(DEV))
    assert_close(got.cpu(), want, rel=2e-4, what="cross.y")
    if N <= 5000:
        assert_close(blk.fn.attn_weights.cpu(), pw, rel=5e-4, what="cross.p")


def test_forced_softmax_rescale_branch(hn):
    """One key far above the rest, placed late in the token stream: the lazy-rescale branch of the
    split-KV kernel must fire mid-stream (cdna guide §5.4 rule 26)."""
    gen = torch.Generator().manual_seed(11)
    b, L, N, D, heads, dh = 1, 32, 3000, 13, 2, 64
    att = hn.Attention(16, D, heads=heads, dim_head=dh).to(DEV)
    x = torch.randn(b, L, 16, generator=gen)
    ctx = torch.randn(b, N, D, generator=gen) * 0.2
    ctx[0, 2500] = 25.0 * torch.sign(torch.randn(D, generator=gen))
    ctx[0, 100] = -18.0
    with torch.no_grad():
        att.to_q.weight.mul_(4.0)
    sd = {k: v.detach().cpu() for k, v in att.state_dict().items()}
    want, pw = O.attention(x, ctx, sd["to_q.weight"], sd["to_kv.weight"], sd["to_out.0.weight"], sd["to_out.0.bias"], heads,
                           return_weights=True)
    assert pw.max() > 0.5          # the spike really dominates some rows
    ctx16 = torch.zeros(b, N, 16)
    ctx16[..., :D] = ctx
    got = att(x.to(DEV), context=ctx.to(DEV))
    assert_close(got.cpu(), want, rel=2e-4, what="spike.y")


@pytest.mark.parametrize("b,L,heads,dh", [(4, 128, 8, 64), (3, 40, 2, 64), (2, 100, 5, 64), (1, 16, 1, 64), (2, 1, 3, 64),
                                          (2, 160, 2, 64), (2, 48, 4, 32)])
def test_self_attention_block_vs_oracle(hn, b, L, heads, dh, monkeypatch):
    """Latent self-attention (healnet.py:241-245).  dim_head 64 with L <= 128 runs on self_core_lds_kernel (one workgroup per
    (sample, head), K / V in LDS): full and ragged row counts (40, 100: masked token tail + idle waves), a single row, one
    head; L = 160 and dim_head 32 stay on the split-KV core.  Output and the probabilities hn_attn_probs rebuilds from the
    kernel's (max, sum) statistics.  (The kernel is only chosen from 192 (sample, head) pairs on: forced here.)"""
    monkeypatch.setenv("HN_FORCE_SELF_LDS", "1")
    gen = torch.Generator().manual_seed(5 + L)
    blk = hn.PreNorm(128, hn.Attention(128, heads=heads, dim_head=dh)).to(DEV)
    x = torch.randn(b, L, 128, generator=gen) * 1.5
    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    xn = O.layer_norm(x, sd["norm.weight"], sd["norm.bias"])
    want, pw = O.attention(xn, None, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"],
                           heads, return_weights=True)
    got = blk(x.to(DEV))
    assert_close(got.cpu(), want, rel=2e-4, what="self.y")
    assert_close(blk.fn.attn_weights.cpu(), pw, rel=5e-4, what="self.p")


def test_fully_masked_row_is_nan_like_reference(hn):
    att = hn.Attention(16, 8, heads=2, dim_head=8).to(DEV)
    x = torch.randn(1, 4, 16)
    ctx = torch.randn(1, 6, 8)
    mask = torch.zeros(1, 6, dtype=torch.bool)
    y = att(x.to(DEV), context=ctx.to(DEV), mask=mask.to(DEV))
    assert torch.isnan(y).all()


# ------------------------------------------------------------------------------------------ feed-forward / head
def test_feedforward_matches_reference_fixtures(hn):
    g = load_golden("g4_feedforward")
    for tag, snn in (("selu", True), ("gelu", False)):
        ffn = hn.FeedForward(16, snn=snn).to(DEV)
        with torch.no_grad():
            ffn.net[0].weight.copy_(g[tag + "_w1"]); ffn.net[0].bias.copy_(g[tag + "_b1"])
            ffn.net[2].weight.copy_(g[tag + "_w2"]); ffn.net[2].bias.copy_(g[tag + "_b2"])
        assert_close(ffn(g[tag + "_x"].to(DEV)).cpu(), g[tag + "_y"], rel=1e-5, what="ff." + tag)


@pytest.mark.parametrize("dim,rows", [(128, 4096), (119, 75), (16, 1)])
def test_prenorm_feedforward_vs_oracle(hn, dim, rows):
    gen = torch.Generator().manual_seed(dim)
    blk = hn.PreNorm(dim, hn.FeedForward(dim, snn=True)).to(DEV)
    x = torch.randn(1, rows, dim, generator=gen) * 2
    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    want = O.feed_forward(O.layer_norm(x, sd["norm.weight"], sd["norm.bias"]), sd["fn.net.0.weight"], sd["fn.net.0.bias"],
                          sd["fn.net.2.weight"], sd["fn.net.2.bias"], True)
    assert_close(blk(x.to(DEV)).cpu(), want, rel=1e-4, what="prenorm.ff")


# ------------------------------------------------------------------------------------------ torch.ops surface
def test_torch_ops_surface(hn):
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(2, 6, 5, 3, generator=gen)
    got = torch.ops.healnet_hip.fourier_encode_concat(x.to(DEV), 2, 10.0, True)
    assert_close(got.cpu(), O.encode_modality(x, 2, 10.0), rel=3e-6, what="ops.encode")
    z = torch.ops.healnet_hip.encode_norm(x.to(DEV), 2, 10.0, True, 16)
    assert_close(z[..., :13].cpu(), F.layer_norm(O.encode_modality(x, 2, 10.0), (13,)), rel=2e-5, what="ops.encode_norm")
    # attention + feed-forward + head chained exactly like one (cross, ff) step followed by the head
    d, heads, dh = 32, 2, 16
    lat = torch.randn(2, 8, d, generator=gen)
    p = {k: torch.randn(*s, generator=gen) * 0.3 for k, s in dict(nw=(d,), nb=(d,), cg=(13,), cb=(13,), wq=(heads * dh, d),
         wkv=(2 * heads * dh, 13), wo=(d, heads * dh), bo=(d,), w1=(8 * d, d), b1=(8 * d,), w2=(d, 4 * d), b2=(d,),
         hw=(3, d), hb=(3,)).items()}
    P = {k: v.to(DEV) for k, v in p.items()}
    y = torch.ops.healnet_hip.attention(lat.to(DEV), z, None, P["nw"], P["nb"], P["cg"], P["cb"], P["wq"], P["wkv"], P["wo"],
                                        P["bo"], heads, True)
    y = torch.ops.healnet_hip.feed_forward(y, P["nw"], P["nb"], P["w1"], P["b1"], P["w2"], P["b2"], False, True)
    logits = torch.ops.healnet_hip.head(y, P["nw"], P["nb"], P["hw"], P["hb"])
    ctx = O.encode_modality(x, 2, 10.0)
    want = O.attention(O.layer_norm(lat, p["nw"], p["nb"]), O.layer_norm(ctx, p["cg"], p["cb"]), p["wq"], p["wkv"], p["wo"],
                       p["bo"], heads) + lat
    want = O.feed_forward(O.layer_norm(want, p["nw"], p["nb"]), p["w1"], p["b1"], p["w2"], p["b2"], True) + want
    want = O.layer_norm(want.mean(1), p["nw"], p["nb"]) @ p["hw"].t() + p["hb"]
    assert_close(logits.cpu(), want, rel=1e-4, what="ops.chain")


def test_one_token_context_paths_agree(hn):
    """N == 1 takes the degenerate fast path (P == 1); with an all-true mask the general split-KV path runs."""
    gen = torch.Generator().manual_seed(21)
    blk = hn.PreNorm(128, hn.Attention(128, 2005, heads=8, dim_head=64), context_dim=2005).to(DEV)
    x = torch.randn(5, 128, 128, generator=gen).to(DEV)
    ctx = torch.rand(5, 1, 2005, generator=gen).to(DEV)
    fast = blk(x, context=ctx)
    p_fast = blk.fn.attn_weights
    general = blk(x, context=ctx, mask=torch.ones(5, 1, dtype=torch.bool, device=DEV))
    p_general = blk.fn.attn_weights
    assert_close(fast.cpu(), general.cpu(), rel=1e-5, what="one-token fast vs general")
    assert torch.equal(p_fast, torch.ones_like(p_fast)) and torch.allclose(p_general, p_fast, atol=1e-6)


def test_temperature_softmax_matches_reference_function(hn):
    """temperature_softmax(logits, temperature, dim) as a stand-alone op, against outputs of the reference's
    own function (tests/golden/g8_temperature_softmax.npz)."""
    g = load_golden("g8_temperature_softmax")
    for i in range(4):
        x, want, t = g[f"x{i}"].to(DEV), g[f"y{i}"], float(g[f"t{i}"])
        got = hn.temperature_softmax(x, temperature=t)
        assert_close(got.cpu(), want, rel=2e-6, floor=1e-7, what=f"temperature_softmax[{i}]")
        assert_close(torch.ops.healnet_hip.temperature_softmax(x, t).cpu(), want, rel=2e-6, floor=1e-7, what="torch.ops")
    got = hn.temperature_softmax(g["xd"].to(DEV), temperature=0.5, dim=1)
    assert_close(got.cpu(), g["yd"], rel=2e-6, floor=1e-7, what="temperature_softmax dim=1")


def test_fourier_encode_function_matches_reference_fixtures(hn):
    """a3 as a stand-alone function: every (S, max_freq, bands) fixture of tests/golden/g1_fourier.npz, bit for bit except
    sin/cos at huge arguments (device sinf/cosf vs ATen: <= 2 ulp)."""
    g = load_golden("g1_fourier")
    for key, want in g.items():
        S, mf, nb = key.split("_")
        S, mf, nb = int(S[1:]), float(mf[2:]), int(nb[2:])
        pos = torch.linspace(-1.0, 1.0, S)[:, None].to(DEV)
        got = hn.fourier_encode(pos, mf, nb)
        assert got.shape == want.shape
        assert_close(got.cpu(), want, rel=2e-6, floor=2e-7, what="fourier_encode " + key)


def test_gate_modules_match_torch(hn):
    x = torch.randn(5, 7, 64, device=DEV) * 2
    a, g_ = x.chunk(2, dim=-1)
    assert_close(hn.SELU()(x).cpu(), (a * F.selu(g_)).cpu(), rel=2e-6, floor=1e-7, what="SELU gate")
    assert_close(hn.GELU()(x).cpu(), (a * F.gelu(g_)).cpu(), rel=2e-6, floor=1e-7, what="GELU gate")


@pytest.mark.parametrize("qscale", [1.0, 6.0, 60.0])
def test_bounded_softmax_reference_and_its_fallback(hn, qscale):
    """Shared-context binding with a LayerNorm-ed context: the core takes the Cauchy-Schwarz score bound |q| sqrt(D) of each
    row as its softmax reference (no running max, no overflow guard).  qscale = 1 / 6 stays inside the usable range (bounds
    of a few / a few dozen log2 units: probabilities as small as 2^-100 against the reference), qscale = 60 pushes rows
    past the limit, which must flip the launch back to the running reference -- all against the oracle."""
    gen = torch.Generator().manual_seed(17)
    b, L, N, D, heads, dh, qd = 2, 48, 1500, 13, 4, 64, 64
    blk = hn.PreNorm(qd, hn.Attention(qd, D, heads=heads, dim_head=dh), context_dim=D).eval().to(DEV)
    with torch.no_grad():
        blk.fn.to_q.weight.mul_(qscale)
        blk.norm_context.weight.copy_(1 + 0.3 * torch.randn(D, generator=gen))
        blk.norm_context.bias.copy_(0.2 * torch.randn(D, generator=gen))
    x = torch.randn(b, L, qd, generator=gen)
    ctx = torch.rand(b, N, D, generator=gen)
    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    xn = O.layer_norm(x, sd["norm.weight"], sd["norm.bias"])
    cn = O.layer_norm(ctx, sd["norm_context.weight"], sd["norm_context.bias"])
    want, pw = O.attention(xn, cn, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"],
                           heads, return_weights=True)
    with torch.no_grad():
        got = blk(x.to(DEV), context=ctx.to(DEV))
    assert_close(got.cpu(), want, rel=3e-4, what=f"bounded softmax qscale={qscale}")
    assert_close(blk.fn.attn_weights.cpu(), pw, rel=1e-3, floor=1e-5, what="attn_weights from the bound-referenced statistics")
    if qscale >= 60:
        assert pw.max() > 0.9          # genuinely saturated rows


def test_integration_md_ctypes_stub_is_valid(hn):
    """The ctypes binding INTEGRATION.md shows a reference maintainer (struct mirror + hn_attn_fwd call) is executed as
    written and must reproduce the package's own Attention.forward."""
    import os, re, types
    from healnet_amd import _capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C, torch\n.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("libhealnet_hip.so")', f'C.CDLL({_capi.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    torch.manual_seed(12)
    att = hn.Attention(32, 13, heads=2, dim_head=16).to(DEV)
    x, ctx = torch.randn(3, 8, 32, device=DEV), torch.rand(3, 50, 13, device=DEV)
    with torch.no_grad():
        want = att(x, context=ctx)
        got = types.MethodType(ns["forward"], att)(x, context=ctx)
        assert_close(got.cpu(), want.cpu(), rel=1e-6, what="INTEGRATION.md stub (cross)")
        assert_close(types.MethodType(ns["forward"], att.__class__(32, heads=2, dim_head=16).to(DEV))(x).cpu().isfinite().float(),
                     torch.ones(3, 8, 32), rel=0, what="INTEGRATION.md stub (self) finite")
