"""GPU: the fused latent chain (chain.hip; SURVEY.md 8(b) hn_latent_block_fwd / _bwd).

  * hn_latent_block_fwd (chain: LN + Q|K|V -> attention core -> chain: out-projection + LeakyReLU + residual + LN + gated FF +
    residual) against the oracle's _self_block + _ff_block (healnet.py:241-245), SELU and GELU gates, several head shapes,
    incl. shapes that fall back to the unfused launches;
  * its registered backward (hn_latent_block_bwd) against oracle autograd;
  * the whole model with the chain vs the unfused launch sequence (HN_NO_CHAIN=1 in a subprocess): same logits to fp32 noise,
    same attention weights -- the chain also absorbs the cross blocks' out-projection / query projection and the one-token
    broadcast add there.
"""
import os
import subprocess
import sys

import pytest
import torch

from conftest import assert_close
from oracle import healnet_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


def _blocks(hn, d, heads, dh, snn, seed):
    torch.manual_seed(seed)
    att = hn.PreNorm(d, hn.Attention(d, heads=heads, dim_head=dh))
    ff = hn.PreNorm(d, hn.FeedForward(d, snn=snn))
    with torch.no_grad():
        for blk in (att, ff):
            blk.norm.weight.add_(0.2 * torch.randn(d))
            blk.norm.bias.add_(0.2 * torch.randn(d))
    return att, ff


def _oracle_block(att, ff, x, heads, snn):
    sa = {k: v for k, v in att.state_dict().items()}
    sf = {k: v for k, v in ff.state_dict().items()}
    return _oracle_block_sd(sa, sf, x, heads, snn)


def _oracle_block_sd(sa, sf, x, heads, snn):
    xn = O.layer_norm(x, sa["norm.weight"], sa["norm.bias"])
    x1 = O.attention(xn, None, sa["fn.to_q.weight"], sa["fn.to_kv.weight"], sa["fn.to_out.0.weight"], sa["fn.to_out.0.bias"], heads) + x
    h = O.feed_forward(O.layer_norm(x1, sf["norm.weight"], sf["norm.bias"]), sf["fn.net.0.weight"], sf["fn.net.0.bias"],
                       sf["fn.net.2.weight"], sf["fn.net.2.bias"], snn=snn)
    return h + x1


@pytest.mark.parametrize("d,heads,dh,b,L,snn", [
    (128, 8, 64, 4, 128, True),      # the default latent block: fused chain
    (128, 8, 64, 3, 48, False),      # GELU gate, l_c = 48 (3 row tiles per sample)
    (128, 4, 32, 2, 128, True),      # inner = 128
    (128, 2, 128, 2, 64, True),      # dim_head 128
    (128, 8, 16, 5, 16, True),       # inner = 128 with 8 heads of 16, one tile per sample
    (128, 3, 64, 2, 128, True),      # inner = 192: not a multiple of 128 -> unfused fallback
    (128, 8, 64, 2, 40, True),       # l_c = 40: rows not a multiple of 16 per sample but b*L is -> still row tiles of 16
    (64, 4, 16, 2, 32, True),        # l_d = 64 -> unfused fallback
])
def test_latent_block_forward_vs_oracle(hn, d, heads, dh, b, L, snn):
    att, ff = _blocks(hn, d, heads, dh, snn, seed=d + heads + dh)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(b, L, d, generator=gen) * 1.2
    with torch.no_grad():
        want = _oracle_block(att, ff, x, heads, snn)
        got = hn.latent_block(att.to(DEV), ff.to(DEV), x.to(DEV))
    assert_close(got.cpu(), want, rel=3e-4, floor=2e-5, what=f"latent block d={d} h={heads} dh={dh} b={b} L={L}")


@pytest.mark.parametrize("heads,dh,snn", [(8, 64, True), (4, 32, False)])
def test_latent_block_backward_vs_oracle_autograd(hn, heads, dh, snn):
    d, b, L = 128, 3, 32
    att, ff = _blocks(hn, d, heads, dh, snn, seed=77)
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(b, L, d, generator=gen)
    dy = torch.randn(b, L, d, generator=gen)
    sa = {k: v.detach().clone().requires_grad_(True) for k, v in att.state_dict().items()}
    sf = {k: v.detach().clone().requires_grad_(True) for k, v in ff.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    want = _oracle_block_sd(sa, sf, xr, heads, snn)
    want.backward(dy)
    att.to(DEV), ff.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    got = hn.latent_block(att, ff, xd)
    assert_close(got.detach().cpu(), want.detach(), rel=3e-4, floor=2e-5, what="latent block (training form)")
    got.backward(dy.to(DEV))
    assert_close(xd.grad.cpu(), xr.grad, rel=2e-3, floor=1e-3, what="dx")
    for blk, sd, nm in ((att, sa, "attn"), (ff, sf, "ff")):
        for k, p in blk.named_parameters():
            assert p.grad is not None, k
            assert_close(p.grad.cpu(), sd[k].grad, rel=2e-3, floor=1e-3, what=f"{nm}.{k}")


_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
import healnet_amd as hn
kw = dict(n_modalities=3, channel_dims=[2000, 3, 96], num_spatial_axes=[1, 2, 1], out_dims=4, depth=2)
torch.manual_seed(5)
model = hn.HealNet(**kw).eval().to("cuda:0")
gen = torch.Generator().manual_seed(6)
ins = [torch.rand(3, 1, 2000, generator=gen).cuda(), torch.rand(3, 40, 36, 3, generator=gen).cuda(), torch.rand(3, 200, 96, generator=gen).cuda()]
with torch.no_grad():
    outs = [model(list(ins)), model([ins[0], None, ins[2]]), model(list(ins), return_embeddings=True)]
    model(list(ins))
    w = model.get_attention_weights()
torch.save([o.cpu() for o in outs] + [t.cpu() for t in w], sys.argv[1])
"""


def test_whole_model_chain_vs_unfused_launches(hn, tmp_path):
    """tab (one-token look-ahead -> broadcast-add head) + image (rank-D binding: Q-only projection) + patch bag (explicit
    binding: scaled Q projection), default width: the chain route and the HN_NO_CHAIN=1 route must agree."""
    script = tmp_path / "run.py"
    script.write_text(_SCRIPT.format(root=ROOT))
    res = {}
    for tag, env in (("chain", {}), ("plain", {"HN_NO_CHAIN": "1"})):
        out = tmp_path / f"{tag}.pt"
        e = dict(os.environ, **env)
        if tag == "chain":
            e.pop("HN_NO_CHAIN", None)
        r = subprocess.run([sys.executable, str(script), str(out)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = torch.load(out)
    assert len(res["chain"]) == len(res["plain"]) == 3 + 2 * 4
    diffs = []
    for i, (a, b_) in enumerate(zip(res["chain"], res["plain"])):
        assert_close(a, b_, rel=2e-5, floor=2e-6, what=f"output {i}: chain vs unfused")
        diffs.append(float((a - b_).abs().max()))
    assert max(diffs) > 0.0, "both runs took the same route (HN_NO_CHAIN had no effect)"


@pytest.mark.parametrize("grad_mode", [False, True], ids=["inference", "taping"])
def test_whole_model_without_attention_trace_vs_oracle(hn, grad_mode):
    """keep_attention_stats=False: no statistics / trace slots are handed to the forward, the latent array then lives in the
    workspace (every chain runs in place, the merge head gets no statistics pointer).  Same three-modality model as above
    (one-token look-ahead head, merge head behind the image block, plain out-projection head behind the patch bag and the
    self-attention), against the oracle and against the default mode."""
    kw = dict(n_modalities=3, channel_dims=[2000, 3, 96], num_spatial_axes=[1, 2, 1], out_dims=4, depth=2)
    torch.manual_seed(15)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(16)
    ins = [torch.rand(3, 1, 2000, generator=gen), torch.rand(3, 40, 36, 3, generator=gen), torch.rand(3, 200, 96, generator=gen)]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins])
    with torch.set_grad_enabled(grad_mode):
        kept = model([t.to(DEV) for t in ins]).detach().cpu()
        model.keep_attention_stats = False
        bare = model([t.to(DEV) for t in ins]).detach().cpu()
        bare2 = model([t.to(DEV) for t in ins]).detach().cpu()
    assert_close(bare, want, rel=1e-3, floor=0.0, abs_floor=1e-5, what="no-trace forward vs oracle")
    assert torch.equal(bare, bare2), "no-trace forward is not deterministic"
    assert_close(bare, kept, rel=1e-6, floor=0.0, abs_floor=1e-6, what="no-trace vs default mode")


@pytest.mark.parametrize("x_heads,cross_dim_head,shape", [(8, 64, (40, 36)), (4, 32, (24, 20)), (2, 64, (16, 48)), (8, 16, (30, 30)),
                                                          (3, 64, (20, 20))])
def test_merge_head_shapes_vs_oracle(hn, x_heads, cross_dim_head, shape):
    """The chain's merge head (split partials of the image block -> normalised average -> folded value projection, chain.hip
    head == 3) at other head counts / head widths than the default 8 x 64: 4 x 32 and 2 x 64 (inner = 128: waves 2..7 idle in
    the merge), 8 x 16; 3 x 64 (inner = 192, not a multiple of 128) stays on merge_vproj_kernel + the unfused out-projection."""
    kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2, x_heads=x_heads,
              cross_dim_head=cross_dim_head)
    torch.manual_seed(31 + x_heads)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(32)
    ins = [torch.rand(5, 1, 2000, generator=gen), torch.rand(5, *shape, 3, generator=gen)]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins])
        got = model([t.to(DEV) for t in ins]).cpu()
    assert_close(got, want, rel=1e-3, floor=0.0, abs_floor=1e-5, what=f"merge head {x_heads}x{cross_dim_head}")
    # the statistics the merge head writes feed the attention-weight export: compare the image block's probabilities
    with torch.no_grad():
        probs = model.get_attention_weights()
    assert all(torch.isfinite(p).all() for p in probs)
    row_sums = probs[1].float().sum(-1)            # layer 0, image block: (b * heads, l_c, N) -> rows sum to 1
    assert_close(row_sums.cpu(), torch.ones_like(row_sums).cpu(), rel=1e-4, floor=0.0, abs_floor=1e-4, what="image block probabilities")


@pytest.mark.parametrize("grad_mode", [False, True], ids=["inference", "taping"])
def test_masked_image_model_on_the_chain_vs_oracle(hn, grad_mode):
    """One image modality with a key mask at the default widths: the masked split-KV core feeds the chain's merge head (splits
    whose tokens are all masked carry weight 0), the self-attention blocks take the LDS core / the chain as usual."""
    kw = dict(n_modalities=1, channel_dims=[3], num_spatial_axes=[2], out_dims=4, depth=2)
    torch.manual_seed(41)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(42)
    img = torch.rand(4, 48, 40, 3, generator=gen)
    mask = torch.rand(4, 48 * 40, generator=gen) > 0.4
    mask[:, 0] = True
    mask[1, 600:] = False                      # a sample whose later splits are masked out entirely
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = O.fusion_forward(sd, O.FusionConfig(**kw), [img.clone()], mask=mask)
    with torch.set_grad_enabled(grad_mode):
        got = model([img.to(DEV)], mask=mask.to(DEV)).detach().cpu()
    assert_close(got, want, rel=1e-3, floor=0.0, abs_floor=1e-5, what="masked image model")


@pytest.mark.parametrize("b", [1, 2, 4, 8, 16])
def test_cluster_chain_small_batches_deterministic_and_vs_oracle(hn, b):
    """Round 3: up to 128 row tiles the chain runs as a CLUSTER (4 or 2 workgroups per 16-row tile, one exchange of FF2 partials
    per chain through the shared L2).  The exchange must be ordered (every partial before its member's flag): 40 repetitions
    of the inference forward and of the taping forward are bitwise identical, and agree with the oracle."""
    kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2)
    torch.manual_seed(71)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(72 + b)
    ins = [torch.rand(b, 1, 2000, generator=gen), torch.rand(b, 56, 40, 3, generator=gen)]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins])
    dins = [t.to(DEV) for t in ins]
    for grad_mode in (False, True):
        with torch.set_grad_enabled(grad_mode):
            first = model(list(dins)).detach().clone()
            for rep in range(40):
                again = model(list(dins)).detach()
                assert torch.equal(first, again), f"b={b} grad_mode={grad_mode}: repetition {rep} differs (unordered exchange?)"
        assert_close(first.cpu(), want, rel=1e-3, floor=0.0, abs_floor=1e-5, what=f"cluster chain b={b} grad_mode={grad_mode}")


def test_cluster_chain_equals_one_workgroup_per_tile():
    """Same model, cluster mode on / off (HN_NO_CHAIN_CLUSTER=1 in a subprocess): logits equal to fp32 summation noise (the FF2
    contraction is summed per hidden chunk in cluster mode), and the cluster really ran (the bits differ)."""
    script = textwrap_dedent("""
        import sys, torch
        sys.path.insert(0, {root!r})
        import healnet_amd as hn
        torch.manual_seed(81)
        m = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to("cuda:0")
        g = torch.Generator().manual_seed(82)
        ins = [torch.rand(4, 1, 2000, generator=g).cuda(), torch.rand(4, 64, 48, 3, generator=g).cuda()]
        with torch.no_grad():
            torch.save(m(ins).cpu(), {dst!r})
    """)
    import tempfile
    outs = {}
    with tempfile.TemporaryDirectory() as d:
        for tag, env in (("cluster", {}), ("single", {"HN_NO_CHAIN_CLUSTER": "1"})):
            dst = os.path.join(d, tag + ".pt")
            r = subprocess.run([sys.executable, "-c", script.format(root=ROOT, dst=dst)], cwd=ROOT, capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-3000:]
            outs[tag] = torch.load(dst)
    assert_close(outs["cluster"], outs["single"], rel=1e-5, floor=0.0, abs_floor=1e-6, what="cluster vs one workgroup per tile")
    assert not torch.equal(outs["cluster"], outs["single"]), "HN_NO_CHAIN_CLUSTER did not change the route?"


def textwrap_dedent(s):
    import textwrap
    return textwrap.dedent(s)


def test_query_fold_in_the_chain_vs_qfold_kernel_and_oracle():
    """Default widths (8 heads of 16 packed slots = the chain's 128-column Q stage): the chain in front of an image block projects
    LN(x) with the staged product W_q . folded W_k (vfold launch) and leaves the folded query + its score bounds itself
    (ChainArgs.qf); HN_NO_QFOLD_CHAIN=1 (subprocess) keeps the 512-column projection + qfold_mfma_kernel.  Both against the
    oracle, against each other to fp32 reassociation noise, at a cluster-mode batch and at a plain one; attention weights (which
    re-run the block from the trace on the un-folded path) must still match the oracle."""
    script = (
        "import sys, torch\n"
        "sys.path.insert(0, {root!r})\n"
        "import healnet_amd as hn\n"
        "from oracle import healnet_cpu as O\n"
        "kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2)\n"
        "torch.manual_seed(7)\n"
        "m = hn.HealNet(**kw).eval().to('cuda:0')\n"
        "with torch.no_grad():\n"
        "    for p in m.parameters():\n"
        "        if p.dim() == 1: p.add_(0.1 * torch.randn_like(p))\n"
        "outs = []\n"
        "for b in (3, 20):\n"
        "    g = torch.Generator().manual_seed(8 + b)\n"
        "    ins = [torch.rand(b, 1, 2000, generator=g), torch.rand(b, 44, 52, 3, generator=g)]\n"
        "    sd = {{k: v.detach().cpu() for k, v in m.state_dict().items()}}\n"
        "    with torch.no_grad():\n"
        "        got = m([t.to('cuda:0') for t in ins]).cpu()\n"
        "        probs = m.get_attention_weights()[1].float().sum(-1).cpu()\n"
        "        want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins[:1]] + [ins[1].clone()]) if b == 3 else None\n"
        "    outs.append((got, want, probs))\n"
        "torch.save(outs, {dst!r})\n"
    )
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, env in (("fold", {}), ("plain", {"HN_NO_QFOLD_CHAIN": "1"})):
            dst = os.path.join(td, tag + ".pt")
            e = dict(os.environ, **env)
            if not env:
                e.pop("HN_NO_QFOLD_CHAIN", None)
            r = subprocess.run([sys.executable, "-c", script.format(root=ROOT, dst=dst)], cwd=ROOT, env=e, capture_output=True, text=True,
                               timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            res[tag] = torch.load(dst)
    for (gf, wf, pf), (gp, wp, pp) in zip(res["fold"], res["plain"]):
        assert torch.isfinite(gf).all()
        assert_close(gf, gp, rel=2e-5, floor=2e-6, what="query fold in the chain vs qfold kernel")
        if wf is not None:
            assert_close(gf, wf, rel=1e-3, floor=0.0, abs_floor=1e-5, what="query fold in the chain vs oracle")
            assert_close(gp, wp, rel=1e-3, floor=0.0, abs_floor=1e-5, what="qfold kernel vs oracle")
        assert_close(pf, torch.ones_like(pf), rel=1e-4, floor=0.0, abs_floor=1e-4, what="image block probabilities")
    assert any(not torch.equal(a[0], b_[0]) for a, b_ in zip(res["fold"], res["plain"])), "HN_NO_QFOLD_CHAIN did not change the route?"
