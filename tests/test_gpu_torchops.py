"""GPU: torch.ops.healnet_hip.* as THE route (north_star: "exposed as torch.ops through a thin C-ABI extension").

  * operator registrations pass torch.library.opcheck (schema, fake kernel, autograd registration, AOT dispatch);
  * the stand-alone Attention / FeedForward modules are differentiable through the registered backward (vs oracle autograd);
  * HealNet.forward is ONE operator call: torch.compile(model, fullgraph=True) traces it without a graph break, inference
    and training (gradients equal to the eager ones);
  * attention weights after a train-mode forward with dropout, the verbose quirk with a short tensor list.
"""
import pytest
import torch

from conftest import assert_close
from oracle import healnet_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ops = torch.ops.healnet_hip


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


def _attn_args(gen, b=2, L=24, N=70, D=9, heads=2, dh=16, qd=32, cross=True, grad=True):
    inner = heads * dh
    kdim = D if cross else qd
    r = lambda *s: torch.randn(*s, generator=gen)      # noqa: E731
    P = dict(norm_w=1 + 0.2 * r(qd), norm_b=0.2 * r(qd), w_q=r(inner, qd) * qd ** -0.5, w_kv=r(2 * inner, kdim) * kdim ** -0.5,
             w_out=r(qd, inner) * inner ** -0.5, b_out=0.1 * r(qd))
    P = {k: v.to(DEV).requires_grad_(grad) for k, v in P.items()}
    x = r(b, L, qd).to(DEV).requires_grad_(grad)
    ctx = torch.rand(b, N, D, generator=gen).to(DEV) if cross else None
    return x, ctx, P


def test_opcheck_blocks(hn):
    gen = torch.Generator().manual_seed(3)
    utils = ("test_schema", "test_autograd_registration", "test_faketensor", "test_aot_dispatch_static")
    for cross in (True, False):
        x, ctx, P = _attn_args(gen, cross=cross)
        args = (x, ctx, None, P["norm_w"], P["norm_b"], None, None, P["w_q"], P["w_kv"], P["w_out"], P["b_out"], 2, True, True)
        torch.library.opcheck(ops.attention_fwd.default, args, test_utils=utils)
    d = 32
    r = lambda *s: torch.randn(*s, generator=gen).to(DEV).requires_grad_(True)      # noqa: E731
    ff_args = (r(3, 16, d), r(d), r(d), r(8 * d, d), r(8 * d), r(d, 4 * d), r(d), False, True)
    torch.library.opcheck(ops.feed_forward.default, ff_args, test_utils=utils)
    torch.library.opcheck(ops.head.default, (r(3, 16, d), r(d), r(d), r(4, d), r(4)), test_utils=utils)
    torch.library.opcheck(ops.fourier_encode_concat.default, (torch.rand(2, 5, 6, 3, device=DEV), 2, 10.0, True),
                          test_utils=("test_schema", "test_faketensor"))


def test_opcheck_fusion(hn):
    torch.manual_seed(2)
    model = hn.HealNet(n_modalities=2, channel_dims=[7, 3], num_spatial_axes=[1, 2], out_dims=3, depth=2, l_c=16, l_d=32, x_heads=2,
                       l_heads=2, cross_dim_head=16, latent_dim_head=8).to(DEV)
    params = list(model.parameters())
    tab, img = torch.rand(3, 1, 7, device=DEV), torch.rand(3, 6, 5, 3, device=DEV)
    torch.library.opcheck(ops.fusion_forward.default, ([tab, img], None, [p.detach() for p in params], model._spec_text, 0, False, True),
                          test_utils=("test_schema", "test_faketensor"))
    # (no test_aot_dispatch_static here: it compares ALL outputs of two runs bit for bit, and the tape output has alignment gaps
    # nothing writes; AOT tracing of the op is covered by test_torch_compile_traces_the_model_without_graph_breaks, which
    # compares logits and gradients)
    torch.library.opcheck(ops.fusion_forward_train.default, ([tab, None], None, params, model._spec_text, 0, False, None, None, []),
                          test_utils=("test_schema", "test_autograd_registration", "test_faketensor"))


@pytest.mark.parametrize("cross", [True, False], ids=["cross", "self"])
def test_standalone_attention_module_autograd_vs_oracle(hn, cross):
    """PreNorm(Attention) called directly (the way the reference's Attention.forward :400 is used) with x and parameters that
    require grad: outputs and every gradient against torch autograd of the oracle."""
    torch.manual_seed(7)
    qd, D, heads, dh = 32, 13, 2, 16
    blk = hn.PreNorm(qd, hn.Attention(qd, D if cross else None, heads=heads, dim_head=dh), context_dim=D if cross else None).to(DEV)
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(3, 20, qd, generator=gen)
    ctx = torch.rand(3, 90, D, generator=gen) if cross else None
    dy = torch.randn(3, 20, qd, generator=gen)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    xn = O.layer_norm(xr, sd["norm.weight"], sd["norm.bias"])
    cn = O.layer_norm(ctx, sd["norm_context.weight"], sd["norm_context.bias"]) if cross else None
    want = O.attention(xn, cn, sd["fn.to_q.weight"], sd["fn.to_kv.weight"], sd["fn.to_out.0.weight"], sd["fn.to_out.0.bias"], heads)
    want.backward(dy)
    xd = x.to(DEV).requires_grad_(True)
    got = blk(xd, context=ctx.to(DEV)) if cross else blk(xd)
    assert_close(got.detach().cpu(), want.detach(), rel=3e-4, what="standalone attention fwd")
    got.backward(dy.to(DEV))
    assert_close(xd.grad.cpu(), xr.grad, rel=2e-3, floor=1e-3, what="dx")
    for k, p in blk.named_parameters():
        assert p.grad is not None, k
        assert_close(p.grad.cpu(), sd[k].grad, rel=2e-3, floor=1e-3, what="grad " + k)
    assert blk.fn.attn_weights.shape == (3 * heads, 20, 90 if cross else 20)


def test_standalone_feedforward_module_autograd_vs_oracle(hn):
    torch.manual_seed(9)
    d = 32
    blk = hn.PreNorm(d, hn.FeedForward(d, snn=True)).to(DEV)
    gen = torch.Generator().manual_seed(10)
    x, dy = torch.randn(4, 10, d, generator=gen), torch.randn(4, 10, d, generator=gen)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    want = O.feed_forward(O.layer_norm(xr, sd["norm.weight"], sd["norm.bias"]), sd["fn.net.0.weight"], sd["fn.net.0.bias"],
                          sd["fn.net.2.weight"], sd["fn.net.2.bias"], snn=True)
    want.backward(dy)
    xd = x.to(DEV).requires_grad_(True)
    got = blk(xd)
    assert_close(got.detach().cpu(), want.detach(), rel=3e-4, what="standalone ff fwd")
    got.backward(dy.to(DEV))
    assert_close(xd.grad.cpu(), xr.grad, rel=2e-3, floor=1e-3, what="dx")
    for k, p in blk.named_parameters():
        assert_close(p.grad.cpu(), sd[k].grad, rel=2e-3, floor=1e-3, what="grad " + k)


KW = dict(n_modalities=2, channel_dims=[12, 3], num_spatial_axes=[1, 2], out_dims=3, depth=2, l_c=16, l_d=32, x_heads=2, l_heads=2,
          cross_dim_head=16, latent_dim_head=8)


@pytest.mark.parametrize("backend", ["aot_eager", "inductor"])
def test_torch_compile_traces_the_model_without_graph_breaks(hn, backend):
    """fullgraph=True raises on any graph break: the whole forward must be one torch.ops.healnet_hip.fusion_* node."""
    import torch._dynamo
    torch._dynamo.reset()
    torch.manual_seed(12)
    model = hn.HealNet(**KW).eval().to(DEV)
    gen = torch.Generator().manual_seed(13)
    tab, img = torch.rand(3, 1, 12, generator=gen).to(DEV), torch.rand(3, 7, 6, 3, generator=gen).to(DEV)
    compiled = torch.compile(model, fullgraph=True, backend=backend)
    with torch.no_grad():
        want = model([tab, img])
        got = compiled([tab, img])
        assert torch.equal(got, want)
        assert torch.equal(compiled([tab, None]), model([tab, None]))
    # training through the compiled module: the registered backward runs hn_fusion_backward
    model.train()
    for p in model.parameters():
        p.grad = None
    dl = torch.randn(3, 3, generator=gen).to(DEV)
    (model([tab, img]) * dl).sum().backward()
    eager = {k: p.grad.clone() for k, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    (compiled([tab, img]) * dl).sum().backward()
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        assert torch.equal(p.grad, eager[k]), k
    explain = torch._dynamo.explain(model)([tab, img])
    assert explain.graph_break_count == 0, explain.break_reasons
    names = [n.target for g in explain.graphs for n in g.graph.nodes if n.op == "call_function"]
    assert any("fusion_forward" in str(t) for t in names), names


def test_attention_weights_after_a_dropout_training_forward(hn):
    """the tape of a train-mode forward with dropout is laid out with the dropout descriptor (a one-token
    block keeps the general-path tensors, D == 16 contexts get a wider pitch); the attention-weight export must read the
    statistics at THOSE offsets.  attn_weights are the undropped probabilities (healnet.py:420), so they must equal the
    weights of an eval-mode forward of the same model and inputs."""
    torch.manual_seed(21)
    kw = dict(n_modalities=3, channel_dims=[40, 3, 11], num_spatial_axes=[1, 2, 1], out_dims=3, depth=2, l_c=16, l_d=32, x_heads=2,
              l_heads=2, cross_dim_head=16, latent_dim_head=8, attn_dropout=0.3, ff_dropout=0.0)
    model = hn.HealNet(**kw).to(DEV)
    gen = torch.Generator().manual_seed(22)
    ins = [torch.rand(3, 1, 40, generator=gen).to(DEV), torch.rand(3, 6, 5, 3, generator=gen).to(DEV),
           torch.rand(3, 9, 11, generator=gen).to(DEV)]            # one-token tabular, image (D = 13), sequence with D = 16
    model.eval()
    with torch.no_grad():
        model(list(ins))
    ref_first = [w.clone() for w in model.get_attention_weights()[:3]]     # layer 0 blocks see the same latent array in both modes
    model.train()
    model(list(ins))
    got = model.get_attention_weights()
    assert len(got) == 8 and all(w is not None for w in got)
    for i, (a, b_) in enumerate(zip(got[:1], ref_first[:1])):            # the very first block: identical input, no dropout upstream
        assert_close(a, b_, rel=1e-5, floor=1e-6, what=f"attn_weights[{i}] after a dropout training forward")
    for w in got:                                                          # every exported matrix is a proper softmax
        assert_close(w.sum(-1), torch.ones_like(w.sum(-1)), rel=1e-4, what="rows sum to one")
    imp = model.get_attention_importance()
    for a, w in zip(imp, got):
        assert_close(a, w.mean(dim=1), rel=1e-5, floor=1e-6, what="importance vs full matrix (dropout tape)")


@pytest.mark.parametrize("grad_mode", [False, True], ids=["inference", "taping"])
def test_verbose_quirk_applies_only_inside_the_tensor_list(hn, manifest, capsys, grad_mode):
    """healnet.py:193 builds missing_idx from the None entries INSIDE the list; a modality beyond a shorter
    list fails in the bare try/except (:238) and still runs the latent self block, verbose or not.  Expected logits are the
    reference's own (fixture g9, tools/gen_goldens_quirks.py)."""
    from conftest import load_golden
    g = load_golden("g9_verbose_shortlist")
    m = manifest["g9_verbose_shortlist"]
    model = hn.HealNet(**m["kwargs"]).eval()
    model.load_state_dict({k[4:]: v for k, v in g.items() if k.startswith("sd::")}, strict=True)
    model.to(DEV)
    ins = [g[f"in{i}"].to(DEV) for i in range(3)]
    with torch.set_grad_enabled(grad_mode):
        for name, idx in m["cases"].items():
            lst = [None if i == "None" else ins[i] for i in idx]
            for mode in ("quiet", "verbose"):
                y = model(list(lst), verbose=(mode == "verbose"))
                assert_close(y.detach().cpu(), g[f"logits::{name}::{mode}"], rel=2e-4, what=f"{name}/{mode}")
    capsys.readouterr()


def test_out_of_range_survival_label_poisons_the_loss(hn):
    """y outside [0, n_bins) makes the reference's torch.gather raise; the kernel answers NaN (loss and that
    sample's gradient row) and reads nothing out of bounds."""
    logits = torch.randn(4, 4, device=DEV, requires_grad=True)
    c = torch.tensor([0, 1, 0, 1], device=DEV)
    good = hn.train.surv_nll_loss(logits, torch.tensor([0, 3, 2, 1], device=DEV), c)
    assert torch.isfinite(good.loss)
    bad = hn.train.surv_nll_loss(logits, torch.tensor([0, 4, 2, 1], device=DEV), c, weights=torch.ones(4, device=DEV))
    assert torch.isnan(bad.loss)
    bad.loss.backward()
    assert torch.isnan(logits.grad[1]).all() and torch.isfinite(logits.grad[[0, 2, 3]]).all()
