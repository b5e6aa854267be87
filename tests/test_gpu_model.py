"""GPU: whole-model parity of the fused forward (hn_fusion_forward) against reference-generated
fixtures, the CPU oracle, and size-independent properties at the BASELINE configs' full sizes."""
import pytest
import torch

from conftest import assert_close, load_golden, rel_err
from oracle import healnet_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3          # BASELINE.json north_star: 1e-3 rel fp32 on identical inputs


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


@pytest.fixture(autouse=True, params=["inference", "taping"])
def forward_path(request):
    """Every test of this module runs twice: under torch.no_grad() -> hn_fusion_forward (the inference forward bench.py
    measures: ones column, packed context, score-bound softmax, chained trace buffers) and with grad mode on ->
    hn_fusion_forward_train (the tape-recording forward autograd uses, parameters require grad)."""
    with torch.set_grad_enabled(request.param == "taping"):
        yield request.param


G5 = ["m1_d1", "m2_d3", "m3_d3", "m2_d3_tied", "m2_d2_noself", "m2_d2_nofourier", "m2_d2_gelu", "m2_d2_nohead",
      "m2_d2_bands4", "m2_d2_masked"]


@pytest.mark.parametrize("name", G5)
def test_tiny_models_match_reference_fixtures(hn, name, manifest):
    g = load_golden("g5_" + name)
    kw = manifest["g5_" + name]["kwargs"]
    model = hn.HealNet(**kw).eval()
    model.load_state_dict({k[4:]: v for k, v in g.items() if k.startswith("sd::")}, strict=True)
    model.to(DEV)
    ins = [g[f"in{i}"].to(DEV) for i in range(kw["n_modalities"])]
    mask = g["mask"].to(DEV) if "mask" in g else None
    y = model(list(ins), mask=mask)
    assert_close(y.cpu(), g["logits"], rel=2e-4, what=name + ".logits")
    if "emb" in g:
        assert_close(model(list(ins), mask=mask, return_embeddings=True).cpu(), g["emb"], rel=2e-4, what=name + ".emb")
    if "attn0" in g:
        model(list(ins), mask=mask)
        got = model.get_attention_weights()
        i = 0
        while f"attn{i}" in g:
            assert got[i].shape == g[f"attn{i}"].shape
            assert_close(got[i].cpu(), g[f"attn{i}"], rel=5e-4, what=f"{name}.attn{i}")
            i += 1
        assert i == len(got)
    if "logits_missing1" in g:
        miss = [ins[0], None] + ins[2:]
        assert_close(model(list(miss)).cpu(), g["logits_missing1"], rel=2e-4, what=name + ".missing1")
        assert_close(model(list(miss), verbose=True).cpu(), g["logits_missing1_verbose"], rel=2e-4, what=name + ".missing1v")
        if kw["n_modalities"] == 2:
            assert_close(model([ins[0]]).cpu(), g["logits_missing1"], rel=2e-4, what=name + ".short-list")
    if "logits_missing0" in g:
        assert_close(model([None, ins[1]]).cpu(), g["logits_missing0"], rel=2e-4, what=name + ".missing0")


@pytest.mark.parametrize("name", ["cfg1", "cfg3s", "cfg4", "tuned"])
def test_default_size_configs_match_reference_fixtures(hn, name, manifest):
    """BASELINE shapes, default hyper-parameters, closed-form weights; expected logits / embeddings /
    latent-mean attention rows were produced by the reference itself."""
    m = manifest["g6_" + name]
    cfg = O.FusionConfig(**m["kwargs"])
    model = hn.HealNet(**m["kwargs"]).eval()
    model.load_state_dict(O.filler_state_dict(cfg, gain=m["gain"]), strict=True)
    model.to(DEV)
    ins = [O.filler_input(s, 20 + i).to(DEV) for i, s in enumerate(m["shapes"])]
    g = load_golden("g6_" + name)
    y = model(list(ins))
    assert_close(y.cpu(), g["logits"], rel=TOL, floor=0.0, abs_floor=1e-5, what=name + ".logits")
    assert_close(model(list(ins), return_embeddings=True).cpu(), g["emb"], rel=TOL, what=name + ".emb")
    big = int(g["attn_mean_index"])
    model(list(ins))
    p = model.layers[0][2 * big].fn.attn_weights
    assert_close(p.mean(dim=1)[:, :4096].cpu(), g["attn_mean"], rel=TOL, floor=1e-3, what=name + ".attn_mean")


def test_kat0_seed_route(hn, manifest):
    """torch.manual_seed(0) model + seed-continued torch.rand inputs -> the reference's logits (KAT-0)."""
    g = load_golden("kat0")
    torch.manual_seed(0)
    model = hn.HealNet(**manifest["kat0"]["kwargs"]).eval()
    tab = torch.rand(4, 1, 2000)
    img = torch.rand(4, 224, 224, 3)
    model.to(DEV)
    y = model([tab.to(DEV), img.to(DEV)]).cpu()
    assert_close(y, g["logits"], rel=TOL, what="kat0.logits")
    want_row0 = torch.tensor([1.24044466, 0.22287285, 0.77036822, -0.45615000])
    assert (y[0] - want_row0).abs().max() < 1e-3
    emb = model([tab.to(DEV), img.to(DEV)], return_embeddings=True)
    assert abs(float(emb.mean()) - float(g["emb_mean"])) < 1e-4
    assert_close(emb[0, :4, :8].cpu(), g["emb_row0"], rel=TOL, what="kat0.emb")


def _cfg2_model(hn, seed=0):
    torch.manual_seed(seed)
    return hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).eval().to(DEV)


def test_cfg2_full_size_properties(hn):
    """b=32 at the headline config: (i) bitwise determinism, (ii) every sample's logits equal the
    logits of that sample run in a smaller batch (no cross-sample coupling; different split-KV
    geometry), (iii) sample permutation equivariance, (iv) oracle parity on a 2-sample slice."""
    model = _cfg2_model(hn)
    gen = torch.Generator().manual_seed(1234)
    tab = torch.rand(32, 1, 2000, generator=gen)
    img = torch.rand(32, 224, 224, 3, generator=gen)
    tab_d, img_d = tab.to(DEV), img.to(DEV)
    y1 = model([tab_d, img_d])
    y2 = model([tab_d, img_d])
    assert torch.equal(y1, y2)
    y_small = torch.cat([model([tab_d[i:i + 4], img_d[i:i + 4]]) for i in range(0, 32, 4)])
    assert_close(y_small.cpu(), y1.cpu(), rel=1e-4, what="batch-slicing")
    perm = torch.randperm(32, generator=gen)
    y_perm = model([tab_d[perm.to(DEV)], img_d[perm.to(DEV)]])
    assert_close(y_perm.cpu(), y1.cpu()[perm], rel=1e-5, what="permutation")
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.FusionConfig(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
    with torch.no_grad():
        want = O.fusion_forward(sd, cfg, [tab[:2], img[:2]])
    assert_close(y1[:2].cpu(), want, rel=TOL, what="cfg2 vs oracle")


def test_list_mutation_compat_flag(hn):
    model = hn.HealNet(n_modalities=2, channel_dims=[20, 3], num_spatial_axes=[1, 2], out_dims=3, l_c=8, l_d=16, x_heads=2,
                       l_heads=2, cross_dim_head=4, latent_dim_head=4).eval().to(DEV)
    tab, img = torch.rand(2, 1, 20).to(DEV), torch.rand(2, 6, 5, 3).to(DEV)
    lst = [tab, img]
    model(lst)
    assert lst[0] is tab and lst[1] is img                 # default: caller's list untouched
    model.compat_mutate_inputs = True
    model(lst)
    assert lst[0].shape == (2, 1, 25) and lst[1].shape == (2, 30, 13)      # reference behaviour (:222)
    with pytest.raises(AssertionError):
        model(lst)                                                          # second call fails the axis check, as in the reference


def test_shape_errors_raise_instead_of_being_swallowed(hn):
    model = hn.HealNet(n_modalities=2, channel_dims=[20, 3], num_spatial_axes=[1, 2], out_dims=3, l_c=8, l_d=16, x_heads=2,
                       l_heads=2, cross_dim_head=4, latent_dim_head=4).eval().to(DEV)
    with pytest.raises(ValueError):
        model([torch.rand(2, 7, 33).to(DEV)])              # patch bag in the tabular slot (main.py:536-538 pattern)
    with pytest.raises(AssertionError):
        model([torch.rand(2, 20).to(DEV), None])


# ------------------------------------------------------------------------------------------------
# the other BASELINE.json configurations at their full shapes
# ------------------------------------------------------------------------------------------------
def _oracle_logits(model, kw, ins):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        return O.fusion_forward(sd, O.FusionConfig(**kw), [None if t is None else t.cpu() for t in ins])


def test_cfg4_full_size_vs_oracle(hn):
    """TCGA-BRCA-shaped: omic (8,1,2000) + WSI patch bag (8,4096,768), explicit K/V path at N=4096, D=773."""
    kw = dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4)
    torch.manual_seed(3)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(77)
    ins = [torch.rand(8, 1, 2000, generator=gen).to(DEV), torch.rand(8, 4096, 768, generator=gen).to(DEV)]
    y = model(list(ins))
    assert_close(y.cpu(), _oracle_logits(model, kw, ins), rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg4 logits")
    y2 = torch.cat([model([t[i:i + 3] for t in ins]) for i in range(0, 8, 3)])
    assert_close(y2.cpu(), y.cpu(), rel=1e-4, what="cfg4 batch-slicing")


def test_cfg3_volume_vs_oracle_and_properties(hn):
    """3-modality incl. the 12x224x224x3 volume (N = 602 112 tokens, D = 18 -> rank-D path with dp = 32):
    oracle parity on one sample (the CPU side needs ~18 GB for the materialised scores), then b=4 consistency."""
    kw = dict(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4)
    torch.manual_seed(4)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(78)
    ins = [torch.rand(4, 1, 2000, generator=gen).to(DEV), torch.rand(4, 224, 224, 3, generator=gen).to(DEV),
           torch.rand(4, 12, 224, 224, 3, generator=gen).to(DEV)]
    y = model(list(ins))
    assert torch.isfinite(y).all()
    want = _oracle_logits(model, kw, [t[:1] for t in ins])
    assert_close(y[:1].cpu(), want, rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg3 logits (sample 0)")
    y1 = torch.cat([model([t[i:i + 1] for t in ins]) for i in range(4)])
    assert_close(y1.cpu(), y.cpu(), rel=1e-4, what="cfg3 batch-slicing")
    # missing volume == 2-modality schedule with the self blocks of the skipped iterations still running
    y_missing = model([ins[0], ins[1], None])
    want_missing = _oracle_logits(model, kw, [ins[0][:1], ins[1][:1], None])
    assert_close(y_missing[:1].cpu(), want_missing, rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg3 missing volume")


def test_cfg5_shape_depth8_properties(hn):
    """4 modalities (tab + 2 WSI bags + volume), depth 8: determinism, slicing invariance, permutation equivariance
    at b = 4 per GPU (the config's per-GPU share of its global batch 32 on 8 GPUs)."""
    kw = dict(n_modalities=4, channel_dims=[2000, 768, 768, 3], num_spatial_axes=[1, 1, 1, 3], out_dims=4, depth=8)
    torch.manual_seed(5)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(79)
    ins = [torch.rand(4, 1, 2000, generator=gen).to(DEV), torch.rand(4, 4096, 768, generator=gen).to(DEV),
           torch.rand(4, 4096, 768, generator=gen).to(DEV), torch.rand(4, 12, 224, 224, 3, generator=gen).to(DEV)]
    y = model(list(ins))
    assert torch.isfinite(y).all() and torch.equal(y, model(list(ins)))
    y2 = torch.cat([model([t[i:i + 2] for t in ins]) for i in range(0, 4, 2)])
    assert_close(y2.cpu(), y.cpu(), rel=1e-4, what="cfg5 batch-slicing")
    perm = torch.tensor([2, 0, 3, 1], device=DEV)
    assert_close(model([t[perm] for t in ins]).cpu(), y[perm].cpu(), rel=1e-5, what="cfg5 permutation")


@pytest.mark.parametrize("name", ["cfg1", "cfg4"])
def test_attention_importance_matches_reference_row_means(hn, name, manifest):
    """the (b*h, N) latent-row mean of attn_weights -- what the reference's explainer reduces every matrix
    to -- straight from the softmax statistics; expected rows were produced by the reference (fixture attn_mean)."""
    m = manifest["g6_" + name]
    cfg = O.FusionConfig(**m["kwargs"])
    model = hn.HealNet(**m["kwargs"]).eval()
    model.load_state_dict(O.filler_state_dict(cfg, gain=m["gain"]), strict=True)
    model.to(DEV)
    ins = [O.filler_input(s, 20 + i).to(DEV) for i, s in enumerate(m["shapes"])]
    g = load_golden("g6_" + name)
    big = int(g["attn_mean_index"])
    with torch.no_grad():
        model(list(ins))
    imp = model.get_attention_importance()
    full = model.get_attention_weights()
    assert len(imp) == len(full)
    for a, w in zip(imp, full):
        assert a.shape == (w.shape[0], w.shape[2])
        assert_close(a, w.mean(dim=1), rel=1e-5, floor=1e-6, what="importance vs full matrix")
    got = model.layers[0][2 * big].fn.attn_importance
    assert_close(got[:, :4096].cpu(), g["attn_mean"], rel=TOL, floor=1e-3, what=name + ".attn_mean")


def test_attention_importance_masked_and_standalone(hn):
    torch.manual_seed(5)
    att = hn.Attention(32, 7, heads=2, dim_head=16).to(DEV)
    x = torch.randn(3, 10, 32, device=DEV)
    ctx = torch.randn(3, 77, 7, device=DEV)
    mask = torch.rand(3, 77, device=DEV) > 0.3
    with torch.no_grad():
        att(x, context=ctx, mask=mask)
    w, a = att.attn_weights, att.attn_importance
    assert a.shape == (6, 77)
    assert_close(a, w.mean(dim=1), rel=1e-5, floor=1e-6, what="standalone importance")
    assert float(a[~mask.repeat_interleave(2, dim=0)].abs().max()) == 0.0
    assert_close(a.sum(-1), torch.ones(6, device=DEV), rel=1e-5, what="rows of a softmax average to a distribution")


@pytest.mark.parametrize("chan,axes_shape,dim_head", [
    (6, (1,), 8),        # one-token context with D = 11 <= 15: rank-D pitch 16, but the one-token shortcut (natural channel order)
    (6, (3,), 8),        # same D with 3 tokens: packed rank-D core
    (20, (1,), 4),       # D = 29 -> pitch 32 from the 4-float rounding with dim_head 4: explicit binding, natural order
    (20, (5,), 4),
    (3, (6, 5), 4),      # D = 13, dim_head 4 (padded to 16): rank-D with dp == padded head dim
    (3, (6, 5), 64),     # the default image block at a tiny size
    (9, (4, 3, 2), 16),  # volume, D = 9 + 15 = 24 -> dp = 32 > padded head 16: explicit
    (9, (4, 3, 2), 32),  # ... and the packed dp = 32 rank-D core (ks = 6)
    (1, (7,), 64),       # D = 6: ks = 2
    (2, (40,), 64),      # D = 7: ks = 2
])
def test_context_layout_corner_cases_vs_oracle(hn, chan, axes_shape, dim_head):
    """Which context layout a modality gets (ones column, packed channels, natural) depends on D, the head dim and the
    token count; every combination must give the oracle's logits on BOTH forwards (see forward_path)."""
    kw = dict(n_modalities=2, channel_dims=[5, chan], num_spatial_axes=[1, len(axes_shape)], out_dims=3, depth=2, l_c=8, l_d=16,
              x_heads=2, l_heads=2, cross_dim_head=dim_head, latent_dim_head=8)
    torch.manual_seed(chan * 100 + dim_head)
    model = hn.HealNet(**kw).eval().to(DEV)
    gen = torch.Generator().manual_seed(9)
    ins = [torch.rand(3, 2, 5, generator=gen).to(DEV), torch.rand(3, *axes_shape, chan, generator=gen).to(DEV)]
    y = model(list(ins))
    assert_close(y.cpu(), _oracle_logits(model, kw, ins), rel=2e-4, what=f"layout chan={chan} axes={axes_shape} dh={dim_head}")


def test_one_token_route_above_32_samples(hn):
    """b > 32 takes the row-chunked form of the weight-streaming GEMV (the tabular K/V projection has K = 2005, which
    no tile kernel accepts): every sample must equal its logits from a batch of 8."""
    torch.manual_seed(6)
    model = hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4, depth=1).eval().to(DEV)
    gen = torch.Generator().manual_seed(80)
    for b in (33, 72):
        tab, img = torch.rand(b, 1, 2000, generator=gen).to(DEV), torch.rand(b, 8, 8, 3, generator=gen).to(DEV)
        y = model([tab, img])
        y8 = torch.cat([model([tab[i:i + 8], img[i:i + 8]]) for i in range(0, b, 8)])
        assert_close(y.detach().cpu(), y8.detach().cpu(), rel=1e-5, what=f"b={b} vs batches of 8")


def test_narrow_input_dtypes_on_both_forwards(hn):
    """bf16 tensors are read in place and uint8 images as byte / 255 by the encode kernel: on the inference AND the
    tape-recording forward the result must equal the fp32 path on the same (already rounded) values, bit for bit."""
    torch.manual_seed(8)
    model = hn.HealNet(n_modalities=2, channel_dims=[30, 3], num_spatial_axes=[1, 2], out_dims=3, depth=2, l_c=16, l_d=32,
                       x_heads=2, l_heads=2, cross_dim_head=16, latent_dim_head=8).eval().to(DEV)
    gen = torch.Generator().manual_seed(81)
    tab = torch.rand(3, 4, 30, generator=gen).to(DEV)
    img8 = torch.randint(0, 256, (3, 9, 7, 3), generator=gen, dtype=torch.uint8).to(DEV)
    img_f = img8.cpu().float().div(255).to(DEV)      # on the CPU: torch's GPU division by a scalar multiplies by 1/255 (1 ulp off)
    want8 = model([tab, img_f])
    assert torch.equal(model([tab, img8]).detach(), want8.detach())
    tab16, img16 = tab.to(torch.bfloat16), img_f.to(torch.bfloat16)
    want16 = model([tab16.float(), img16.float()])
    assert torch.equal(model([tab16, img16]).detach(), want16.detach())
