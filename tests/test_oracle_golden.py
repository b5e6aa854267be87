"""CPU: pin oracle/healnet_cpu.py against the fixtures generated from the reference
(tools/gen_goldens.py).  This is what makes the oracle a trustworthy checker."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden, rel_err
from oracle import healnet_cpu as O


def test_g1_fourier_exact():
    g = load_golden("g1_fourier")
    for key, want in g.items():
        S, mf, nb = key.split("_")
        S, mf, nb = int(S[1:]), float(mf[2:]), int(nb[2:])
        got = O.fourier_features(torch.linspace(-1.0, 1.0, S)[:, None], mf, nb)
        assert torch.equal(got, want), key
    # known-answer rows quoted in SURVEY.md §8c (S=3, max_freq 10, 2 bands)
    rows = g["S3_mf10_nb2"][:, 0, :]
    assert torch.allclose(rows[1], torch.tensor([0.0, 0.0, 1.0, 1.0, 0.0]))
    assert torch.allclose(rows[0], torch.tensor([8.7423e-08, 6.7553e-07, -1.0, -1.0, -1.0]), atol=1e-9)


def test_g2_preprocess_exact():
    g = load_golden("g2_preprocess")
    for name in ("tab", "img", "vol", "one"):
        got = O.encode_modality(g[name + "_in"], 2, 10.0, True)
        assert torch.equal(got, g[name + "_enc"]), name
    assert g["img_enc"].shape == (2, 30, 13) and g["vol_enc"].shape == (2, 60, 17) and g["tab_enc"].shape == (2, 1, 25)


def test_g3_attention(manifest):
    g = load_golden("g3_attention")
    for name, c in manifest["g3_attention"]["cases"].items():
        y, p = O.attention(g[name + "_x"], g.get(name + "_ctx"), g[name + "_wq"], g[name + "_wkv"], g[name + "_wo"],
                           g[name + "_bo"], c["heads"], g.get(name + "_mask"), return_weights=True)
        assert rel_err(y, g[name + "_y"]) < 5e-6, name
        assert rel_err(p, g[name + "_p"]) < 5e-6, name
        assert p.shape == g[name + "_p"].shape


def test_g4_feedforward():
    g = load_golden("g4_feedforward")
    for tag, snn in (("selu", True), ("gelu", False)):
        y = O.feed_forward(g[tag + "_x"], g[tag + "_w1"], g[tag + "_b1"], g[tag + "_w2"], g[tag + "_b2"], snn)
        assert rel_err(y, g[tag + "_y"]) < 5e-6


G5 = ["m1_d1", "m2_d3", "m3_d3", "m2_d3_tied", "m2_d2_noself", "m2_d2_nofourier", "m2_d2_gelu", "m2_d2_nohead",
      "m2_d2_bands4", "m2_d2_masked"]


@pytest.mark.parametrize("name", G5)
def test_g5_tiny_models(name, manifest):
    g = load_golden("g5_" + name)
    kw = manifest["g5_" + name]["kwargs"]
    cfg = O.FusionConfig(**kw)
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd::")}
    ins = [g[f"in{i}"] for i in range(kw["n_modalities"])]
    mask = g.get("mask")
    tr = O.FusionTrace()
    y = O.fusion_forward(sd, cfg, ins, mask=mask, trace=tr)
    assert rel_err(y, g["logits"]) < 1e-5
    if "emb" in g:
        assert rel_err(O.fusion_forward(sd, cfg, ins, mask=mask, return_embeddings=True), g["emb"]) < 1e-5
    if "attn0" in g and not kw.get("weight_tie_layers", False):
        mine = O.attention_weights_in_module_order(tr, cfg)
        i = 0
        while f"attn{i}" in g:
            assert rel_err(mine[i], g[f"attn{i}"]) < 1e-5, f"attn{i}"
            i += 1
        assert i == len(mine)
    if "logits_missing1" in g:
        miss = [ins[0], None] + ins[2:]
        assert rel_err(O.fusion_forward(sd, cfg, miss), g["logits_missing1"]) < 1e-5
        assert rel_err(O.fusion_forward(sd, cfg, miss, verbose=True), g["logits_missing1_verbose"]) < 1e-5
        assert rel_err(O.fusion_forward(sd, cfg, ins[:1]), g["logits_missing1"]) < 1e-5 or kw["n_modalities"] > 2
    if "logits_missing0" in g:
        assert rel_err(O.fusion_forward(sd, cfg, [None, ins[1]]), g["logits_missing0"]) < 1e-5


def test_g5_gradients_via_autograd(manifest):
    """The oracle is differentiable torch code: its autograd gradients must match the reference's."""
    name = "m2_d3"
    g = load_golden("g5_" + name)
    kw = manifest["g5_" + name]["kwargs"]
    cfg = O.FusionConfig(**kw)
    sd = {k[4:]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("sd::")}
    ins = [g[f"in{i}"] for i in range(kw["n_modalities"])]
    out = O.fusion_forward(sd, cfg, ins)
    (out * O.filler_input(out.shape, 77)).sum().backward()
    for k, p in sd.items():
        want = g["grad::" + k]
        if want.abs().max() == 0:
            continue
        assert rel_err(p.grad, want) < 2e-4, k


@pytest.mark.parametrize("name", ["cfg3s", "cfg4", "tuned", "cfg1"])
def test_g6_default_size(name, manifest):
    """Default hyper-parameters at the BASELINE shapes, closed-form weights (outputs-only fixtures)."""
    m = manifest["g6_" + name]
    cfg = O.FusionConfig(**m["kwargs"])
    sd = O.filler_state_dict(cfg, gain=m["gain"])
    ins = [O.filler_input(s, 20 + i) for i, s in enumerate(m["shapes"])]
    g = load_golden("g6_" + name)
    with torch.no_grad():
        y = O.fusion_forward(sd, cfg, ins)
    assert rel_err(y, g["logits"]) < 3e-5


def test_kat0_seed_route(manifest):
    """KAT-0: logits of the seed-0 default model on seed-continued torch.rand inputs."""
    g = load_golden("kat0")
    want_row0 = torch.tensor([1.24044466, 0.22287285, 0.77036822, -0.45615000])
    want_row3 = torch.tensor([0.58372909, 1.38270676, 1.87094307, 0.29817703])
    assert torch.allclose(g["logits"][0], want_row0, atol=2e-6)
    assert torch.allclose(g["logits"][3], want_row3, atol=2e-6)
    assert abs(float(g["emb_mean"]) - 0.8022665) < 1e-5 and abs(float(g["emb_absmax"]) - 6.958313) < 1e-4


# ---------------------------------------------------------------------------------------------------------------
# training-step tail: oracle/train_cpu.py against fixtures generated from the reference's
# survival_loss.py / train_utils.py / torch.optim.Adam + OneCycleLR (tools/gen_goldens_train.py)
# ---------------------------------------------------------------------------------------------------------------
def test_train_oracle_surv_nll_matches_reference(manifest):
    from oracle import train_cpu as T
    g = load_golden("g7_surv_nll")
    for case in manifest["g7_surv_nll"]["cases"]:
        t = case["tag"]
        w = g[t + "weights"] if case["weighted"] else None
        loss, dl, hz, sv = T.surv_nll(g[t + "logits"], g[t + "y"], g[t + "c"], w, alpha=manifest["g7_surv_nll"]["alpha"],
                                      eps=manifest["g7_surv_nll"]["eps"])
        assert abs(float(loss) - float(g[t + "loss"])) <= 1e-6 * max(1.0, abs(float(g[t + "loss"]))), t
        assert torch.equal(dl, g[t + "dlogits"]), t
        assert torch.equal(hz, g[t + "hazards"]) and torch.equal(sv, g[t + "survival"]), t


def test_train_oracle_l1_adam_matches_reference(manifest):
    from oracle import train_cpu as T
    g = load_golden("g7_l1_adam")
    m = manifest["g7_l1_adam"]
    ps = [g[f"p{i}_init"].clone() for i in range(len(m["shapes"]))]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for t in range(m["n_steps"]):
        reg = 0.0
        for i, p in enumerate(ps):
            q = p.clone().requires_grad_(True)
            c = g[f"coef{i}"]
            loss = (((q * c).sum() ** 2 + (q * q * c).sum()) * (0.1 * (t + 1))) / m["gc"]
            (gr,) = torch.autograd.grad(loss, q)
            reg += T.l1_adam_step(p, gr, ms[i], vs[i], step=t + 1, l1=m["l1"], lr=float(g["lrs"][t]), beta1=float(g["beta1s"][t]))
        assert abs(reg - float(g["reg_losses"][t])) <= 1e-5 * float(g["reg_losses"][t])
        for i, p in enumerate(ps):
            assert_close(p, g[f"p{i}_step{t}"], rel=2e-6, floor=1e-7, what=f"adam p{i} step{t}")


def test_temperature_softmax_restatement_matches_reference():
    """a7: the oracle's attention() restates temperature_softmax as softmax(sim / temperature) (oracle/healnet_cpu.py);
    pinned here against outputs of the reference's own function."""
    g = load_golden("g8_temperature_softmax")
    for i in range(4):
        assert torch.equal(torch.softmax(g[f"x{i}"] / float(g[f"t{i}"]), dim=-1), g[f"y{i}"])
    assert torch.equal(torch.softmax(g["xd"] / 0.5, dim=1), g["yd"])


def test_verbose_quirk_with_short_tensor_lists(manifest):
    """g9 (tools/gen_goldens_quirks.py, outputs of the reference itself): `verbose=True` skips the latent self block only for
    None entries INSIDE the tensor list (healnet.py:193, :229-232); a modality beyond a shorter list still runs it (:238)."""
    g = load_golden("g9_verbose_shortlist")
    m = manifest["g9_verbose_shortlist"]
    cfg = O.FusionConfig(**m["kwargs"])
    sd = {k[4:]: v for k, v in g.items() if k.startswith("sd::")}
    ins = [g[f"in{i}"] for i in range(3)]
    for name, idx in m["cases"].items():
        lst = [None if i == "None" else ins[i] for i in idx]
        for mode in ("quiet", "verbose"):
            got = O.fusion_forward(sd, cfg, lst, verbose=(mode == "verbose"))
            assert rel_err(got, g[f"logits::{name}::{mode}"]) < 1e-5, (name, mode)
    assert torch.equal(g["logits::short1::verbose"], g["logits::short1::quiet"])
    assert not torch.equal(g["logits::none_tail::verbose"], g["logits::none_tail::quiet"])
