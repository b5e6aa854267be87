"""GPU: staged models (include/healnet_hip.h "Staged models") -- the reference's tuned TCGA shapes
(config/best_hyperparams.yml: l_d = 119 / 126 / 62 / 65, ONE cross head of 16 / 63 / 27 / 103, 25 / 17 / 17 / 16 latents, no latent
self-attention, both dropouts on) and other shapes outside the latent chains' own (narrow widths, a latent self-attention with
padded heads, tied layers) run as zero-padded images on the fast path.

  * the route is actually taken (hn_fusion_is_staged), and only for shapes that need it;
  * logits, embeddings, Attention.attn_weights and every parameter gradient vs the CPU oracle -- with dropout through the masks the
    build itself exports (the oracle restates nn.Dropout for a given mask);
  * the same numbers as the generic per-block route (HN_NO_STAGING=1, a second process) to fp32 rounding;
  * from an all-NaN workspace (HN_POISON_WS=1): finite, deterministic -- pad rows / columns never leak;
  * the flat-gradient route with two forwards in front of one backward.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

from conftest import assert_close


def _grad_close(got, ref, scale, what, rel=2e-3):
    """Gradients that are exactly zero in exact arithmetic (the query side of a one-token softmax: to_q, its LayerNorm) are
    rounding noise on both sides: bounded, not compared."""
    if float(ref.abs().max()) < 1e-6 * scale:
        assert float(got.abs().max()) < 1e-4 * scale, what
        return
    assert_close(got, ref, rel=rel, floor=1e-3, abs_floor=1e-5 * scale, what=what)
from oracle import healnet_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TUNED = {   # config/best_hyperparams.yml of the reference
    "blca": dict(depth=2, l_c=25, l_d=119, cross_dim_head=16, latent_dim_head=127, attn_dropout=0.0830, ff_dropout=0.4733),
    "brca": dict(depth=2, l_c=17, l_d=126, cross_dim_head=63, latent_dim_head=20, attn_dropout=0.4553, ff_dropout=0.3647),
    "kirp": dict(depth=5, l_c=17, l_d=62, cross_dim_head=27, latent_dim_head=113, attn_dropout=0.3179, ff_dropout=0.0474),
    "ucec": dict(depth=2, l_c=16, l_d=65, cross_dim_head=103, latent_dim_head=51, attn_dropout=0.2488, ff_dropout=0.0571),
}
COMMON = dict(n_modalities=2, channel_dims=[200, 48], num_spatial_axes=[1, 1], out_dims=4, x_heads=1, l_heads=8, self_per_cross_attn=0,
              num_freq_bands=2, max_freq=2.0)
OTHER = {   # not the reference's tuned files: what else the staged route has to carry
    "self_attn_padded_heads": dict(n_modalities=2, channel_dims=[30, 3], num_spatial_axes=[1, 2], out_dims=3, depth=2, l_c=12, l_d=48,
                                   x_heads=2, cross_dim_head=24, l_heads=3, latent_dim_head=20),
    "tied_narrow": dict(n_modalities=2, channel_dims=[64, 20], num_spatial_axes=[1, 1], out_dims=2, depth=3, l_c=16, l_d=64, x_heads=1,
                        cross_dim_head=64, l_heads=2, latent_dim_head=32, weight_tie_layers=True, snn=False),
    "ragged_rows_only": dict(n_modalities=1, channel_dims=[40], num_spatial_axes=[1], out_dims=3, depth=2, l_c=25, l_d=128, x_heads=2,
                             cross_dim_head=64, l_heads=2, latent_dim_head=64),
}


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


def _inputs(kw, b, n_tokens, seed):
    gen = torch.Generator().manual_seed(seed)
    ins = []
    for ch, ax in zip(kw["channel_dims"], kw["num_spatial_axes"]):
        shape = (b, 1, ch) if ch >= 200 else ((b, n_tokens, ch) if ax == 1 else (b, 6, 7, ch))
        ins.append(torch.rand(*shape, generator=gen))
    return ins


def test_route_selection(hn):
    for name, kw in TUNED.items():
        assert hn.HealNet(**COMMON, **kw).runs_staged(), name
    for name, kw in OTHER.items():
        assert hn.HealNet(**kw).runs_staged(), name
    # the chains' own shapes are run as they are
    assert not hn.HealNet(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4).runs_staged()
    # ... and shapes that do not fit them after padding take the generic route (l_d > 128; 8 heads of 127 -> 8 x 128 > 512)
    assert not hn.HealNet(n_modalities=1, channel_dims=[8], num_spatial_axes=[1], out_dims=2, l_d=160, l_c=16).runs_staged()
    assert not hn.HealNet(n_modalities=1, channel_dims=[8], num_spatial_axes=[1], out_dims=2, l_d=64, l_c=16, l_heads=8,
                          latent_dim_head=127).runs_staged()


@pytest.mark.parametrize("name", list(TUNED))
def test_tuned_shapes_with_dropout_vs_oracle_under_the_exported_masks(hn, name):
    from test_gpu_dropout import _oracle_masks
    kw = dict(COMMON, **TUNED[name])
    b = 3
    torch.manual_seed(900 + len(name))
    model = hn.HealNet(**kw).train()
    ins = _inputs(kw, b, 70, 901)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model.to(DEV)
    got = model([t.to(DEV) for t in ins])
    dl = torch.randn(got.shape, generator=torch.Generator().manual_seed(902))
    (got * dl.to(DEV)).sum().backward()
    assert model.runs_staged()
    seed, offset = model._last_rng
    n_tokens = [t.numel() // (b * t.shape[-1]) for t in ins]
    drop = _oracle_masks(hn, model, kw, b, n_tokens, seed, offset, [True] * kw["n_modalities"])
    want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins], drop=drop)
    (want * dl).sum().backward()
    assert_close(got.detach().cpu(), want.detach(), rel=1e-3, floor=0.0, abs_floor=1e-5, what=name + " logits (train, dropout)")
    scale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, p in model.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        _grad_close(p.grad.cpu(), ref, scale, f"{name} grad[{k}]")


@pytest.mark.parametrize("name", list(TUNED) + list(OTHER))
def test_staged_shapes_vs_oracle_without_dropout(hn, name):
    kw = dict(COMMON, **TUNED[name]) if name in TUNED else dict(OTHER[name])
    kw.pop("attn_dropout", None); kw.pop("ff_dropout", None)
    b = 4
    torch.manual_seed(910 + len(name))
    model = hn.HealNet(**kw).train()
    assert model.runs_staged()
    ins = _inputs(kw, b, 90, 911)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins])
    dl = torch.randn(want.shape, generator=torch.Generator().manual_seed(912))
    (want * dl).sum().backward()
    emb_want = O.fusion_forward({k: v.detach() for k, v in sd.items()}, O.FusionConfig(**kw), [t.clone() for t in ins], return_embeddings=True)
    model.to(DEV)
    dins = [t.to(DEV) for t in ins]
    with torch.no_grad():
        inf = model(list(dins)).cpu()
        emb = model(list(dins), return_embeddings=True).cpu()
    assert_close(inf, want.detach(), rel=1e-3, floor=0.0, abs_floor=1e-5, what=name + " inference logits")
    assert emb.shape == emb_want.shape
    assert_close(emb, emb_want, rel=1e-3, floor=0.0, abs_floor=1e-5, what=name + " embeddings")
    got = model(list(dins))
    assert_close(got.detach().cpu(), want.detach(), rel=1e-3, floor=0.0, abs_floor=1e-5, what=name + " taping logits")
    (got * dl.to(DEV)).sum().backward()
    # tied layers: the oracle's state_dict leaves are per alias, the module's parameter is one tensor
    ref_by_ptr = {}
    for k, v in model.state_dict(keep_vars=True).items():
        g = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        ref_by_ptr[v.data_ptr()] = ref_by_ptr.get(v.data_ptr(), 0) + g
    scale = max(float(g.abs().max()) for g in ref_by_ptr.values())
    for k, p in model.named_parameters():
        _grad_close(p.grad.cpu(), ref_by_ptr[p.data_ptr()], scale, f"{name} grad[{k}]")
    # Attention.attn_weights after the taping forward (un-padded trace on the tape) and after an inference forward
    trace = O.FusionTrace()
    O.fusion_forward({k: v.detach() for k, v in sd.items()}, O.FusionConfig(**kw), [t.clone() for t in ins], trace=trace)
    if kw.get("weight_tie_layers"):
        return                                  # (a tied block reports its last slot only)
    for mode in ("taping", "inference"):
        if mode == "inference":
            with torch.no_grad():
                model(list(dins))
        weights = [w for w in model.get_attention_weights() if w is not None]
        last = {}                                # a layer's latent self-attention runs once per modality: the module reports its last run
        for tag, pr in zip(trace.attn_tags, trace.attn):
            last[(tag[0], tag[1], tag[2] if tag[1] == "cross" else 0)] = pr
        wanted = [last[k] for k in sorted(last, key=lambda t: (t[0], 0 if t[1] == "cross" else 1, t[2]))]
        assert len(weights) == len(wanted)
        for w_got, w_want in zip(weights, wanted):
            assert_close(w_got.cpu(), w_want.reshape(w_got.shape), rel=2e-3, floor=1e-4, what=f"{name} attn_weights ({mode})")


_TWIN = textwrap.dedent("""
    import json, sys, torch
    sys.path.insert(0, {root!r})
    import healnet_amd as hn
    kw = {kw!r}
    torch.manual_seed(77)
    model = hn.HealNet(**kw).train().to("cuda:0")
    gen = torch.Generator().manual_seed(78)
    ins = [torch.rand(*s, generator=gen).to("cuda:0") for s in {shapes!r}]
    outs = {{"staged": bool(model.runs_staged())}}
    runs = []
    for rep in range(2):
        model._rng_offset = 0
        for p in model.parameters():
            p.grad = None
        out = model(list(ins))
        out.square().sum().backward()
        runs.append([out.detach().cpu()] + [p.grad.cpu() for p in model.parameters()])
    outs["deterministic"] = all(torch.equal(a, c) for a, c in zip(*runs))
    outs["finite"] = all(bool(torch.isfinite(t).all()) for t in runs[0])
    torch.save(runs[0], {path!r})
    print("RESULT " + json.dumps(outs))
""")


def _twin(kw, shapes, path, **env):
    out = subprocess.run([sys.executable, "-c", _TWIN.format(root=ROOT, kw=kw, shapes=shapes, path=path)], cwd=ROOT, capture_output=True,
                         text=True, timeout=600, env=dict(os.environ, **env))
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.parametrize("name", ["blca", "kirp", "self_attn_padded_heads"])
def test_staged_equals_the_generic_route_and_survives_a_poisoned_workspace(hn, name, tmp_path):
    """Same seed, same dropout generator state: the staged route, the generic per-block route (HN_NO_STAGING=1) and the staged
    route on an all-NaN workspace (HN_POISON_WS=1) produce the same logits and gradients (fp32 rounding apart)."""
    kw = dict(COMMON, **TUNED[name]) if name in TUNED else dict(OTHER[name], attn_dropout=0.2, ff_dropout=0.1)
    shapes = [tuple(t.shape) for t in _inputs(kw, 5, 130, 0)]
    a = _twin(kw, shapes, str(tmp_path / "a.pt"))
    g = _twin(kw, shapes, str(tmp_path / "g.pt"), HN_NO_STAGING="1")
    p = _twin(kw, shapes, str(tmp_path / "p.pt"), HN_POISON_WS="1")
    assert a["staged"] and not g["staged"] and p["staged"]
    for r in (a, g, p):
        assert r["finite"] and r["deterministic"], r
    A, G, P = (torch.load(str(tmp_path / f)) for f in ("a.pt", "g.pt", "p.pt"))
    for x, y in zip(A, P):
        assert torch.equal(x, y), "the poisoned-workspace run differs: something reads a pad row / column"
    scale = max(float(t.abs().max()) for t in G[1:])
    assert_close(A[0], G[0], rel=2e-4, floor=0.0, abs_floor=1e-6, what=name + " logits staged vs generic")
    for i, (x, y) in enumerate(zip(A[1:], G[1:])):
        _grad_close(x, y, scale, f"{name} gradient {i} staged vs generic", rel=1e-3)


def test_two_forwards_one_backward_on_the_flat_route_staged(hn):
    kw = dict(COMMON, **TUNED["brca"])
    kw.pop("attn_dropout"); kw.pop("ff_dropout")
    torch.manual_seed(5)
    model = hn.HealNet(**kw).train().to(DEV)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    a, c = _inputs(kw, 3, 40, 6), _inputs(kw, 3, 40, 7)
    cfg = O.FusionConfig(**kw)
    want = O.fusion_forward(sd, cfg, [t.clone() for t in a]).square().sum() + O.fusion_forward(sd, cfg, [t.clone() for t in c]).sum()
    want.backward()
    flat = hn.train.flatten_parameters(model)
    flat.zero_grad()
    loss = model([t.to(DEV) for t in a]).square().sum() + model([t.to(DEV) for t in c]).sum()
    loss.backward()
    assert_close(loss.detach().cpu(), want.detach(), rel=1e-4, floor=0.0, abs_floor=1e-5, what="summed loss")
    scale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, p in model.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        assert float((p.grad.cpu() - ref).abs().max()) <= 5e-4 * scale, k


def test_a_host_that_keeps_padded_weights_passes_the_staged_descriptor_itself(hn):
    """include/healnet_hip.h "Staged models": instead of letting every forward stage the weights, a host may keep the zero-padded
    images itself and hand over the padded descriptor with the *_valid fields set.  Built here by hand from the documented layout
    (l_d -> 128 columns; head h of a projection at rows h * dhp; w_q rows / w_out pitch / a self block's w_kv rows up to a multiple
    of 128; W1's value rows at 0 and gate rows at 512) and run through hn_fusion_forward directly: same logits as the model's own
    (auto-staged) forward, and hn_fusion_is_staged says 0 for it -- it is not staged a second time."""
    import ctypes as C
    import json
    from healnet_amd import _capi, ops
    from healnet_amd._rt import stream_ptr
    kw = dict(n_modalities=2, channel_dims=[40, 30], num_spatial_axes=[1, 1], out_dims=3, depth=2, l_c=16, l_d=62, x_heads=2,
              cross_dim_head=27, l_heads=2, latent_dim_head=20)
    torch.manual_seed(321)
    model = hn.HealNet(**kw).eval().to(DEV)
    assert model.runs_staged()
    ins = [t.to(DEV) for t in _inputs(kw, 3, 50, 322)]
    with torch.no_grad():
        want = model(list(ins))
    ld, L = kw["l_d"], kw["l_c"]
    up128 = lambda v: (v + 127) // 128 * 128                                  # noqa: E731
    pad_dh = lambda d: 16 if d <= 16 else 32 if d <= 32 else 64 if d <= 64 else 128      # noqa: E731
    spec = json.loads(model._spec_text)
    params = list(model.parameters())
    padded = list(params)

    def vec(i, n):
        if i >= 0:
            v = torch.zeros(n, device=DEV)
            v[:params[i].numel()] = params[i].detach()
            padded[i] = v

    def attn(a, cross):
        H, dh = a["heads"], a["dim_head"]
        dhp, inner_s = pad_dh(dh), H * pad_dh(dh)
        ip = up128(inner_s)
        nw, nb, cg, cb, wq, wkv, wo, bo = a["p"]
        vec(nw, 128); vec(nb, 128); vec(bo, 128)
        q = torch.zeros(ip, 128, device=DEV)
        o = torch.zeros(128, ip, device=DEV)
        for h in range(H):
            q[h * dhp:h * dhp + dh, :ld] = params[wq].detach()[h * dh:(h + 1) * dh]
            o[:ld, h * dhp:h * dhp + dh] = params[wo].detach()[:, h * dh:(h + 1) * dh]
        padded[wq], padded[wo] = q, o
        src = params[wkv].detach()
        if cross:
            kv = torch.zeros(2 * inner_s, src.shape[1], device=DEV)
            for j in range(2 * H):
                kv[j * dhp:j * dhp + dh] = src[j * dh:(j + 1) * dh]
        else:
            kv = torch.zeros(up128(2 * inner_s), 128, device=DEV)
            for j in range(2 * H):
                kv[j * dhp:j * dhp + dh, :ld] = src[j * dh:(j + 1) * dh]
        padded[wkv] = kv
        a["dim_head"], a["query_dim"] = dhp, 128
        return dh

    def ff(f):
        nw, nb, w1, b1, w2, b2 = f["p"]
        vec(nw, 128); vec(nb, 128); vec(b2, 128)
        hid = 4 * ld
        W1 = torch.zeros(1024, 128, device=DEV)
        W1[:hid, :ld] = params[w1].detach()[:hid]
        W1[512:512 + hid, :ld] = params[w1].detach()[hid:]
        B1 = torch.zeros(1024, device=DEV)
        B1[:hid] = params[b1].detach()[:hid]
        B1[512:512 + hid] = params[b1].detach()[hid:]
        W2 = torch.zeros(128, 512, device=DEV)
        W2[:ld, :hid] = params[w2].detach()
        padded[w1], padded[b1], padded[w2] = W1, B1, W2
        f["dim"] = 128

    real_dh_cross = [attn(a, True) for a in spec["cross_attn"]]
    for f in spec["cross_ff"]:
        ff(f)
    real_dh_self = [attn(a, False) for a in spec["self_attn"]]
    for f in spec["self_ff"]:
        ff(f)
    lat = torch.zeros(L, 128, device=DEV)
    lat[:, :ld] = params[spec["latents"]].detach()
    padded[spec["latents"]] = lat
    hnw, hnb, hw, hb = spec["head_p"]
    vec(hnw, 128); vec(hnb, 128)
    HW = torch.zeros(kw["out_dims"], 128, device=DEV)
    HW[:, :ld] = params[hw].detach()
    padded[hw] = HW
    spec["l_d"] = 128
    sp = ops.Spec(json.dumps(spec, sort_keys=True))
    desc, keep = sp.model([t.contiguous() for t in padded])
    ca, cf, sa, sf = keep[0], keep[1], keep[2], keep[3]
    desc.l_d_valid = ld
    for k, dh in enumerate(real_dh_cross):
        ca[k].dim_head_valid, ca[k].query_dim_valid, cf[k].dim_valid = dh, ld, ld
    for k, dh in enumerate(real_dh_self):
        sa[k].dim_head_valid, sa[k].query_dim_valid, sf[k].dim_valid = dh, ld, ld
    lib = _capi.lib()
    assert lib.hn_fusion_is_staged(C.byref(desc)) == 0
    inp, held, b = sp.inputs(ins)
    need = lib.hn_fusion_workspace_bytes(C.byref(desc), inp, b)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    out = torch.empty(b, kw["out_dims"], device=DEV)
    _capi.check(lib.hn_fusion_forward(C.byref(desc), inp, b, None, 0, 0, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(),
                                      stream_ptr(torch.device(DEV)), None), "hn_fusion_forward")
    assert_close(out.cpu(), want.cpu(), rel=1e-5, floor=0.0, abs_floor=1e-6, what="pre-staged descriptor vs the model's own forward")


def test_graph_replay_of_a_staged_model(hn):
    """HealNet.capture on a staged shape: the weight-staging launch is part of the captured graph, so a replay equals the eager
    forward bit for bit on new input values, sees an in-place weight update, and serves attention weights from its static buffers."""
    kw = dict(COMMON, **TUNED["brca"])
    kw.pop("attn_dropout"); kw.pop("ff_dropout")
    torch.manual_seed(41)
    model = hn.HealNet(**kw).eval().to(DEV)
    assert model.runs_staged()
    graph = model.capture([t.to(DEV) for t in _inputs(kw, 2, 300, 42)])
    for trial in range(2):
        ins = [t.to(DEV) for t in _inputs(kw, 2, 300, 43 + trial)]
        with torch.no_grad():
            got = graph(list(ins)).clone()
            eager = model(list(ins))
        assert torch.equal(got, eager), f"trial {trial}: graph replay differs from the eager forward"
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)                                  # in place: the replay stages the updated weights
        got = graph(list(ins)).clone()
        eager = model(list(ins))
    assert torch.equal(got, eager) and not torch.equal(got, torch.zeros_like(got))
    with torch.no_grad():
        graph(list(ins))
        w_graph = [w.clone() for w in model.get_attention_weights() if w is not None]
        model(list(ins))
        w_eager = [w for w in model.get_attention_weights() if w is not None]
    assert len(w_graph) == len(w_eager) > 0
    for a, c in zip(w_graph, w_eager):
        assert torch.equal(a, c)
