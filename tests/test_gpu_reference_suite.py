"""GPU ports of the reference's OWN tests (healnet/tests/test_healnet.py:26-67) -- same constructor calls, same input shapes, same
shape assertions, written against the reference's import lines (`from healnet.models import *`) -- with what the reference does
not pin added on top: parity against the CPU oracle, batch-slicing invariance, a poisoned (all-NaN) workspace, forward + backward.

  test_attention        Attention(query_dim=32, context_dim=2189) on latent (10, 256, 32), context (10, 1, 2189)      :26-32
  test_healnet (m1)     HealNet(1, [2189], [1], 5) on (10, 1, 2189)                                                   :36-49
  test_healnet (m2)     HealNet(2, [2189, 100], [1, 2], 4) on (10, 1, 2189) + (10, 224, 224, 100)                      :50-58
                        a regime no BASELINE config reaches: 100 image channels -> D = 110 >= dim_head, i.e. the EXPLICIT K/V binding
                        at N = 50 176 tokens (2 GB of K|V per block at b = 10, the K/V projection at M = 501 760 rows)
  constructor assert    tests/test_host_logic.py (CPU)                                                                 :63-67
"""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest
import torch

from conftest import assert_close
from oracle import healnet_cpu as O
from healnet.models import *  # noqa: F401,F403  (the reference test's import line)
from healnet.models import Attention, HealNet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

B, T_C, T_D, I_C, I_H, I_W, L_C, L_D = 10, 1, 2189, 100, 224, 224, 256, 32      # test_healnet.py:9-20


def _oracle_threads():
    return min(32, torch.get_num_threads())          # (a 256-core host runs the oracle's small-GEMM mix slower on every core)


def test_attention():
    torch.manual_seed(2601)
    query = torch.randn(B, T_C, T_D)
    latent = torch.randn(B, L_C, L_D)
    attention = Attention(query_dim=L_D, context_dim=T_D)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in attention.state_dict().items()}
    attention.to(DEV)
    x = latent.to(DEV).requires_grad_(True)
    updated_latent = attention(x=x, context=query.to(DEV))
    assert updated_latent.shape == (B, L_C, L_D)                                  # the reference's assertion (:32)
    xr = latent.clone().requires_grad_(True)
    want, probs = O.attention(xr, query, sd["to_q.weight"], sd["to_kv.weight"], sd["to_out.0.weight"], sd["to_out.0.bias"], heads=8,
                              return_weights=True)
    assert_close(updated_latent.detach().cpu(), want.detach(), rel=1e-3, floor=0.0, abs_floor=1e-5, what="Attention(32, 2189)")
    # attn_weights: (b * heads, l_c, 1), all ones for a single key
    w = attention.attn_weights
    assert w.shape == (B * 8, L_C, 1) and torch.equal(w.cpu(), probs.detach())
    gen = torch.Generator().manual_seed(2602)
    dy = torch.randn(want.shape, generator=gen)
    (want * dy).sum().backward()
    (updated_latent * dy.to(DEV)).sum().backward()
    for k, p in attention.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(ref)
        scale = float(ref.abs().max())
        if scale == 0.0:                     # to_q: dead for a one-token context (softmax over one key is constant)
            assert float(got.abs().max()) == 0.0, k
        else:
            assert float((got - ref).abs().max()) <= 2e-4 * scale, (k, float((got - ref).abs().max()) / scale)
    gx = x.grad.cpu() if x.grad is not None else torch.zeros_like(latent)
    rx = xr.grad if xr.grad is not None else torch.zeros_like(latent)
    assert float((gx - rx).abs().max()) <= 1e-6 + 2e-4 * float(rx.abs().max())


@pytest.mark.parametrize("grad", [False, True], ids=["inference", "taping"])
def test_healnet_unimodal(grad):
    torch.manual_seed(2611)
    tabular_data = torch.randn(B, T_C, T_D)
    m1 = HealNet(n_modalities=1, channel_dims=[T_D], num_spatial_axes=[1], out_dims=5)
    sd = {k: v.detach().clone() for k, v in m1.state_dict().items()}
    m1.to(DEV)
    with torch.set_grad_enabled(grad):
        logits1 = m1([tabular_data.to(DEV)])
    assert logits1.shape == (B, 5)                                                # :49
    with torch.no_grad():
        want = O.fusion_forward(sd, O.FusionConfig(n_modalities=1, channel_dims=[T_D], num_spatial_axes=[1], out_dims=5), [tabular_data])
    assert_close(logits1.detach().cpu(), want, rel=1e-3, floor=0.0, abs_floor=1e-5, what="HealNet(1, [2189], [1], 5)")


KW2 = dict(n_modalities=2, channel_dims=[T_D, I_C], num_spatial_axes=[1, 2], out_dims=4)

_BIMODAL_SCRIPT = textwrap.dedent("""
    import json, sys, time, torch
    sys.path.insert(0, {root!r})
    from healnet.models import HealNet
    from healnet_amd import _rt
    assert _rt._POISON == {poison!r}
    torch.manual_seed(2621)
    m2 = HealNet(**{kw!r}).eval().to("cuda:0")
    gen = torch.Generator().manual_seed(2622)
    tab = torch.randn({b}, 1, {t_d}, generator=gen).to("cuda:0")
    img = torch.randn({b}, {h}, {w}, {c}, generator=gen).to("cuda:0")
    out = {{}}
    with torch.no_grad():
        full = m2([tab, img])
        again = m2([tab, img])
        part = m2([tab[3:5], img[3:5]])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m2([tab, img])
        torch.cuda.synchronize()
        out["ms_per_forward"] = (time.perf_counter() - t0) / 5 * 1e3
    out["shape"] = list(full.shape)
    out["finite"] = bool(torch.isfinite(full).all())
    out["deterministic"] = bool(torch.equal(full, again))
    out["slice_rel"] = float((part - full[3:5]).abs().max() / full.abs().max())
    out["logits"] = full.cpu().tolist()
    print("RESULT " + json.dumps(out))
""")


def _run_bimodal(poison):
    env = dict(os.environ, HN_POISON_WS="1" if poison else "0")
    src = _BIMODAL_SCRIPT.format(root=ROOT, poison=poison, kw=KW2, b=B, t_d=T_D, h=I_H, w=I_W, c=I_C)
    out = subprocess.run([sys.executable, "-c", src], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_healnet_bimodal_reference_shapes_vs_oracle():
    """m2 of the reference's test at its own size, b = 10, from an all-NaN workspace: shape as asserted there, bitwise determinism,
    batch-slicing invariance (another split geometry), equality with the un-poisoned run, and the oracle on a 2-sample slice (its
    scores alone are 0.8 GB per sample and block)."""
    res = _run_bimodal(poison=True)
    assert res["shape"] == [B, 4]                                                 # :58
    assert res["finite"] and res["deterministic"], res
    assert res["slice_rel"] < 2e-5, res["slice_rel"]
    clean = _run_bimodal(poison=False)
    assert clean["logits"] == res["logits"], "a poisoned workspace changed the result: something reads scratch it did not write"
    print(f"HealNet(2, [2189, 100], [1, 2], 4) at b = 10: {clean['ms_per_forward']:.2f} ms per forward")
    got = torch.tensor(res["logits"])
    torch.manual_seed(2621)
    sd = {k: v.detach().clone() for k, v in HealNet(**KW2).state_dict().items()}
    gen = torch.Generator().manual_seed(2622)
    tab = torch.randn(B, 1, T_D, generator=gen)
    img = torch.randn(B, I_H, I_W, I_C, generator=gen)
    threads = torch.get_num_threads()
    torch.set_num_threads(_oracle_threads())
    try:
        with torch.no_grad():
            want = torch.cat([O.fusion_forward(sd, O.FusionConfig(**KW2), [tab[i:i + 1].clone(), img[i:i + 1].clone()]) for i in (0, 7)])
    finally:
        torch.set_num_threads(threads)
    assert_close(got[[0, 7]], want, rel=1e-3, floor=0.0, abs_floor=1e-5, what="HealNet(2, [2189, 100], [1, 2], 4)")


def test_healnet_bimodal_forward_backward():
    """The same model through the tape-recording forward and the fused backward: at b = 10 (shape, finite gradients for every
    parameter, time), and ONE sample against the oracle's autograd (logits and all gradients; criteria of test_gpu_fullsize)."""
    from test_gpu_fullsize import _grad_parity
    import healnet_amd
    torch.manual_seed(2631)
    m2 = HealNet(**KW2).train().to(DEV)
    gen = torch.Generator().manual_seed(2632)
    tab = torch.randn(B, 1, T_D, generator=gen).to(DEV)
    img = torch.randn(B, I_H, I_W, I_C, generator=gen).to(DEV)
    for it in range(2):
        m2.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        logits2 = m2([tab, img])
        assert logits2.shape == (B, 4)
        logits2.square().sum().backward()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    print(f"HealNet(2, [2189, 100], [1, 2], 4) at b = 10: forward + backward {ms:.1f} ms")
    for k, p in m2.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    del m2, tab, img
    torch.cuda.empty_cache()
    gen = torch.Generator().manual_seed(2633)
    ins = [torch.randn(1, 1, T_D, generator=gen), torch.randn(1, I_H, I_W, I_C, generator=gen)]
    threads = torch.get_num_threads()
    torch.set_num_threads(_oracle_threads())
    try:
        _grad_parity(healnet_amd, KW2, ins, seed=2634, what="ref_m2_b1")
    finally:
        torch.set_num_threads(threads)
