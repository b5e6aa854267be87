"""GPU: regression tests for the advisor's round-2 findings, each against the CPU oracle.

  * explicit (D > 31) multi-head CROSS block behind a chain with N >> l_c and self_per_cross_attn = 0: its K/V projection must
    live in the block's own scratch, not in the chain's latent-K/V buffer (sized for l_c rows) -- run under HN_POISON_WS=1
    (every call starts from an all-NaN workspace) at a split and an unsplit geometry, both forwards;
  * padded head width behind a chain (dim_head 48 x 8 heads, inner = 384): the pad columns of Q are zero-filled in the buffer
    the Q projection actually writes;
  * two forwards in front of one backward on the FlatParameters route (the flat gradient buffer is not an autograd-saved tensor);
  * a parameter that is a view at an odd float offset (unaligned for the chain's 16-byte loads) takes the per-block launches
    instead of failing the forward.
"""
import os
import subprocess
import sys
import textwrap

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


_POISON_SCRIPT = textwrap.dedent("""
    import json, sys, torch
    sys.path.insert(0, {root!r})
    import healnet_amd as hn
    from healnet_amd import _rt
    from oracle import healnet_cpu as O
    assert _rt._POISON
    kw = {kw!r}
    shapes = {shapes!r}
    torch.manual_seed(101)
    model = hn.HealNet(**kw).eval()
    gen = torch.Generator().manual_seed(102)
    ins = [torch.rand(*s, generator=gen) for s in shapes]
    sd = {{k: v.detach().clone() for k, v in model.state_dict().items()}}
    with torch.no_grad():
        want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins])
    model.to("cuda:0")
    out = {{}}
    for name, grad in (("inference", False), ("taping", True)):
        with torch.set_grad_enabled(grad):
            runs = [model([t.to("cuda:0") for t in ins]).detach().cpu() for _ in range(3)]
        out[name] = {{"finite": bool(all(torch.isfinite(r).all() for r in runs)),
                     "deterministic": bool(all(torch.equal(runs[0], r) for r in runs[1:])),
                     "rel_err": float((runs[0] - want).abs().max() / want.abs().max())}}
    print("RESULT " + json.dumps(out))
""")


def _run_poisoned(kw, shapes):
    env = dict(os.environ, HN_POISON_WS="1")
    out = subprocess.run([sys.executable, "-c", _POISON_SCRIPT.format(root=ROOT, kw=kw, shapes=shapes)], cwd=ROOT, capture_output=True,
                         text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.parametrize("n_tokens,b", [(200, 2), (600, 2), (3000, 1)], ids=["unsplit", "n600", "split"])
def test_explicit_cross_block_behind_a_chain_keeps_its_own_kv(n_tokens, b):
    kw = dict(n_modalities=1, channel_dims=[40], num_spatial_axes=[1], out_dims=3, depth=3, l_c=16, l_d=128, x_heads=2,
              cross_dim_head=64, self_per_cross_attn=0, num_freq_bands=2)
    res = _run_poisoned(kw, [(b, n_tokens, 40)])
    for name, r in res.items():
        assert r["finite"] and r["deterministic"], (name, r)
        assert r["rel_err"] < 1e-3, (name, r)


def test_padded_head_width_behind_a_chain():
    kw = dict(n_modalities=2, channel_dims=[30, 50], num_spatial_axes=[1, 1], out_dims=3, depth=2, l_c=32, l_d=128, x_heads=8,
              cross_dim_head=48, l_heads=8, latent_dim_head=48, num_freq_bands=2)
    res = _run_poisoned(kw, [(3, 1, 30), (3, 70, 50)])
    for name, r in res.items():
        assert r["finite"] and r["deterministic"], (name, r)
        assert r["rel_err"] < 1e-3, (name, r)


def test_two_forwards_one_backward_on_the_flat_route(hn):
    kw = dict(n_modalities=2, channel_dims=[20, 3], num_spatial_axes=[1, 2], out_dims=3, depth=2, l_c=16, l_d=32, x_heads=2, l_heads=2,
              cross_dim_head=16, latent_dim_head=16, num_freq_bands=2)
    torch.manual_seed(5)
    model = hn.HealNet(**kw).train().to(DEV)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(6)
    a = [torch.rand(3, 1, 20, generator=gen), torch.rand(3, 10, 12, 3, generator=gen)]
    c = [torch.rand(3, 1, 20, generator=gen), torch.rand(3, 10, 12, 3, generator=gen)]
    cfg = O.FusionConfig(**kw)
    want = (O.fusion_forward(sd, cfg, [t.clone() for t in a]).square().sum() + O.fusion_forward(sd, cfg, [t.clone() for t in c]).sum())
    want.backward()
    flat = hn.train.flatten_parameters(model)
    flat.zero_grad()
    loss = model([t.to(DEV) for t in a]).square().sum() + model([t.to(DEV) for t in c]).sum()
    loss.backward()                                   # two fusion_backward_into calls on one flat buffer
    assert_close(loss.detach().cpu(), want.detach(), rel=1e-4, floor=0.0, abs_floor=1e-5, what="summed loss")
    scale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, p in model.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        assert float((p.grad.cpu() - ref).abs().max()) <= 5e-4 * scale, k


@pytest.mark.parametrize("grad_mode", [False, True], ids=["inference", "taping"])
def test_unaligned_parameter_views_take_the_unfused_route(hn, grad_mode):
    kw = dict(n_modalities=2, channel_dims=[60, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2, l_c=32)
    torch.manual_seed(9)
    model = hn.HealNet(**kw).eval().to(DEV)
    # re-home some parameters the chain would read with 16-byte loads as views at an odd float offset
    moved = 0
    for name, p in model.named_parameters():
        if name.endswith(("fn.net.0.weight", "fn.to_out.0.weight", "fn.to_q.weight")) and moved < 6:
            buf = torch.empty(p.numel() + 1, device=DEV)
            buf[1:].copy_(p.data.reshape(-1))
            p.data = buf[1:].view_as(p)
            assert p.data_ptr() % 16 != 0
            moved += 1
    assert moved == 6
    gen = torch.Generator().manual_seed(10)
    ins = [torch.rand(3, 1, 60, generator=gen), torch.rand(3, 20, 24, 3, generator=gen)]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins])
    with torch.set_grad_enabled(grad_mode):
        got = model([t.to(DEV) for t in ins]).detach().cpu()
    assert_close(got, want, rel=1e-3, floor=0.0, abs_floor=1e-5, what="unaligned parameter views")


@pytest.mark.parametrize("dh,l_d,l_c", [(16, 119, 25), (63, 126, 17), (27, 62, 17), (103, 65, 16)], ids=["blca", "brca", "kirp", "ucec"])
def test_tuned_shapes_on_the_large_tile_routes_vs_oracle(hn, dh, l_d, l_c):
    """The reference's tuned TCGA shapes (config/best_hyperparams.yml: one cross head of 16 / 63 / 27 / 103, odd latent widths,
    no latent self-attention) on a patch bag large enough (b * N >= 2048 rows) for the routes round 3 opened to them: the
    128 x 128 gemm_big_kernel for the unaligned K/V projection (N = 2 dh >= 32 columns) and the LDS-staged weight-gradient
    kernel from 32 x 32 outputs (G = dKV^T z).  Logits on both forwards and every gradient vs the oracle."""
    kw = dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=l_c, l_d=l_d, x_heads=1,
              cross_dim_head=dh, l_heads=8, latent_dim_head=20, self_per_cross_attn=0, num_freq_bands=2, max_freq=2.0)
    torch.manual_seed(500 + dh)
    model = hn.HealNet(**kw).train()
    gen = torch.Generator().manual_seed(600 + dh)
    ins = [torch.rand(3, 1, 2000, generator=gen), torch.rand(3, 1024, 768, generator=gen)]
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins])
    dl = torch.randn(want.shape, generator=gen)
    (want * dl).sum().backward()
    model.to(DEV)
    dins = [t.to(DEV) for t in ins]
    with torch.no_grad():
        inf = model(list(dins)).cpu()
    assert_close(inf, want.detach(), rel=1e-3, floor=0.0, abs_floor=1e-5, what="tuned shape, inference forward")
    got = model(list(dins))
    assert_close(got.detach().cpu(), want.detach(), rel=1e-3, floor=0.0, abs_floor=1e-5, what="tuned shape, taping forward")
    (got * dl.to(DEV)).sum().backward()
    scale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, p in model.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        assert_close(p.grad.cpu(), ref, rel=2e-3, floor=1e-3, abs_floor=1e-5 * scale, what=f"tuned shape grad[{k}]")


# ------------------------------------------------------------------------------------------------
# advisor's round-5 findings
# ------------------------------------------------------------------------------------------------
def test_status_word_is_cleared_only_after_the_device_has_drained(hn):
    """ADVICE r5 (chain.hip cluster_poll): an hn_l1_adam_step that was ENQUEUED before the host consumes a lost-exchange report
    decides on the device, when it executes, whether to skip -- by reading the status word.  The consuming entry point must
    therefore drain the device BEFORE it clears the word.  Deterministic replay of the window: a long kernel holds the stream, an
    Adam step is enqueued behind it (its own poll sees a clean word), the report arrives (the host-mapped word is written from
    the host, as a kernel would), and the next entry point polls: it has to block until the queued Adam has run -- and skipped --
    before it returns HN_E_CORESIDENCY.  (Before the fix the word was zeroed at once and the queued step applied its update.)"""
    import ctypes as C
    import time
    from healnet_amd import _capi
    lib = _capi.lib()
    torch.cuda.synchronize()
    _capi.cluster_status(0, acknowledge=True)
    info = _capi.ClusterInfo()
    _capi.check(lib.hn_cluster_status(0, 0, C.byref(info)), "hn_cluster_status")
    assert info.status_word, "the device's status word does not exist"
    n = 4096
    p = torch.randn(n, device=DEV)
    g = torch.randn(n, device=DEV)
    m1, m2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p0 = p.clone()
    ws = torch.empty(lib.hn_l1_adam_workspace_bytes(), dtype=torch.uint8, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    adam = lambda: lib.hn_l1_adam_step(p.data_ptr(), g.data_ptr(), m1.data_ptr(), m2.data_ptr(), n, 0.0, 1.0, 1e-2, 0.9, 0.999, 1e-8, 1, None,   # noqa: E731
                                       ws.data_ptr(), ws.numel(), s)
    try:
        torch.cuda.synchronize()
        torch.cuda._sleep(int(2.0e8))                          # ~0.1 s of device time in front of everything below
        assert adam() == 0                                     # enqueued behind the sleep; its poll saw a clean word
        info.status_word[0] = 77                               # the report arrives while the Adam step is still queued
        t0 = time.time()
        rc = adam()                                            # the next entry point: polls, must drain, then report
        waited = time.time() - t0
        assert rc == _capi.HN_E_CORESIDENCY, rc
        assert waited > 0.02, f"the poll returned after {waited * 1e3:.1f} ms: it did not wait for the device"
        torch.cuda.synchronize()
        assert torch.equal(p, p0), "the queued optimizer step applied its update although the report was pending when it ran"
        st = _capi.cluster_status(0)
        assert not st["pending"] and not st["enabled"]
    finally:
        torch.cuda.synchronize()
        _capi.cluster_status(0, acknowledge=True)
        _capi.cluster_config(0, enable=True, timeout_us=0)


def test_in_place_parameter_update_between_forward_and_backward_raises(hn):
    """ADVICE r5 (ops.py FusionTrainFn): the eager training route keeps parameters as plain attributes; what save_for_backward's
    version counters would have caught -- opt.step() before a delayed backward -- must still raise, and the tape of a finished
    backward is released (a second backward through the same graph raises as autograd's own would)."""
    torch.manual_seed(7)
    kw = dict(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=3, depth=1, l_c=16, l_d=32, x_heads=2, l_heads=2,
              cross_dim_head=16, latent_dim_head=16)
    model = hn.HealNet(**kw).train().to(DEV)
    ins = [torch.rand(2, 5, 40, device=DEV), torch.rand(2, 6, 7, 3, device=DEV)]
    y = model(list(ins))
    with torch.no_grad():
        next(model.parameters()).add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        y.sum().backward()
    y = model(list(ins))
    loss = y.sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        loss.backward()


def test_graphed_step_after_model_zero_grad_set_to_none():
    """Found with tests/test_gpu_x6.py (round 6): `model.zero_grad()` (set_to_none by default) on a flattened model leaves every
    .grad None; the eager loop repairs that in FusedL1Adam.step (FlatParameters.relink), a GraphedStep created right behind it did
    not -- its warm-up backward handed ordinary gradients to autograd, the capture recorded `p.grad += g` into tensors of its own
    that flat.zero_grad() never clears (4 x the gradient after three warm-ups and one replay, none of it where the fused optimizer
    reads), and with an output tensor of an earlier eager step still referenced the capture could not even end (SIGSEGV inside
    hipStreamEndCapture: the old AccumulateGrad nodes belong to another stream).  GraphedStep now relinks before its warm-up runs,
    checks the layout again behind them, and refuses to capture when torch reports an AccumulateGrad stream mismatch in them.
    In a subprocess: the failure mode being guarded against is a crash."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = f"""
import sys, torch
sys.path.insert(0, {root!r})
import healnet_amd as hn
model = hn.HealNet(n_modalities=2, channel_dims=[300, 60], num_spatial_axes=[1, 1], out_dims=3, depth=2, x_heads=2).train().to("cuda:0")
flat = hn.train.flatten_parameters(model)
gen = torch.Generator().manual_seed(1)
ins = [torch.rand(3, 1, 300, generator=gen).cuda(), torch.rand(3, 301, 60, generator=gen).cuda()]
y = model(list(ins))                    # (kept alive on purpose)
y.sum().backward()
want = {{k: p.grad.detach().clone() for k, p in model.named_parameters()}}
model.zero_grad()                       # set_to_none: every .grad leaves the flat buffer
g = hn.train.GraphedStep(model, lambda logits: logits.sum(), list(ins), ())
g(ins, ())
torch.cuda.synchronize()
ok = all(p.grad is not None and torch.equal(p.grad, want[k]) for k, p in model.named_parameters())
inflat = flat.direct_offsets(list(flat.views)) is not None
print("REPLAYED", ok, inflat)
"""
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    assert "REPLAYED True True" in r.stdout, r.stdout + r.stderr[-800:]
