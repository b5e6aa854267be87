"""GPU: the fused latent backward (bchain.hip + gemm_tn_lds_multi_kernel; VERDICT r2 Next 3).

One launch per chain runs, in backward order, the projection backward of the attention block behind a feed-forward block
(dx_hat = dQ W_q + dKV W_kv, LayerNorm backward, residual), the feed-forward block's backward (recompute, gate backward, both
products, LayerNorm backward) and the out-projection backward of the attention block in front of it (LeakyReLU sign, dO); the
weight gradients of the chain follow in one batched launch + one fixed-order reduce.  Autograd of healnet.py:236-245,
:313-321, :339-351, :385, :403-405, :426.

  * whole-model gradients vs oracle autograd on models that take the chains (l_d = 128): 3 modalities exercising all three
    attention bindings (one-token, shared-context image, explicit patch bag), SELU / GELU gates, other head shapes, weight
    tying, a missing modality, a key mask, fewer than 256 rows (the transposed-weight cache at small row counts);
  * fused route == the per-block launches (HN_NO_BCHAIN=1 in a subprocess) to fp32 summation noise;
  * bitwise reproducible gradients (fixed-order reductions everywhere).
"""
import json
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

CASES = {
    # name: (kwargs, input shapes, batch, masked)
    "three_bindings": (dict(n_modalities=3, channel_dims=[600, 3, 96], num_spatial_axes=[1, 2, 1], out_dims=4, depth=2, l_c=32),
                       [(1, 600), (20, 18, 3), (150, 96)], 3, False),
    "gelu_heads4x32": (dict(n_modalities=2, channel_dims=[40, 3], num_spatial_axes=[1, 2], out_dims=3, depth=2, l_c=48, x_heads=4,
                            cross_dim_head=32, l_heads=4, latent_dim_head=32, snn=False), [(1, 40), (16, 12, 3)], 6, False),
    "tied_depth3": (dict(n_modalities=2, channel_dims=[2000, 64], num_spatial_axes=[1, 1], out_dims=4, depth=3, l_c=16,
                         weight_tie_layers=True), [(1, 2000), (70, 64)], 16, False),
    "missing_modality": (dict(n_modalities=3, channel_dims=[50, 3, 64], num_spatial_axes=[1, 2, 1], out_dims=3, depth=2, l_c=32),
                         [(1, 50), None, (90, 64)], 8, False),
    "masked_bags": (dict(n_modalities=2, channel_dims=[30, 48], num_spatial_axes=[1, 1], out_dims=3, depth=2, l_c=32, x_heads=2,
                         cross_dim_head=64), [(60, 30), (60, 48)], 4, True),
    "no_self_block": (dict(n_modalities=2, channel_dims=[700, 3], num_spatial_axes=[1, 2], out_dims=4, depth=2, l_c=64,
                           self_per_cross_attn=0), [(1, 700), (10, 14, 3)], 4, False),
    "inner256_dh128": (dict(n_modalities=1, channel_dims=[80], num_spatial_axes=[1], out_dims=2, depth=2, l_c=16, x_heads=2,
                            cross_dim_head=128, l_heads=2, latent_dim_head=64), [(50, 80)], 5, False),
}


def _make(name):
    import healnet_amd as hn
    kw, shapes, b, masked = CASES[name]
    gen = torch.Generator().manual_seed(301)
    ins = [None if s is None else torch.rand(b, *s, generator=gen) for s in shapes]
    mask = None
    if masked:
        n = shapes[0][0]
        mask = torch.rand(b, n, generator=gen) > 0.3
        mask[:, 0] = True
    torch.manual_seed(302)
    model = hn.HealNet(**kw).train()
    with torch.no_grad():                       # non-trivial LayerNorm affines and biases
        for k, p in model.named_parameters():
            if k.endswith("norm.weight") or k.endswith("norm_context.weight"):
                p.add_(0.2 * torch.randn(p.shape))
            elif k.endswith("bias"):
                p.add_(0.1 * torch.randn(p.shape))
    return model, kw, ins, mask, gen


@pytest.mark.parametrize("name", sorted(CASES))
def test_model_gradients_through_the_backward_chains_vs_oracle(name):
    model, kw, ins, mask, gen = _make(name)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    # tied blocks: the state_dict lists one tensor under several keys; the oracle's leaves are per key, so the gradient of a
    # tied parameter is the SUM over its aliases
    alias = {}
    for k, v in model.state_dict().items():
        alias.setdefault(v.data_ptr(), []).append(k)
    want = O.fusion_forward(sd, O.FusionConfig(**kw), ins, mask=mask)
    dl = torch.randn(want.shape, generator=gen)
    (want * dl).sum().backward()
    ref_grad = {}
    for keys in alias.values():
        tot = sum((sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in keys)
        for k in keys:
            ref_grad[k] = tot
    model.to(DEV)
    dins = [None if t is None else t.to(DEV) for t in ins]
    dmask = None if mask is None else mask.to(DEV)
    got = model(list(dins), mask=dmask)
    assert_close(got.detach().cpu(), want.detach(), rel=1e-3, what=name + ".fwd_train")
    (got * dl.to(DEV)).sum().backward()
    first = {}
    scale = max(float(v.abs().max()) for v in ref_grad.values())
    for k, p in model.named_parameters():
        ref = ref_grad[k]
        assert p.grad is not None, k
        assert_close(p.grad.cpu(), ref, rel=2e-3, floor=1e-3, abs_floor=1e-5 * scale, what=f"{name}.grad[{k}]")
        first[k] = p.grad.clone()
    # bitwise reproducible: the same step again
    model.zero_grad(set_to_none=True)
    (model(list(dins), mask=dmask) * dl.to(DEV)).sum().backward()
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, first[k]), f"{name}: gradient of {k} is not bitwise reproducible"


_SCRIPT = textwrap.dedent("""
    import json, sys, torch
    sys.path.insert(0, {root!r})
    sys.path.insert(0, {root!r} + "/tests")
    import test_gpu_bchain as T
    model, kw, ins, mask, gen = T._make({name!r})
    model.to("cuda:0")
    out = model([None if t is None else t.to("cuda:0") for t in ins], mask=None if mask is None else mask.to("cuda:0"))
    dl = torch.randn(out.shape, generator=gen).to("cuda:0")
    (out * dl).sum().backward()
    torch.save({{k: p.grad.cpu() for k, p in model.named_parameters()}}, {dst!r})
""")


@pytest.mark.parametrize("name", ["three_bindings", "tied_depth3"])
def test_fused_backward_equals_the_per_block_launches(name, tmp_path):
    grads = {}
    for tag, env in (("fused", {}), ("unfused", {"HN_NO_BCHAIN": "1"})):
        dst = str(tmp_path / f"{tag}.pt")
        out = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT, name=name, dst=dst)], cwd=ROOT, capture_output=True, text=True,
                             timeout=600, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stderr[-3000:]
        grads[tag] = torch.load(dst)
    scale = max(float(v.abs().max()) for v in grads["unfused"].values())
    differ = 0
    for k, ref in grads["unfused"].items():
        got = grads["fused"][k]
        assert float((got - ref).abs().max()) <= 2e-5 * scale + 1e-4 * float(ref.abs().max()), k
        differ += int(not torch.equal(got, ref))
    assert differ > 0, "the fused route produced bit-identical gradients: HN_NO_BCHAIN did not change the route?"


@pytest.mark.parametrize("switch", ["HN_NO_TN_BATCH", "HN_NO_ONETOK_FUSED", "HN_NO_NARROW_COLSUM", "HN_FORCE_TORCH_OPS", "HN_NO_DQ_DIRECT"])
@pytest.mark.parametrize("name", ["three_bindings", "tied_depth3"])
def test_round5_routes_equal_their_predecessors(name, switch, tmp_path):
    """Round 5: the chains' weight-gradient products of a layer in one batched launch (duplicate destinations folded in one reduce
    pass), the one-token block's backward in four launches, the narrow column sums of the shared-context backward, and the eager route
    behind a plain autograd.Function, and the single-split dQ kernel writing its finished rows itself instead of a partial + dq_reduce
    -- each against the route it replaced (its switch, in a subprocess): gradients equal to fp32 summation noise; bit for bit for
    the host route and for the direct dQ rows."""
    grads = {}
    for tag, env in (("new", {}), ("old", {switch: "1"})):
        dst = str(tmp_path / f"{tag}.pt")
        out = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT, name=name, dst=dst)], cwd=ROOT, capture_output=True, text=True,
                             timeout=600, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stderr[-3000:]
        grads[tag] = torch.load(dst)
    scale = max(float(v.abs().max()) for v in grads["old"].values())
    for k, ref in grads["old"].items():
        got = grads["new"][k]
        if switch in ("HN_FORCE_TORCH_OPS", "HN_NO_DQ_DIRECT"):
            assert torch.equal(got, ref), k
        else:
            assert float((got - ref).abs().max()) <= 2e-5 * scale + 1e-4 * float(ref.abs().max()), k


@pytest.mark.parametrize("b,l_c", [(8, 128), (16, 128), (4, 32)], ids=["64tiles_c4", "128tiles_c2", "8tiles_c4"])
def test_cluster_mode_gradients_bitwise_reproducible_and_vs_oracle(b, l_c):
    """Row-tile counts that run the forward AND backward chains as clusters (4 workgroups per tile up to 64 tiles, 2 up to 128; two
    exchanges per backward chain): ten repetitions of the step give bitwise identical gradients (ordered exchanges, fixed
    summation order), and they agree with oracle autograd."""
    import healnet_amd as hn
    kw = dict(n_modalities=2, channel_dims=[300, 64], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=l_c)
    torch.manual_seed(401)
    model = hn.HealNet(**kw).train()
    gen = torch.Generator().manual_seed(402)
    ins = [torch.rand(b, 1, 300, generator=gen), torch.rand(b, 40, 64, generator=gen)]
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins])
    dl = torch.randn(want.shape, generator=gen)
    (want * dl).sum().backward()
    model.to(DEV)
    dins = [t.to(DEV) for t in ins]
    first = None
    for rep in range(10):
        model.zero_grad(set_to_none=True)
        out = model(list(dins))
        (out * dl.to(DEV)).sum().backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        if first is None:
            first = grads
            assert_close(out.detach().cpu(), want.detach(), rel=1e-3, what="cluster mode forward")
        else:
            for k in grads:
                assert torch.equal(grads[k], first[k]), f"repetition {rep}: gradient of {k} differs"
    scale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for k, g in first.items():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        assert_close(g.cpu(), ref, rel=2e-3, floor=1e-3, abs_floor=1e-5 * scale, what=f"cluster mode grad[{k}]")
