"""GPU: the bf16-MFMA K/V projection of explicit (patch-bag) contexts under core_precision='bf16' (gemm_bf16.hip).

The product (z gamma + beta) W_kv^T of a cross block whose context is NOT the shared-context (rank-D) binding runs with both
operands rounded to bf16 once and fp32 accumulation when the bag is large enough (>= 2048 context rows, D >= 256, 2 * inner a
multiple of 128); the bias term W beta stays fp32.  Held to the bf16 configuration's tolerance (2e-2 max-norm against the fp32
oracle) -- observed deviations are ~1e-4 -- and checked to be the path that actually ran: a model with explicit
bindings only is bit-identical under 'fp32' and 'bf16' wherever the projection is NOT eligible, and differs where it is.
Shapes cover a k tail (D % 64 != 0, D % 4 != 0), a row tail (b * N % 128 != 0), a key-padding mask, two patch-bag modalities
and the unchanged training forward (training always projects in fp32).  With heads of 64 the projection writes bf16 K / V images
and the block's attention core runs on bf16 MFMA too (attn_core_bf16_kernel<4, 2, 1, true>): every default-heads case here (token
counts that are multiples of 32, of 4 only -- 1100 -- and odd -- 1101: zeroed pad slots of the last V tile, element-wise stores --
ragged query tiles, masks, graph replay, poisoned workspace); cross heads of 32 and the staged one-head model keep the fp32 core
behind the bf16 projection.
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


def _case(hn, kw, shapes, seed):
    torch.manual_seed(seed)
    model = hn.HealNet(**kw).eval()
    gen = torch.Generator().manual_seed(seed + 1)
    ins = [torch.rand(*s, generator=gen) for s in shapes]
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model, ins, sd


CASES = {
    # channels 300 + 5 position columns = D 305: k tail 49 of 64, D % 4 = 1; rows 2 * 1100 = 2200 = 17 tiles + 24 rows; 1100 = 34 blocks of 32 + 12
    "k_and_row_tails": (dict(n_modalities=2, channel_dims=[40, 300], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=32, l_d=128,
                             num_freq_bands=2, max_freq=2.0), [(2, 1, 40), (2, 1100, 300)]),
    # two heads of 64 -> N = 256 (two column tiles); D = 251 + 5 = 256 exactly (no tail); 2048 rows exactly
    "odd_token_count": (dict(n_modalities=2, channel_dims=[40, 300], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=32, l_d=128,
                             num_freq_bands=2, max_freq=2.0), [(2, 1, 40), (2, 1101, 300)]),
    "two_column_tiles": (dict(n_modalities=1, channel_dims=[251], num_spatial_axes=[1], out_dims=3, depth=2, l_c=16, l_d=64, x_heads=2,
                              l_heads=2, latent_dim_head=32, num_freq_bands=2, max_freq=2.0), [(1, 2048, 251)]),
    # default heads (8 x 64), 2 x 1024 tokens (N % 32 == 0): projection straight into the bf16 K / V images + the explicit bf16 core
    "bf16_explicit_core": (dict(n_modalities=2, channel_dims=[40, 300], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=32, l_d=128,
                                num_freq_bands=2, max_freq=2.0), [(2, 1, 40), (2, 1024, 300)]),
    # ... with l_c = 40 (Lp = 48: ragged query tiles), one layer, no latent self-attention, 3 x 704 tokens (22 blocks of 32)
    "bf16_explicit_core_ragged_queries": (dict(n_modalities=1, channel_dims=[300], num_spatial_axes=[1], out_dims=4, depth=1, l_c=40, l_d=64,
                                               l_heads=2, latent_dim_head=32, self_per_cross_attn=0, num_freq_bands=2, max_freq=2.0),
                                          [(3, 704, 300)]),
    # a STAGED model (odd latent width, one cross head of 63 run as a zero-padded head of 64: DESIGN 4.10): N = 2 x 64 = one column tile
    "staged_one_head": (dict(n_modalities=2, channel_dims=[40, 300], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=17, l_d=126,
                             x_heads=1, cross_dim_head=63, l_heads=8, latent_dim_head=20, self_per_cross_attn=0, num_freq_bands=2,
                             max_freq=2.0), [(3, 1, 40), (3, 1024, 300)]),
    # two patch bags of different widths in one model, cross head width 32 x 4 heads -> N = 256
    "two_bags": (dict(n_modalities=3, channel_dims=[30, 280, 400], num_spatial_axes=[1, 1, 1], out_dims=4, depth=2, l_c=32, l_d=128,
                      x_heads=4, cross_dim_head=32, num_freq_bands=2, max_freq=2.0), [(2, 1, 30), (2, 1024, 280), (2, 1200, 400)]),
}


@pytest.mark.parametrize("name", list(CASES))
def test_bf16_projection_vs_oracle_and_vs_fp32_route(hn, name):
    kw, shapes = CASES[name]
    model, ins, sd = _case(hn, kw, shapes, 700 + len(name))
    with torch.no_grad():
        want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins])
    model.to(DEV)
    dins = [t.to(DEV) for t in ins]
    with torch.no_grad():
        model.core_precision = "fp32"
        full = model(list(dins)).cpu()
        model.core_precision = "bf16"
        low = model(list(dins)).cpu()
        again = model(list(dins)).cpu()
    scale = float(want.abs().max())
    assert_close(full, want, rel=1e-3, floor=0.0, abs_floor=1e-5, what=f"{name}: fp32 route")
    assert torch.isfinite(low).all()
    assert torch.equal(low, again), "the bf16 projection must be deterministic"
    assert not torch.equal(low, full), "core_precision='bf16' did not change the explicit binding's K/V projection"
    assert float((low - want).abs().max()) <= 2e-2 * scale, (name, float((low - want).abs().max()) / scale)
    # far inside the tolerance in practice: K and V carry one bf16 rounding of z and of W, averaged over D terms
    assert float((low - full).abs().max()) <= 2e-3 * scale, (name, float((low - full).abs().max()) / scale)


def test_small_bags_keep_the_fp32_projection(hn):
    """Below the eligibility thresholds (1000 context rows) nothing changes under core_precision='bf16'."""
    kw = dict(n_modalities=2, channel_dims=[40, 300], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=32, l_d=128, num_freq_bands=2,
              max_freq=2.0)
    model, ins, _ = _case(hn, kw, [(2, 1, 40), (2, 500, 300)], 811)
    model.to(DEV)
    dins = [t.to(DEV) for t in ins]
    with torch.no_grad():
        model.core_precision = "fp32"
        full = model(list(dins)).cpu()
        model.core_precision = "bf16"
        low = model(list(dins)).cpu()
    assert torch.equal(low, full)


@pytest.mark.parametrize("n_tokens", [1100, 1056, 1101], ids=["ragged_block", "whole_blocks", "odd_tokens"])
def test_masked_bag_and_embeddings(hn, n_tokens):
    """Key-padding mask on the patch bag (zero-padded bags) + return_embeddings: the projection covers the padded
    tokens too (they are masked in the core), the latent array agrees with the oracle at the bf16 tolerance.  (One modality: the
    reference hands the same mask to every cross block, healnet.py:236.)"""
    kw = dict(n_modalities=1, channel_dims=[300], num_spatial_axes=[1], out_dims=4, depth=2, l_c=32, l_d=128, num_freq_bands=2,
              max_freq=2.0)
    model, ins, sd = _case(hn, kw, [(2, n_tokens, 300)], 905)
    mask = torch.ones(2, n_tokens, dtype=torch.bool)
    mask[0, 700:] = False
    mask[1, 1033:] = False
    mask[1, 5:40] = False
    with torch.no_grad():
        want = O.fusion_forward(sd, O.FusionConfig(**kw), [t.clone() for t in ins], mask=mask, return_embeddings=True)
    model.to(DEV)
    with torch.no_grad():
        model.core_precision = "fp32"
        full = model([t.to(DEV) for t in ins], mask=mask.to(DEV), return_embeddings=True).cpu()
        model.core_precision = "bf16"
        got = model([t.to(DEV) for t in ins], mask=mask.to(DEV), return_embeddings=True).cpu()
    assert not torch.equal(got, full)
    assert float((got - want).abs().max()) <= 2e-2 * float(want.abs().max())


def test_graph_replay_with_the_bf16_projection(hn):
    """HealNet.capture() on a fresh model (the FIRST bf16 projection launch of the process for this kernel configuration happens
    inside the stream capture: the kernel's dynamic-LDS attribute is set there): replay == eager forward bit for bit on new values."""
    kw = dict(n_modalities=2, channel_dims=[40, 300], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=32, l_d=128, num_freq_bands=2,
              max_freq=2.0)
    torch.manual_seed(31)
    model = hn.HealNet(**kw, core_precision="bf16").eval().to(DEV)
    gen = torch.Generator().manual_seed(32)
    shapes = [(2, 1, 40), (2, 1024, 300)]
    graph = model.capture([torch.rand(*s, generator=gen).to(DEV) for s in shapes])
    for trial in range(2):
        ins = [torch.rand(*s, generator=gen).to(DEV) for s in shapes]
        with torch.no_grad():
            got = graph(list(ins)).clone()
            eager = model(list(ins))
        assert torch.equal(got, eager), f"trial {trial}"
    model.core_precision = "fp32"
    with torch.no_grad():
        full = model(list(ins))
    assert not torch.equal(full, eager) and float((full - eager).abs().max()) <= 2e-3 * float(full.abs().max())


def test_training_forward_is_untouched(hn):
    """Training always projects in fp32 (the backward differentiates that product): logits and gradients of a
    core_precision='bf16' model in train mode are bit-identical to the fp32 model's."""
    kw = dict(n_modalities=2, channel_dims=[40, 300], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=32, l_d=128, num_freq_bands=2,
              max_freq=2.0)
    outs = []
    for prec in ("fp32", "bf16"):
        model, ins, _ = _case(hn, kw, [(2, 1, 40), (2, 1100, 300)], 1003)
        model.core_precision = prec
        model.train().to(DEV)
        y = model([t.to(DEV) for t in ins])
        y.square().sum().backward()
        outs.append((y.detach().cpu(), {k: p.grad.cpu().clone() for k, p in model.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


_POISON = textwrap.dedent("""
    import json, sys, torch
    sys.path.insert(0, {root!r})
    import healnet_amd as hn
    from healnet_amd import _rt
    assert _rt._POISON
    kw = dict(n_modalities=2, channel_dims=[40, 300], num_spatial_axes=[1, 1], out_dims=4, depth=2, l_c=32, l_d=128, num_freq_bands=2,
              max_freq=2.0)
    torch.manual_seed(3)
    model = hn.HealNet(**kw, core_precision="bf16").eval().to("cuda:0")
    gen = torch.Generator().manual_seed(4)
    ins = [torch.rand(2, 1, 40, generator=gen).cuda(), torch.rand(2, {n_tokens}, 300, generator=gen).cuda()]
    with torch.no_grad():
        runs = [model(list(ins)).cpu() for _ in range(3)]
    print("RESULT " + json.dumps(dict(finite=bool(all(torch.isfinite(r).all() for r in runs)),
                                      same=bool(all(torch.equal(runs[0], r) for r in runs[1:])))))
""")


@pytest.mark.parametrize("n_tokens", [1100, 1024, 1101], ids=["ragged_block", "whole_blocks", "odd_tokens"])
def test_poisoned_workspace(n_tokens):
    """HN_POISON_WS=1: every call starts from an all-NaN workspace -- the staged weight image, its zero pad columns, the bias row and
    the context's pad columns (never read past D - 1), and with the explicit bf16 core the K / V / query images (pad query rows
    included) must all be produced / masked by the call itself."""
    import json
    env = dict(os.environ, HN_POISON_WS="1")
    out = subprocess.run([sys.executable, "-c", _POISON.format(root=ROOT, n_tokens=n_tokens)], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert res["finite"] and res["same"], res
