"""GPU: parity against the CPU oracle AT THE SIZES the BASELINE configs run -- the size-gated kernel routes (split-K
weight-gradient GEMM at K = 32 768, the 128x128 K/V projection + K/V tape, the multi-split dQ core at N = 50 176, the
depth-8 four-modality schedule) are only reached there, so the small-size gradient tests do not cover them.

  cfg4  b=8, N=4096 patch bag   : logits + every parameter gradient vs oracle autograd        (healnet.py:190-250, main.py:464)
  cfg2  b=2, 224x224x3 image    : logits + every parameter gradient vs oracle autograd
  cfg5  4 modalities, depth 8   : logits vs oracle (full-size patch bags; volume 4x224x224 at b=2, 12x224x224 at b=1)
  cfg3  b=16 bf16 tensors + core: sample 0 vs the ORACLE (fp32, on the same bf16-rounded inputs) at 2e-2
"""
import pytest
import torch

from conftest import assert_close, rel_err
from oracle import healnet_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


@pytest.fixture(scope="module")
def hn():
    import healnet_amd
    return healnet_amd


def _grad_parity(hn, kw, ins, seed, what, mask=None):
    """Whole-model gradients of sum(logits * dl): HIP tape forward + hn_fusion_backward vs torch autograd of the oracle."""
    torch.manual_seed(seed)
    model = hn.HealNet(**kw).train()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    want = O.fusion_forward(sd, O.FusionConfig(**kw), ins, mask=mask)
    gen = torch.Generator().manual_seed(seed + 1)
    dl = torch.randn(want.shape, generator=gen)
    (want * dl).sum().backward()
    model.to(DEV)
    got = model([None if t is None else t.to(DEV) for t in ins], mask=None if mask is None else mask.to(DEV))
    assert_close(got.detach().cpu(), want.detach(), rel=TOL, floor=0.0, abs_floor=1e-5, what=what + ".fwd_train")
    (got * dl.to(DEV)).sum().backward()
    # At these sizes ~10^6 LeakyReLU / SELU pre-activations exist per block, so a few lie within fp32 rounding of the kink
    # and two correct forwards disagree on their side: such a flip perturbs a handful of gradient elements by
    # O(1e-3) of the tensor's scale.  Hence two criteria per parameter: every element within 5e-3 of the scale (a kink flip
    # passes, a wrong kernel route does not), and the relative L2 error <= 3e-4 (isolated flips vanish in it, a systematic
    # error -- a dropped split, a mis-scaled partial -- does not).
    n, worst = 0, []
    for k, p in model.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        assert p.grad is not None, k
        got = p.grad.double().cpu()
        scale = float(ref.abs().max().clamp_min(1e-30))
        linf = float((got - ref.double()).abs().max()) / scale
        l2 = float((got - ref.double()).norm() / ref.double().norm().clamp_min(1e-30))
        outliers = int(((got - ref.double()).abs() > 5e-4 * scale).sum())
        worst.append((linf, l2, k, outliers, p.numel()))
        n += p.numel()
    bad = [w for w in worst if w[0] > 5e-3 or w[1] > 3e-4]
    # ceiling on the kink allowance: the elements beyond 5e-4 of their tensor's scale must stay a handful of
    # isolated flips -- at most 1e-4 of all gradient elements -- so a small systematic error cannot hide under it
    n_out = sum(w[3] for w in worst)
    assert n_out <= max(64, int(1e-4 * n)), f"{what}: {n_out} of {n} gradient elements beyond 5e-4 of their tensor's scale"
    top = sorted(worst, reverse=True)[:3]
    print(f"{what}: worst max-norm {top[0][0]:.2e} ({top[0][2]}: {top[0][3]} of {top[0][4]} elements beyond 5e-4 of the scale), "
          f"worst L2 {max(w[1] for w in worst):.2e}; next: {[(f'{w[0]:.1e}', w[2], w[3]) for w in top[1:]]}")
    assert not bad, f"{what}: {len(bad)} parameter gradients off: {sorted(bad, reverse=True)[:5]}"
    return n


def test_cfg4_full_size_gradients_vs_oracle_autograd(hn):
    """BASELINE configs[3] at its real per-GPU size: omic (8,1,2000) + WSI bag (8,4096,768), default model (11.9 M parameters).
    Routes only this size takes: gemm_big_kernel for the K/V projection (M = 32 768) with the K/V tape, gemm_tn_lds_kernel
    for G = dKV^T z (K = 32 768, split-K), attn_bwd_dq / attn_bwd_dkv at N = 4096 with the full split geometry."""
    kw = dict(n_modalities=2, channel_dims=[2000, 768], num_spatial_axes=[1, 1], out_dims=4)
    gen = torch.Generator().manual_seed(4101)
    ins = [torch.rand(8, 1, 2000, generator=gen), torch.rand(8, 4096, 768, generator=gen)]
    n = _grad_parity(hn, kw, ins, seed=41, what="cfg4_b8")
    assert n == sum(p.numel() for p in hn.HealNet(**kw).parameters())


def test_cfg2_full_size_gradients_vs_oracle_autograd(hn):
    """BASELINE configs[1]'s model at the full 224x224x3 image (N = 50 176), b = 2: the multi-split rank-D dQ core
    (attn_bwd_dq_kernel<1,4,true,false,3>), the packed shared context in training, the fold gradients at full N."""
    kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
    gen = torch.Generator().manual_seed(4102)
    ins = [torch.rand(2, 1, 2000, generator=gen), torch.rand(2, 224, 224, 3, generator=gen)]
    _grad_parity(hn, kw, ins, seed=42, what="cfg2_b2_224")


def test_cfg2_b32_inference_forward_every_sample_vs_oracle(hn):
    """The driver's metric exactly: the eval / no_grad forward (hn_fusion_forward: folded queries in the chains, the score-bound
    core, the merge inside the chains) of the seed-0 default model on bench.py's own inputs (generator 1234, tab then img) at
    b = 32 -- ALL 32 logit rows against the oracle (the inference forward at the headline batch had the oracle on a
    2-sample slice only).  The oracle runs sample by sample (0.6 GB of scores per sample and layer); nothing couples samples."""
    kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
    torch.manual_seed(0)
    model = hn.HealNet(**kw).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(1234)
    tab = torch.rand(32, 1, 2000, generator=gen)
    img = torch.rand(32, 224, 224, 3, generator=gen)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))
    try:
        with torch.no_grad():
            want = torch.cat([O.fusion_forward(sd, O.FusionConfig(**kw), [tab[i:i + 1], img[i:i + 1]]) for i in range(32)])
    finally:
        torch.set_num_threads(threads)
    model.to(DEV)
    with torch.no_grad():
        got = model([tab.to(DEV), img.to(DEV)])
    assert got.shape == (32, 4) and torch.isfinite(got).all()
    worst = max(rel_err(got[i].cpu(), want[i]) for i in range(32))
    print(f"cfg2 b=32 inference forward: worst per-sample rel err vs oracle {worst:.2e}")
    for i in range(32):
        assert_close(got[i].cpu(), want[i], rel=TOL, what=f"cfg2_b32.inference[{i}]")


def test_cfg2_b32_full_size_training_step_vs_oracle_autograd(hn):
    """The headline batch itself: cfg2 at b = 32 on the full 224x224x3 image -- the split geometry, the 256-workgroup chains
    (one per row tile, no cluster), the batched weight-gradient launches over 4096 rows that only this size runs.  The oracle's
    materialised scores are 0.6 GB per sample and layer, so its gradient is accumulated sample by sample (the model couples no
    two samples: per-token LayerNorm, per-row softmax, per-sample mean)."""
    kw = dict(n_modalities=2, channel_dims=[2000, 3], num_spatial_axes=[1, 2], out_dims=4)
    gen = torch.Generator().manual_seed(4132)
    b = 32
    ins = [torch.rand(b, 1, 2000, generator=gen), torch.rand(b, 224, 224, 3, generator=gen)]
    torch.manual_seed(43)
    model = hn.HealNet(**kw).train()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    dl = torch.randn(b, 4, generator=gen)
    logits = []
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))                # (a 256-core host runs this small-GEMM mix far slower with every core)
    try:
        for i in range(b):
            out = O.fusion_forward(sd, O.FusionConfig(**kw), [t[i:i + 1] for t in ins])
            (out * dl[i:i + 1]).sum().backward()           # accumulates into sd[k].grad
            logits.append(out.detach())
    finally:
        torch.set_num_threads(threads)
    want = torch.cat(logits)
    model.to(DEV)
    got = model([t.to(DEV) for t in ins])
    assert_close(got.detach().cpu(), want, rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg2_b32.fwd_train")
    (got * dl.to(DEV)).sum().backward()
    # Criteria as in _grad_parity, plus what 16x more rows bring: ~25 M SELU-gate and ~6 M LeakyReLU pre-activations per step, so ONE
    # of them landing within fp32 rounding of the kink on the other side than the oracle's is the expected case, not the rare one.
    # tools/diag_b32_grads.py (this input): fused and per-block GPU routes agree to 7e-6 (L2), the oracle in fp32 and in fp64 to
    # 6.5e-6, and what separates the two pairs is EXACTLY rank 1 -- one row of one dW1 (128 elements, 4e-3 of the tensor's scale:
    # one SELU-derivative flip) plus its trace in that block's b1 / LayerNorm gradients and in the rows of the upstream tensors.
    # So a tensor beyond the 3e-4 L2 bound passes only if its error is such an isolated flip: >= 98 % of the error energy in the
    # four leading singular directions (2-D), L2 <= 1e-3 (1-D) -- a dropped split or a mis-scaled partial is neither.
    n_out, n, flips = 0, 0, []
    for k, p in model.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        g = p.grad.double().cpu()
        d = g - ref.double()
        scale = float(ref.abs().max().clamp_min(1e-30))
        linf = float(d.abs().max()) / scale
        l2 = float(d.norm() / ref.double().norm().clamp_min(1e-30))
        assert linf <= 5e-3, f"cfg2_b32 grad[{k}]: max-norm {linf:.2e}, L2 {l2:.2e}"
        if l2 > 3e-4:
            assert l2 <= 1e-3, f"cfg2_b32 grad[{k}]: L2 {l2:.2e}"
            if d.dim() == 2:
                sv = torch.linalg.svdvals(d)
                lead = float((sv[:4] ** 2).sum() / (sv ** 2).sum())
                assert lead >= 0.98, f"cfg2_b32 grad[{k}]: L2 {l2:.2e} and only {lead:.2f} of the error in four singular directions"
            flips.append((k, l2))
        n_out += int((d.abs() > 5e-4 * scale).sum())
        n += p.numel()
    assert len(flips) <= 6, f"cfg2_b32: {len(flips)} tensors beyond the L2 bound: {flips}"
    assert n_out <= max(64, int(1e-4 * n)), f"cfg2_b32: {n_out} of {n} gradient elements beyond 5e-4 of their tensor's scale"


def _oracle_logits(model, kw, ins, **kwargs):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        return O.fusion_forward(sd, O.FusionConfig(**kw), [None if t is None else t.float().cpu() for t in ins], **kwargs)


CFG5 = dict(n_modalities=4, channel_dims=[2000, 768, 768, 3], num_spatial_axes=[1, 1, 1, 3], out_dims=4, depth=8)


@pytest.mark.parametrize("grad_mode", [False, True], ids=["inference", "taping"])
def test_cfg5_shape_vs_oracle(hn, grad_mode):
    """BASELINE configs[4]: tab + 2 full-size WSI patch bags (4096 x 768 each) + a volume, depth 8.  The volume is cut to
    4 x 224 x 224 x 3 (N = 200 704) so that the oracle's materialised scores stay at 0.8 GB per block; b = 2."""
    torch.manual_seed(51)
    model = hn.HealNet(**CFG5).eval().to(DEV)
    gen = torch.Generator().manual_seed(4105)
    ins = [torch.rand(2, 1, 2000, generator=gen), torch.rand(2, 4096, 768, generator=gen),
           torch.rand(2, 4096, 768, generator=gen), torch.rand(2, 4, 224, 224, 3, generator=gen)]
    with torch.set_grad_enabled(grad_mode):
        y = model([t.to(DEV) for t in ins])
    want = _oracle_logits(model, CFG5, ins)
    assert_close(y.detach().cpu(), want, rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg5 logits (volume 4x224x224)")
    # a missing second bag: its iterations still run the latent self block (Appendix B-1)
    with torch.set_grad_enabled(grad_mode):
        y_m = model([ins[0].to(DEV), ins[1].to(DEV), None, ins[3].to(DEV)])
    want_m = _oracle_logits(model, CFG5, [ins[0], ins[1], None, ins[3]])
    assert_close(y_m.detach().cpu(), want_m, rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg5 logits, bag 2 missing")


def test_cfg5_full_size_one_sample_vs_oracle(hn):
    """The same model on ONE sample at the config's full sizes (volume 12 x 224 x 224 x 3, N = 602 112)."""
    torch.manual_seed(52)
    model = hn.HealNet(**CFG5).eval().to(DEV)
    gen = torch.Generator().manual_seed(4106)
    ins = [torch.rand(1, 1, 2000, generator=gen), torch.rand(1, 4096, 768, generator=gen),
           torch.rand(1, 4096, 768, generator=gen), torch.rand(1, 12, 224, 224, 3, generator=gen)]
    with torch.no_grad():
        y = model([t.to(DEV) for t in ins])
    want = _oracle_logits(model, CFG5, ins)
    assert_close(y.cpu(), want, rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg5 logits (full size, b=1)")


def test_cfg3_b16_bf16_vs_oracle(hn):
    """BASELINE configs[2] (b = 16, bf16 tensors, bf16 MFMA core): sample 0 against the fp32 ORACLE on the same
    bf16-rounded inputs, tolerance 2e-2 max-norm -- not against this build's own fp32 core."""
    torch.manual_seed(0)
    kw = dict(n_modalities=3, channel_dims=[2000, 3, 3], num_spatial_axes=[1, 2, 3], out_dims=4)
    low = hn.HealNet(**kw, core_precision="bf16").eval().to(DEV)
    gen = torch.Generator().manual_seed(1234)
    b = 16
    tab = torch.rand(b, 1, 2000, generator=gen).to(torch.bfloat16)
    img = torch.rand(b, 224, 224, 3, generator=gen).to(torch.bfloat16)
    vol = torch.rand(b, 12, 224, 224, 3, generator=gen).to(torch.bfloat16)
    with torch.no_grad():
        y = low([tab.to(DEV), img.to(DEV), vol.to(DEV)])
    want = _oracle_logits(low, kw, [tab[:1], img[:1], vol[:1]])
    assert rel_err(y[:1].cpu(), want) <= 2e-2, rel_err(y[:1].cpu(), want)
    # the fp32 core on the same inputs must sit (much) closer to the oracle than the bf16 tolerance
    ref = hn.HealNet(**kw).eval().to(DEV)
    ref.load_state_dict(low.state_dict())
    with torch.no_grad():
        y32 = ref([tab[:1].to(DEV), img[:1].to(DEV), vol[:1].to(DEV)])
    assert_close(y32.cpu(), want, rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg3 fp32 core vs oracle (bf16 tensors)")
