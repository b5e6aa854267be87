"""GPU: parity against the CPU oracle AT THE SIZES the BASELINE configs run -- the size-gated kernel routes (split-K
weight-gradient GEMM at K = 32 768, the 128x128 K/V projection + K/V tape, the multi-split dQ core at N = 50 176, the
depth-8 four-modality schedule) are only reached there, so the small-size gradient tests do not cover them.

  cfg4  b=8, N=4096 patch bag   : logits + every parameter gradient vs oracle autograd        (healnet.py:190-250, main.py:464)
  cfg2  b=2, 224x224x3 image    : logits + every parameter gradient vs oracle autograd
  cfg2  b=32 (the headline)     : all 32 logit rows, and the training step's gradients, vs the REFERENCE's own outputs
  cfg5  4 modalities, depth 8   : logits vs the REFERENCE (full-size patch bags; volume 4x224x224 at b=2, 12x224x224 at b=1)
  cfg3  b=16 bf16 tensors + core: sample 0 vs the REFERENCE (fp32, on the same bf16-rounded inputs) at 2e-2

Round 6: the expectations of the four largest cases are committed fixtures generated in the build container from
/root/reference itself (tools/gen_goldens_fullsize.py -> tests/golden/g10_*.npz) instead of oracle forwards / autograd passes re-run
on the GPU box -- those were 230 s of the suite's 617 s, all of it host time -- which also pins the headline configuration directly to
reference outputs rather than through the oracle.  One oracle comparison per configuration stays (cfg2 b = 2 and cfg4 gradients
here, cfg5 below, cfg3 in tests/test_gpu_model.py) so that the oracle route remains exercised at these shapes.
"""
import os

import pytest
import torch

import fullsize_fixtures as F
from conftest import GOLD, assert_close, rel_err
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


def _golden(name):
    import numpy as np
    with np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False) as z:
        return {k: (z[k] if z[k].dtype.kind in "US" else torch.from_numpy(np.asarray(z[k]))) for k in z.files}


def test_cfg2_b32_inference_forward_every_sample_vs_reference(hn):
    """The driver's metric exactly: the eval / no_grad forward (hn_fusion_forward: the layer chains with the latent self-attention
    inside, folded queries, the score-bound core, the merge inside the chains) of the seed-0 default model on bench.py's own inputs
    (generator 1234, tab then img) at b = 32 -- ALL 32 logit rows against what the REFERENCE produced on them
    (tests/golden/g10_cfg2_b32_bench.npz, tools/gen_goldens_fullsize.py; until round 5 the oracle was re-run here, 20 s of host
    time).  Both routes of the latent side are held to it: the second pass runs after cluster mode was switched off, which is the
    per-block chains + the self-attention core."""
    from healnet_amd import _capi
    want = _golden("g10_cfg2_b32_bench")["logits"]
    torch.manual_seed(0)
    model = hn.HealNet(**F.CFG2).eval().to(DEV)
    tab, img = F.bench_inputs(32)
    for route in ("layer chains", "per-block chains"):
        if route == "per-block chains":
            _capi.cluster_config(0, enable=False)
        try:
            with torch.no_grad():
                got = model([tab.to(DEV), img.to(DEV)])
        finally:
            _capi.cluster_config(0, enable=True)
        assert got.shape == (32, 4) and torch.isfinite(got).all()
        worst = max(rel_err(got[i].cpu(), want[i]) for i in range(32))
        print(f"cfg2 b=32 inference forward ({route}): worst per-sample rel err vs the reference {worst:.2e}")
        for i in range(32):
            assert_close(got[i].cpu(), want[i], rel=TOL, what=f"cfg2_b32.inference[{i}] ({route})")


def test_cfg2_b32_full_size_training_step_vs_reference_gradients(hn):
    """The headline batch itself: cfg2 at b = 32 on the full 224x224x3 image -- the split geometry, the 256-workgroup chains
    (one per row tile, no cluster), the batched weight-gradient launches over 4096 rows that only this size runs -- against the
    REFERENCE's logits and gradients of sum(logits * dl), accumulated there sample by sample (tools/gen_goldens_fullsize.py) and
    committed in compressed form (tests/fullsize_fixtures.py: norm, scale, a 128-bucket count sketch and 1024 exact elements per
    parameter tensor).  Criteria as in _grad_parity, on what the fixture allows:
      * sampled elements: every one within 5e-3 of the tensor's scale; those beyond 5e-4 stay isolated (<= 1e-4 of the sample,
        at least 8 allowed in all);
      * the sketch's estimate of |g - g_ref| / |g_ref| <= 3e-4 -- or <= 1e-3 for at most 6 tensors: ~25 M SELU-gate and ~6 M
        LeakyReLU pre-activations per step make ONE kink flip against the reference the expected case (tools/diag_b32_grads.py:
        exactly rank 1, one row of one dW1 plus its trace upstream), a dropped split or a mis-scaled partial is far beyond it."""
    gold = _golden("g10_cfg2_b32_train")
    ins, dl = F.cfg2_train_inputs(32)
    torch.manual_seed(43)
    model = hn.HealNet(**F.CFG2).train().to(DEV)
    got = model([t.to(DEV) for t in ins])
    assert_close(got.detach().cpu(), gold["logits"], rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg2_b32.fwd_train")
    (got * dl.to(DEV)).sum().backward()
    keys = [str(k) for k in gold["keys"]]
    assert keys == [k for k, _ in model.named_parameters()]
    n_out, n_s, flips, worst = 0, 0, [], 0.0
    for k, p in model.named_parameters():
        g = p.grad.double().cpu().reshape(-1)
        norm, scale = float(gold[f"{k}::norm"]), max(float(gold[f"{k}::scale"]), 1e-30)
        d_sk = F.sketch(g, k) - gold[f"{k}::sketch"].double()
        l2 = float(d_sk.pow(2).sum().sqrt()) / max(norm, 1e-30)
        idx = F.sample_index(k, g.numel())
        d = (g[idx] - gold[f"{k}::vals"].double()).abs()
        assert float(d.max()) / scale <= 5e-3, f"cfg2_b32 grad[{k}]: sampled max-norm {float(d.max()) / scale:.2e}, sketch L2 {l2:.2e}"
        if l2 > 3e-4:
            assert l2 <= 1e-3, f"cfg2_b32 grad[{k}]: sketch L2 {l2:.2e}"
            flips.append((k, l2))
        worst = max(worst, l2)
        n_out += int((d > 5e-4 * scale).sum())
        n_s += idx.numel()
    print(f"cfg2 b=32 training step: worst sketch L2 {worst:.2e}, {n_out} of {n_s} sampled elements beyond 5e-4 of their scale, flips {flips}")
    assert len(flips) <= 6, f"cfg2_b32: {len(flips)} tensors beyond the L2 bound: {flips}"
    assert n_out <= max(8, int(1e-4 * n_s)), f"cfg2_b32: {n_out} of {n_s} sampled gradient elements beyond 5e-4 of their tensor's scale"


def _oracle_logits(model, kw, ins, **kwargs):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        return O.fusion_forward(sd, O.FusionConfig(**kw), [None if t is None else t.float().cpu() for t in ins], **kwargs)


CFG5 = F.CFG5


@pytest.mark.parametrize("grad_mode", [False, True], ids=["inference", "taping"])
def test_cfg5_shape_vs_reference(hn, grad_mode):
    """BASELINE configs[4]: tab + 2 full-size WSI patch bags (4096 x 768 each) + a volume, depth 8; the volume cut to
    4 x 224 x 224 x 3 (N = 200 704), b = 2 -- against the REFERENCE's logits on the same seeded model and inputs
    (g10_cfg5_cut_b2: all modalities present / the second bag missing, whose iterations still run the latent self block,
    Appendix B-1)."""
    gold = _golden("g10_cfg5_cut_b2")
    torch.manual_seed(51)
    model = hn.HealNet(**CFG5).eval().to(DEV)
    ins = F.cfg5_cut_inputs()
    with torch.set_grad_enabled(grad_mode):
        y = model([t.to(DEV) for t in ins])
    assert_close(y.detach().cpu(), gold["logits"], rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg5 logits (volume 4x224x224)")
    with torch.set_grad_enabled(grad_mode):
        y_m = model([ins[0].to(DEV), ins[1].to(DEV), None, ins[3].to(DEV)])
    assert_close(y_m.detach().cpu(), gold["logits_bag2_missing"], rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg5 logits, bag 2 missing")


def test_cfg5_full_size_one_sample_vs_reference(hn):
    """The same model on ONE sample at the config's full sizes (volume 12 x 224 x 224 x 3, N = 602 112): g10_cfg5_full_b1."""
    gold = _golden("g10_cfg5_full_b1")
    torch.manual_seed(52)
    model = hn.HealNet(**CFG5).eval().to(DEV)
    with torch.no_grad():
        y = model([t.to(DEV) for t in F.cfg5_full_inputs()])
    assert_close(y.cpu(), gold["logits"], rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg5 logits (full size, b=1)")


def test_cfg5_oracle_spot_check(hn):
    """The oracle route at this config stays exercised: one sample, volume cut to 2 x 224 x 224 (0.4 GB of scores per block)."""
    torch.manual_seed(53)
    model = hn.HealNet(**CFG5).eval().to(DEV)
    gen = torch.Generator().manual_seed(4107)
    ins = [torch.rand(1, 1, 2000, generator=gen), torch.rand(1, 4096, 768, generator=gen),
           torch.rand(1, 4096, 768, generator=gen), torch.rand(1, 2, 224, 224, 3, generator=gen)]
    with torch.no_grad():
        y = model([t.to(DEV) for t in ins])
    assert_close(y.cpu(), _oracle_logits(model, CFG5, ins), rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg5 logits (volume 2x224x224, b=1) vs oracle")


def test_cfg3_b16_bf16_vs_reference(hn):
    """BASELINE configs[2] (b = 16, bf16 tensors, bf16 MFMA core): sample 0 against the fp32 REFERENCE on the same
    bf16-rounded inputs (g10_cfg3_b16_s0), tolerance 2e-2 max-norm -- not against this build's own fp32 core."""
    want = _golden("g10_cfg3_b16_s0")["logits"]
    torch.manual_seed(0)
    kw = F.CFG3
    low = hn.HealNet(**kw, core_precision="bf16").eval().to(DEV)
    tab, img, vol = F.cfg3_inputs(16)
    with torch.no_grad():
        y = low([tab.to(DEV), img.to(DEV), vol.to(DEV)])
    assert rel_err(y[:1].cpu(), want) <= 2e-2, rel_err(y[:1].cpu(), want)
    # the fp32 core on the same inputs must sit (much) closer to the reference than the bf16 tolerance
    ref = hn.HealNet(**kw).eval().to(DEV)
    ref.load_state_dict(low.state_dict())
    with torch.no_grad():
        y32 = ref([tab[:1].to(DEV), img[:1].to(DEV), vol[:1].to(DEV)])
    assert_close(y32.cpu(), want, rel=TOL, floor=0.0, abs_floor=1e-5, what="cfg3 fp32 core vs the reference (bf16 tensors)")
