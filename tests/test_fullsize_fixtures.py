"""CPU: the compressed gradient fixtures of tests/test_gpu_fullsize.py (tests/fullsize_fixtures.py) -- the count sketch is linear and
its squared norm estimates the squared L2 distance; the committed g10_* fixtures hold every key the GPU tests read."""
import os

import numpy as np
import torch

import fullsize_fixtures as F
from conftest import GOLD


def test_sketch_is_linear_and_estimates_the_l2_distance():
    gen = torch.Generator().manual_seed(5)
    errs = []
    for trial, n in enumerate((1000, 131072, 2049000)):
        a = torch.randn(n, generator=gen).double()
        d = torch.randn(n, generator=gen).double() * 1e-3
        key = f"layers.{trial}.w"
        sa, sb = F.sketch(a, key), F.sketch(a + d, key)
        assert torch.allclose(sb - sa, F.sketch(d.double(), key), atol=1e-9)
        est = float((sb - sa).pow(2).sum().sqrt())
        errs.append(abs(est / float(d.double().norm()) - 1.0))
    assert max(errs) < 0.25, errs            # relative standard deviation of the estimate ~ sqrt(2 / 128) = 0.125
    # an isolated flip (one row of a weight gradient) and a systematic error (a dropped 1/8 of the sum) are told apart by size
    g = torch.randn(1024 * 128, generator=gen)
    sys_err = g * 0.125
    assert float(F.sketch(sys_err, "k").pow(2).sum().sqrt()) / float(g.norm()) > 0.05


def test_sample_index_is_seeded_and_sorted():
    i1, i2 = F.sample_index("a.b", 500000), F.sample_index("a.b", 500000)
    assert torch.equal(i1, i2) and i1.numel() == F.SAMPLE and bool((i1[1:] > i1[:-1]).all())
    assert torch.equal(F.sample_index("x", 100), torch.arange(100))


def test_committed_fullsize_fixtures_are_complete():
    with np.load(os.path.join(GOLD, "g10_cfg2_b32_bench.npz")) as z:
        assert z["logits"].shape == (32, 4) and np.isfinite(z["logits"]).all()
    with np.load(os.path.join(GOLD, "g10_cfg2_b32_train.npz")) as z:
        keys = [str(k) for k in z["keys"]]
        assert len(keys) == 125 and z["logits"].shape == (32, 4)
        for k in keys:
            assert z[f"{k}::sketch"].shape == (F.SKETCH_BUCKETS,) and z[f"{k}::norm"].shape == (1,) and z[f"{k}::scale"].shape == (1,)
            assert 1 <= z[f"{k}::vals"].shape[0] <= F.SAMPLE
    with np.load(os.path.join(GOLD, "g10_cfg5_cut_b2.npz")) as z:
        assert z["logits"].shape == (2, 4) and z["logits_bag2_missing"].shape == (2, 4)
    with np.load(os.path.join(GOLD, "g10_cfg5_full_b1.npz")) as z:
        assert z["logits"].shape == (1, 4)
    with np.load(os.path.join(GOLD, "g10_cfg3_b16_s0.npz")) as z:
        assert z["logits"].shape == (1, 4)
