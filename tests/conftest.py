"""pytest configuration: the ``gpu`` marker gates everything that needs a real MI355X."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLD, name + ".npz")) as z:
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


def rel_err(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close(a, b, rel=1e-3, floor=1e-4, abs_floor=0.0, what=""):
    """Two checks: (i) max-norm relative error <= rel; (ii) elementwise |a-b| <= rel*|b| + floor*max|b| + abs_floor.
    BASELINE.md §3's logits criterion is rel=1e-3 with abs_floor=1e-5 (pass floor=0 for exactly that)."""
    a = a.double().cpu()
    b = b.double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    e = rel_err(a, b)
    assert e <= rel, f"{what}: max-norm rel err {e:.3e} > {rel:.1e}"
    scale = b.abs().max().clamp_min(1e-30)
    excess = ((a - b).abs() - (rel * b.abs() + floor * scale + abs_floor)).max()
    assert float(excess) <= 0, f"{what}: elementwise criterion violated by {float(excess):.3e} (scale {float(scale):.3e})"
