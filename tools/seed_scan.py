#!/usr/bin/env python3
"""Scan input seeds of the attention-backward parity cases (the test derived its seed from hash(), i.e. per process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_backward as T
from healnet_amd import _capi
params = T.ATTN_CASES
cases = [int(c) for c in sys.argv[1].split(",")]
lo, hi = int(sys.argv[2]), int(sys.argv[3])
for ci in cases:
    bad = []
    for seed in range(lo, hi):
        try:
            T._attn_case(_capi, seed=seed, _depth=99, **params[ci])      # _depth=99: no seed advance, scan the raw seed
        except AssertionError as e:
            bad.append((seed, str(e)[:60]))
    print("case", ci, "failing seeds", len(bad), "of", hi - lo, bad[:10], flush=True)
