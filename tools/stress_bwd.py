#!/usr/bin/env python3
"""Repeat the attention-backward parity tests in one process and report how close every comparison comes to its tolerance
(flakiness hunt): python tools/stress_bwd.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest
import test_gpu_backward as T
from healnet_amd import _capi
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worst = {}
orig = conftest.assert_close


def rec(a, b, rel=1e-3, floor=1e-4, abs_floor=0.0, what=""):
    a64, b64 = a.double().cpu(), b.double().cpu()
    e = conftest.rel_err(a64, b64) / rel
    scale = b64.abs().max().clamp_min(1e-30)
    ex = float(((a64 - b64).abs() / (rel * b64.abs() + floor * scale + abs_floor + 1e-300)).max())
    key = (cur[0], what)
    worst[key] = max(worst.get(key, 0.0), e, ex)
    return orig(a, b, rel, floor, abs_floor, what)


T.assert_close = rec
params = [m.args[1] for m in T.test_attention_backward.pytestmark if m.name == "parametrize"][0]
cur = [0]
fails = 0
for it in range(reps):
    for i, kw in enumerate(params):
        cur[0] = i
        try:
            T.test_attention_backward(_capi, dict(kw))
        except AssertionError as e:
            fails += 1
            print("FAIL iter", it, "case", i, str(e)[:300].replace("\n", " "), flush=True)
print("done", reps, "reps,", fails, "failures; worst fraction of tolerance used per (case, tensor):")
for k, v in sorted(worst.items(), key=lambda x: -x[1])[:12]:
    print("  ", k, round(v, 3))
