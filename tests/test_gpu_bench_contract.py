"""GPU: `bench.py` keeps its contract -- ONE JSON line on stdout with the keys the driver reads, the roofline object of the
dominant kernel and (without --no-cpu-baseline) the CPU baseline object."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, check=True).stdout
    lines = [l for l in out.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 32 * 1000.0 / j["ms_per_step"]) <= 1e-3 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert 0.3 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
