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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
                          "--train-steps", "3"], cwd=ROOT, capture_output=True, text=True, timeout=600, check=True).stdout
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
    assert 0.3 < r["frac_useful"] < r["frac"]
    assert r["traffic"] is None or r["traffic_source"]["source"].startswith("profiles/")
    t = j["train_step"]                                   # fwd + bwd + all-reduce + optimizer step at cfg4, b=8 / GPU
    assert t["batch_per_gpu"] == 8 and t["world_size"] == 1 and t["unit"] == "samples/s" and t["steps"] == 3
    assert abs(t["value"] - 8 * 1000.0 / t["ms_per_step"]) <= 1e-3 * t["value"] and t["final_loss"] == t["final_loss"]
    assert sum(t["allreduce_buckets_floats"]) == t["gradient_floats"]
    tr = t["roofline"]                                    # the step's rate of fp32 results + its two dominant kernels, timed in the same run
    # (the two patch-bag GEMMs run fp32-exact on the bf16 pipe -- gemm_x6.hip -- so `frac` is a rate against the fp32 MFMA peak, not a
    # utilisation of that pipe; the kernels themselves are priced against the bf16 peak with the 6 x FLOPs they execute)
    assert tr["bound"] == "mfma" and tr["peak"] == 157.3 and 0.2 < tr["frac"] <= 1.5
    assert abs(tr["frac"] - tr["flops_executed_per_step"] / (t["ms_per_step"] * 1e-3) / 1e12 / 157.3) < 2e-3
    assert tr["patch_bag_gemm_route"].startswith("x6")
    for name in ("gemm_nt_x6", "gemm_tn_x6"):
        k = tr["dominant_kernels"][name]
        assert k["launches_timed"] == 3 * t["steps"] and 0.3 < k["frac"] <= 1.0 and k["peak"] == 2500.0, (name, k)
        assert abs(k["frac"] - k["bf16_flops_executed_per_launch"] / (k["avg_launch_ms"] * 1e-3) / 1e12 / 2500.0) < 2e-3
        assert k["bf16_flops_executed_per_launch"] >= 6 * k["fp32_flops_returned_per_launch"]
        assert k["fp32_equivalent_tflops"] > 157.3, (name, k)      # what the route is for: more fp32 results per second than the fp32 pipe's peak
    assert len(j["build_id"]) == 16
    sm = j["staged_models"]                               # the reference's tuned TCGA shapes, run as zero-padded images
    assert set(sm["configs"]) == {"blca", "brca", "kirp", "ucec"} and sm["unit"] == "ms"
    assert all(c["staged"] and 0.1 < c["fwd_bwd_ms"] < 50.0 for c in sm["configs"].values())
    pb = j["patch_bag_precisions"]                        # BASELINE configs[3]'s shape: fp32 vs the bf16 projection + core
    assert 0.1 < pb["bf16_ms"] < pb["fp32_ms"] < 50.0 and 0.0 < pb["bf16_maxnorm_diff_vs_fp32"] < 2e-2
    cf = j["configs"]                                     # the other BASELINE configs at HEAD (round 5): forward + dominant kernel vs its roof
    assert set(cf) == {"cfg1", "cfg3", "cfg4", "cfg5"}
    for name, c in cf.items():
        k = c["dominant_kernel"]
        assert abs(c["samples_per_s"] - c["batch"] * 1000.0 / c["ms_per_forward"]) <= 2e-3 * c["samples_per_s"], name
        assert k["launches_timed"] == k["launches_per_forward"] * c["steps"], (name, k)
        assert 0.2 < k["frac"] <= 1.0 and abs(k["frac"] - k["work_per_launch"] / (k["avg_launch_ms"] * 1e-3) / k["peak"]) < 2e-3, (name, k)
        assert 0.1 < k["share_of_forward"] <= 1.0, (name, k)
    assert cf["cfg3"]["dominant_kernel"]["bound"].startswith("valu") and cf["cfg4"]["dominant_kernel"]["bound"].startswith("mfma")
    cl = t["cluster"]                                     # the cluster-mode failure signal of the step (include/healnet_hip.h, ABI v10)
    assert cl["lost"] == 0 and cl["fallbacks"] == 0 and cl["enabled"] and cl["optimizer_steps_skipped"] == 0


def test_plain_gpus_n_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment (the shape of the driver's command) must start 2 ranks itself.
    On this 1-GPU box both ranks share cuda:0 and talk over gloo (HN_BENCH_SHARED_GPU=1, a test-only switch); the launcher,
    the barrier / max-over-ranks timing and the 2-rank training step with the overlapped all-reduce are the real code."""
    env = dict(os.environ, HN_BENCH_SHARED_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--train-steps", "3"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 64 and "cpu_baseline" not in j
    assert abs(j["value"] - 64 * 1000.0 / j["ms_per_step"]) <= 1e-3 * j["value"]
    t = j["train_step"]
    assert t["world_size"] == 2 and t["global_batch"] == 16 and t["backend"] == "gloo"
    assert "ms_per_step_blocking_allreduce" in t and "allreduce_alone_ms" in t
    # N > 1 dry run: the overlapped all-reduce released every bucket exactly once -- the float ranges add up to
    # the flat gradient buffer (each parameter rounded up to 4 floats) -- the per-rank batch is the config's, the step really ran,
    # and the records only rank 0 of a single-GPU run produces are absent (the driver's N = 2 / 4 / 8 lines stay short)
    assert sum(t["allreduce_buckets_floats"]) == t["gradient_floats"] and len(t["allreduce_buckets_floats"]) >= 2
    assert t["batch_per_gpu"] == 8 and t["steps"] == 3 and t["final_loss"] == t["final_loss"] and "error" not in t
    assert abs(t["value"] - 16 * 1000.0 / t["ms_per_step"]) <= 1e-3 * t["value"]
    for rank0_only in ("cpu_baseline", "configs", "staged_models", "patch_bag_precisions"):
        assert rank0_only not in j, rank0_only
    # data-parallel ranks run without cluster launches (healnet_amd.dist.keep_ranks_in_step: a lost exchange on one rank would
    # poison every peer through the all-reduce): the step's cluster record says so, and nothing was lost or skipped
    cl = t["cluster"]
    assert not cl["enabled"] and cl["lost"] == 0 and cl["optimizer_steps_skipped"] == 0


@pytest.mark.parametrize("inject", ["raise:1", "hang:1", "raise:0"])
def test_a_failing_training_record_at_n2_still_prints_the_headline_line(inject):
    """N > 1: the training-step record runs collectives; a rank that raises in it, or never returns from it, must not cost the
    driver the line (bench.py main(): guarded records + a watchdog armed on every rank).  One JSON line, the headline intact,
    `train_step.error` says what happened, exit code 0."""
    env = dict(os.environ, HN_BENCH_SHARED_GPU="1", HN_BENCH_INJECT=inject)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--train-steps", "3", "--train-watchdog-s", "10"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["roofline"]["frac"] > 0.2 and j["build_id"]
    assert "error" in j["train_step"], j["train_step"]
