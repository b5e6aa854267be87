#!/usr/bin/env python3
"""PMC passes of a forward on the GPU box and their per-kernel summary (profiles/rNN_*_pmc_cfg2_b32.json; with --cfg the other
BASELINE configs: profiles/rNN_*_pmc_cfg3_b16_bf16.json ...).

    python tools/pmc_collect.py --out gpurun_out/pmc --json gpurun_out/r01_pmc_cfg2_b32.json [--core-precision fp32]
    python tools/pmc_collect.py --cfg 3 --core-precision bf16 --out gpurun_out/pmc3 --json gpurun_out/pmc_cfg3_b16_bf16.json

Runs three separate `rocprofv3 --kernel-trace --pmc ...` passes (FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM_GUI_ACTIVE: one
counter set per pass, never combined with a sys/hip/hsa trace) over `tools/quick_cfg2.py 32 3`, then averages the
counters per kernel name.  HBM traffic per launch follows MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB per
dispatch; on gfx950 FETCH_SIZE reads half of the bytes of wide coalesced streams, so the upper estimate is
(2*FETCH + WRITE) KiB and the raw figure (FETCH + WRITE) KiB is reported beside it."""
import argparse, collections, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "sq": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU "
          "SQ_INSTS_VALU GRBM_GUI_ACTIVE",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/pmc")
    ap.add_argument("--json", default="gpurun_out/pmc_cfg2_b32.json")
    ap.add_argument("--core-precision", default="fp32")
    ap.add_argument("--cfg", type=int, default=2, help="BASELINE config (2: tools/quick_cfg2.py 32 3; 3 / 4 / 5: tools/bench_configs.py at its batch, 3 steps)")
    ap.add_argument("--train", action="store_true", help="the cfg4 training step instead of a forward (tools/train_step.py --steps 3 --warmup 2)")
    args = ap.parse_args()
    if args.train:
        args.cfg = 4
        workload = [sys.executable, os.path.join(ROOT, "tools/train_step.py"), "--config", "cfg4", "--steps", "3", "--warmup", "2"]
    elif args.cfg == 2:
        workload = [sys.executable, os.path.join(ROOT, "tools/quick_cfg2.py"), "32", "3"]
    else:
        workload = [sys.executable, os.path.join(ROOT, "tools/bench_configs.py"), "--cfg", str(args.cfg), "--core-precision", args.core_precision,
                    "--steps", "3"]
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", HN_QUICK_PRECISION=args.core_precision)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for tag, counters in PASSES.items():
        d = os.path.join(out, tag)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "-o", tag, "--"] + workload
        subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        assert files, f"no counter_collection.csv under {d}"
        for f in files:
            for r in csv.DictReader(open(f)):
                per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if tag == "sq" and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    durations[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    kernels = {}
    for name, cs in per.items():
        if not name.startswith(("hn::", "void hn::")):
            continue
        row = {"launches_sampled": max(len(v) for v in cs.values())}
        for c, v in cs.items():
            row[c + "_avg"] = sum(v) / len(v)
        if durations[name]:
            row["avg_duration_ns_under_pmc"] = sum(durations[name]) / len(durations[name])
        if "SQ_VALU_MFMA_BUSY_CYCLES_avg" in row and row.get("GRBM_GUI_ACTIVE_avg"):
            # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
            row["mfma_pipe_utilisation"] = row["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / 1024.0 / (row["GRBM_GUI_ACTIVE_avg"] / 8.0)
        if "FETCH_SIZE_avg" in row and "WRITE_SIZE_avg" in row:
            row["hbm_bytes_raw"] = (row["FETCH_SIZE_avg"] + row["WRITE_SIZE_avg"]) * 1024.0
            row["hbm_bytes_fetch_doubled"] = (2 * row["FETCH_SIZE_avg"] + row["WRITE_SIZE_avg"]) * 1024.0
        kernels[name] = row
    dom = max(kernels, key=lambda k: kernels[k].get("avg_duration_ns_under_pmc", 0.0) * kernels[k]["launches_sampled"])
    import hashlib
    def sha(rel):
        with open(os.path.join(ROOT, rel), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    doc = {
        "source_sha256": {"attention.hip": sha("healnet_amd/csrc/attention.hip")},     # bench.py drops the traffic figure when this no longer matches
        "git_head": os.environ.get("HN_GIT_HEAD"),
        "command": "python tools/pmc_collect.py (three rocprofv3 --kernel-trace --pmc passes over " + " ".join(os.path.relpath(w, ROOT) if os.path.isabs(w) and w.startswith(ROOT) else w for w in workload[1:]) + ": "
                   + " | ".join(PASSES.values()) + ")",
        "config": "cfg%d%s" % (args.cfg, " training step" if args.train else ""),
        "core_precision": args.core_precision,
        "note": "FETCH_SIZE/WRITE_SIZE are KiB per dispatch; gfx950: FETCH_SIZE reads 1/2 of the bytes of wide coalesced streams "
                "(MI355X_MICROARCH.md) -> upper estimate (2*FETCH + WRITE) KiB, raw (FETCH + WRITE) KiB",
        "dominant_kernel": dom,
        "dominant_kernel_traffic_bytes_per_launch": {"raw": kernels[dom].get("hbm_bytes_raw"),
                                                     "fetch_doubled": kernels[dom].get("hbm_bytes_fetch_doubled")},
        "algorithmic_bytes_per_launch": ({"z_context_read_once": 32 * 50176 * 16 * 4, "partials_written": 32 * 8 * 8 * 128 * 18 * 4} if args.cfg == 2 else None),
        "kernels": kernels,
    }
    with open(args.json, "w") as f:
        json.dump(doc, f, indent=1)
    k = kernels[dom]
    print(json.dumps({"dominant": dom[:60], "mfma_util": k.get("mfma_pipe_utilisation"), "hbm_raw_MB": (k.get("hbm_bytes_raw") or 0) / 1e6,
                      "hbm_doubled_MB": (k.get("hbm_bytes_fetch_doubled") or 0) / 1e6}))


if __name__ == "__main__":
    main()
