#!/usr/bin/env python3
"""Per-kernel PMC averages for an arbitrary command: one rocprofv3 --kernel-trace --pmc pass per counter set (never combined with
other trace domains), each under its own timeout.

    python tools/pmc_kernels.py --match gemm_big gemm_tn --sets "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" -- python tools/bench_configs.py --cfg 4 --core-precision fp32 --steps 3
"""
import argparse, collections, csv, glob, json, os, shutil, subprocess, sys, tempfile

ap = argparse.ArgumentParser()
ap.add_argument("--match", nargs="+", required=True)
ap.add_argument("--sets", nargs="+", required=True)
ap.add_argument("--timeout", type=int, default=180)
ap.add_argument("cmd", nargs=argparse.REMAINDER)
args = ap.parse_args()
cmd = args.cmd[1:] if args.cmd and args.cmd[0] == "--" else args.cmd
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for i, cs in enumerate(args.sets):
    d = tempfile.mkdtemp(prefix="pmc_")
    try:
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + cs.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=args.timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except subprocess.TimeoutExpired:
        print(f"set {i} ({cs}) timed out", file=sys.stderr)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if any(m in n for m in args.match):
                agg[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if i == 0:
                    dur[n[:70]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    shutil.rmtree(d, ignore_errors=True)
out = {}
for n, c in agg.items():
    row = {k: sum(v) / len(v) for k, v in c.items()}
    row["launches"] = max(len(v) for v in c.values())
    if dur[n]:
        row["avg_ns_under_pmc"] = sum(dur[n]) / len(dur[n])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in row and row.get("GRBM_GUI_ACTIVE"):
        row["mfma_pipe_utilisation"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (row["GRBM_GUI_ACTIVE"] / 8.0)
    if "TCC_HIT_sum" in row and "TCC_REQ_sum" in row and row["TCC_REQ_sum"]:
        row["l2_hit_rate"] = row["TCC_HIT_sum"] / row["TCC_REQ_sum"]
    out[n] = row
print(json.dumps(out, indent=1))
