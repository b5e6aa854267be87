#!/usr/bin/env python3
"""Per-position view of the LAST training step in a rocprofv3 kernel trace of tools/train_step.py: every launch in order with its
duration and the gap to the previous launch's end; then the step's time by kernel name.
    python tools/trace_step.py <dir with *kernel_trace.csv> [marker-kernel-substring = first kernel of a step]"""
import collections, csv, glob, os, re, sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "FillFunctor"
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a step = from one l1_adam_kernel's end to the next one's end
ends = [i for i, r in enumerate(rows) if "l1_adam_kernel" in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1]
t0, prev = int(rows[lo]["Start_Timestamp"]), None
by = collections.OrderedDict()
gaps = 0.0
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:64]
    gap = 0.0 if prev is None else (s - prev) / 1e3
    gaps += max(gap, 0.0)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us gap {gap:6.1f}  {name}")
    prev = e
    k = by.setdefault(name, [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
print(f"--- step: span {(prev - t0) / 1e3:.1f} us, {hi - lo + 1} launches, gaps {gaps:.1f} us")
for name, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us  {n:4d} x {t / n:7.1f}  {name}")
