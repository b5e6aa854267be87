#!/usr/bin/env python3
"""Per-position view of a rocprofv3 kernel trace: the launches of the LAST forward in a `tools/quick_cfg2.py` run, in order, with
duration and the gap to the previous launch's end (us).   python tools/trace_forward.py <dir with *kernel_trace.csv> [first-kernel-substring]"""
import csv, glob, os, re, sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "prelude_kernel"      # first launch of a forward ("encode" for routes without the merged prelude)
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# a forward begins at the first launch whose name holds `first` after a launch that does not
begins = [i for i in starts if i == 0 or first not in rows[i - 1]["Kernel_Name"]]
lo = begins[-1]
prev_end, t0, total = None, int(rows[lo]["Start_Timestamp"]), 0.0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size', '?'):>8}  {name}")
    prev_end = e
    total += (e - s) / 1e3
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernel time {total:.1f} us, {len(rows) - lo} launches")
