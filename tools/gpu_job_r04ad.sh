#!/bin/bash
O=gpurun_out/r04ad; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for p in 0.0 0.25; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/d$p -o t -- python $R/tools/dropout_breakdown.py $p 0.0 > $R/$O/d$p.log 2>&1
tail -1 $R/$O/d$p.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/$O/d$p/t_kernel_stats.csv")))
for r in rows[:8]:
    print("%-90s calls %4s avg %8.1f us tot/step %8.1f"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3/13))
PY
done
