#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
timeout 120 python tools/bench_tuned.py > $O/tuned.log 2>&1; tail -5 $O/tuned.log
cd /tmp && export TMPDIR=/tmp
for c in kirp; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$c -o t -- python $R/tools/bench_tuned.py --configs $c > $R/$O/$c.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/$O/$c/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("$c total ms", tot/1e6)
for r in rows[:30]:
    print("%-70s calls %6s avg %8.1f us  %5.1f%%"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
done
