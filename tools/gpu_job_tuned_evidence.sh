#!/bin/bash
# evidence for the staged route: tuned-config table (staged and HN_NO_STAGING=1), per-kernel stats of a blca / kirp training step,
# the dropout cost table
tag=${1:-r03_zc}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
python tools/bench_tuned.py --json $out/${tag}_tuned_configs_b8.json 2>&1 | grep config
HN_NO_STAGING=1 python tools/bench_tuned.py --json $out/${tag}_tuned_configs_b8_generic_route.json 2>&1 | grep config
python tools/bench_dropout.py 2>&1 | grep -v Warn | tee $out/${tag}_dropout_cost.txt
python tools/host_profile_tuned.py 2>&1 | grep "host-only" | tee $out/${tag}_tuned_host_time.txt
cd /tmp; export TMPDIR=/tmp
for c in blca kirp; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/$c -o t -- python $GRAFT_REPO_ROOT/tools/bench_tuned.py --configs $c > $GRAFT_REPO_ROOT/$out/$c.log 2>&1
  cp $GRAFT_REPO_ROOT/$out/$c/t_kernel_stats.csv $GRAFT_REPO_ROOT/$out/${tag}_tuned_${c}_b8_kernel_stats.csv
done
find $GRAFT_REPO_ROOT/$out -name '*kernel_trace.csv' -delete
