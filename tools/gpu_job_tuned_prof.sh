#!/bin/bash
# kernel trace of the reference's tuned TCGA shapes (tools/bench_tuned.py), one rocprofv3 run per config
cd $GRAFT_REPO_ROOT
out=gpurun_out/tuned; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for c in blca kirp ucec; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/$c -o t -- python $GRAFT_REPO_ROOT/tools/bench_tuned.py --configs $c > $GRAFT_REPO_ROOT/$out/$c.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/$out/$c.log
done
cd $GRAFT_REPO_ROOT
find $out -name '*.db' -delete
