#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03p; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for b in 1 4; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/b$b -o fw -- python $GRAFT_REPO_ROOT/tools/quick_cfg2.py $b 100 > $out/b$b.log 2>&1
done
