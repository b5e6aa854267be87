#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03w; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/blca -o t -- python $GRAFT_REPO_ROOT/tools/bench_tuned.py --configs blca > $out/blca.log 2>&1
