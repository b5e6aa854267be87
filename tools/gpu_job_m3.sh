#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m3
timeout 900 python -m pytest tests/test_gpu_staged.py -q -m gpu > gpurun_out/m3/staged.log 2>&1; echo "staged rc=$?"; tail -25 gpurun_out/m3/staged.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench_contract.py -q -m gpu -x > gpurun_out/m3/full.log 2>&1; echo "full rc=$?"; tail -3 gpurun_out/m3/full.log | cut -c1-250
cd /tmp; export TMPDIR=/tmp
for c in blca kirp; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/m3/$c -o t -- python $GRAFT_REPO_ROOT/tools/bench_tuned.py --configs $c > $GRAFT_REPO_ROOT/gpurun_out/m3/$c.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/m3/$c.log
done
