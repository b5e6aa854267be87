#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m5
timeout 900 python -m pytest tests/test_gpu_staged.py tests/test_gpu_regressions.py tests/test_gpu_backward.py tests/test_gpu_ops.py -q -m gpu -x > gpurun_out/m5/tests.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/m5/tests.log | cut -c1-250
python tools/bench_tuned.py 2>&1 | grep config
python tools/bench_tuned.py --configs blca brca 2>&1 | grep config
HN_NO_TALL_NARROW=1 python tools/bench_tuned.py --configs blca brca 2>&1 | grep config
cd /tmp; export TMPDIR=/tmp
for c in blca; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/m5/$c -o t -- python $GRAFT_REPO_ROOT/tools/bench_tuned.py --configs $c > $GRAFT_REPO_ROOT/gpurun_out/m5/$c.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/m5/$c.log
done
