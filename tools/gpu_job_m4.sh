#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m4
timeout 900 python -m pytest tests/test_gpu_staged.py tests/test_gpu_regressions.py tests/test_gpu_backward.py tests/test_gpu_bchain.py tests/test_gpu_train.py tests/test_gpu_ops.py -q -m gpu -x > gpurun_out/m4/tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/m4/tests.log | cut -c1-250
python tools/bench_tuned.py 2>&1 | grep config
for c in cfg2 cfg4; do timeout 200 python tools/train_step.py --config $c --steps 30 2>/dev/null | tail -1; done
python tools/bench_dropout.py 2>&1 | grep -v Warn
cd /tmp; export TMPDIR=/tmp
for c in blca; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/m4/$c -o t -- python $GRAFT_REPO_ROOT/tools/bench_tuned.py --configs $c > $GRAFT_REPO_ROOT/gpurun_out/m4/$c.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/m4/$c.log
done
