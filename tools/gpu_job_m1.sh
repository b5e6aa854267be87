#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m1
timeout 900 python -m pytest tests/test_gpu_dropout.py tests/test_gpu_chain.py tests/test_gpu_bchain.py tests/test_gpu_train.py tests/test_gpu_regressions.py tests/test_gpu_model.py -q -m gpu -x > gpurun_out/m1/tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/m1/tests.log
python tools/bench_dropout.py 2>&1 | grep -v Warn
python tools/quick_cfg2.py 32 30 2>&1 | tail -2
for c in cfg2 cfg4; do timeout 200 python tools/train_step.py --config $c --steps 30 2>/dev/null | tail -1; done
python tools/bench_tuned.py 2>&1 | grep config
