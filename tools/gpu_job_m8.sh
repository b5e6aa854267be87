#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m8
timeout 900 python -m pytest tests/test_gpu_dropout.py tests/test_gpu_staged.py tests/test_gpu_fuzz.py -q -m gpu -x > gpurun_out/m8/tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/m8/tests.log | cut -c1-250
timeout 600 python tools/fuzz_forward.py --n 40 --seed 107 --backward --dropout 2>&1 | tail -1
timeout 600 python tools/fuzz_forward.py --n 30 --seed 108 --scale chain --backward --dropout 2>&1 | tail -1
python tools/bench_dropout.py 2>&1 | grep -v Warn
