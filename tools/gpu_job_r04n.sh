#!/bin/bash
O=gpurun_out/r04n; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 120 python tools/bench_tuned.py > $O/tuned_new.log 2>&1; tail -12 $O/tuned_new.log
HN_NO_GLDS_GEMM=1 timeout 120 python tools/bench_tuned.py > $O/tuned_old.log 2>&1; tail -12 $O/tuned_old.log
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_staged.py tests/test_gpu_dropout.py tests/test_gpu_reference_suite.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
