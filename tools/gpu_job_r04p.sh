#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 400 tools/ubench/gemm_f32_bench 32768 1024 773 3 > $O/gemm.log 2>&1; echo "bench exit=$?"; grep -E "TN narrow M=(256|512)|TN variant [01]|RACE" $O/gemm.log
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_staged.py tests/test_gpu_dropout.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
timeout 120 python tools/bench_tuned.py > $O/tuned.log 2>&1; tail -5 $O/tuned.log
