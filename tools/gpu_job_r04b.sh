#!/bin/bash
# round 4, call b: NT / TN A/B bench, cfg4 parity tests on the new kernels, cfg4 step time with and without them
O=gpurun_out/r04b; mkdir -p $O
timeout 300 tools/ubench/gemm_f32_bench > $O/gemm.log 2>&1; echo "exit=$?" >> $O/gemm.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_backward.py tests/test_gpu_ops.py tests/test_gpu_bf16proj.py tests/test_gpu_regressions.py -x -q -m gpu > $O/tests.log 2>&1; echo "exit=$?" >> $O/tests.log
timeout 300 python tools/train_step.py --config cfg4 --steps 30 > $O/train_cfg4_new.log 2>&1
HN_NO_GLDS_GEMM=1 timeout 300 python tools/train_step.py --config cfg4 --steps 30 > $O/train_cfg4_old.log 2>&1
timeout 300 python tools/bench_configs.py --cfg 4 5 --core-precision fp32 --steps 10 > $O/fwd_new.log 2>&1
HN_NO_GLDS_GEMM=1 timeout 300 python tools/bench_configs.py --cfg 4 5 --core-precision fp32 --steps 10 > $O/fwd_old.log 2>&1
tail -8 $O/gemm.log; tail -5 $O/tests.log; tail -2 $O/train_cfg4_new.log $O/train_cfg4_old.log $O/fwd_new.log $O/fwd_old.log
