#!/bin/bash
# round 4, call c: TN kernel (inline-asm LDS-DMA, XCD maps, ablations), the reference-suite tests, cfg4 step time
O=gpurun_out/r04c; mkdir -p $O
timeout 300 tools/ubench/gemm_f32_bench > $O/gemm.log 2>&1; echo "exit=$?" >> $O/gemm.log
timeout 300 python tools/train_step.py --config cfg4 --steps 30 > $O/train_cfg4_new.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_reference_suite.py -x -q -m gpu -s > $O/tests_ref.log 2>&1; echo "exit=$?" >> $O/tests_ref.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_backward.py -x -q -m gpu -k "cfg4 or backward" > $O/tests.log 2>&1; echo "exit=$?" >> $O/tests.log
tail -12 $O/gemm.log; tail -n 3 $O/train_cfg4_new.log; tail -n 15 $O/tests_ref.log; tail -n 3 $O/tests.log
