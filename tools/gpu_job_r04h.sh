#!/bin/bash
# round 4, call h: explicit attention cores on the LDS ring (forward + dQ): parity tests, A/B timing
O=gpurun_out/r04h; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_reference_suite.py tests/test_gpu_bchain.py tests/test_gpu_regressions.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -4 $O/tests.log
for i in 1 2; do
timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-150
HN_NO_ATTN_LDS=1 timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-150
done
timeout 300 python tools/bench_configs.py --cfg 4 5 --core-precision fp32 --steps 20 2>/dev/null | cut -c1-160
HN_NO_ATTN_LDS=1 timeout 300 python tools/bench_configs.py --cfg 4 5 --core-precision fp32 --steps 20 2>/dev/null | cut -c1-160
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_cfg4 -o t -- python $R/tools/train_step.py --config cfg4 --steps 20 > $R/$O/train_cfg4.log 2>&1
head -9 $R/$O/train_cfg4/t_kernel_stats.csv | cut -c1-140
