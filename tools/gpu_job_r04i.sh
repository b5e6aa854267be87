#!/bin/bash
# round 4, call i: dQ on the LDS ring with a two-wave split plan, TN clamp in 32 bits; A/B in one run
O=gpurun_out/r04i; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 300 tools/ubench/gemm_f32_bench 32768 1024 773 3 > $O/gemm.log 2>&1; grep -E "variant [01] \[|TN variant [01]|RACE|differ" $O/gemm.log
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_reference_suite.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -2 $O/tests.log
for i in 1 2 3; do
timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-120
HN_NO_ATTN_LDS=1 timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-120
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_cfg4 -o t -- python $R/tools/train_step.py --config cfg4 --steps 20 > $R/$O/train_cfg4.log 2>&1
head -8 $R/$O/train_cfg4/t_kernel_stats.csv | cut -c1-140
