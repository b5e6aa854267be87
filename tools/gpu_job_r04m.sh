#!/bin/bash
O=gpurun_out/r04m; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 300 tools/ubench/gemm_f32_bench 32768 1024 773 3 > $O/gemm.log 2>&1; grep -E "TN variant [01]|TN glds|TN 1024|RACE" $O/gemm.log
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_torchops.py tests/test_gpu_graph.py tests/test_gpu_dist.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -2 $O/tests.log
for i in 1 2; do timeout 200 python tools/train_step.py --config cfg4 --steps 40 2>/dev/null | tail -1 | cut -c60-130; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_cfg4 -o t -- python $R/tools/train_step.py --config cfg4 --steps 20 > $R/$O/train_cfg4.log 2>&1
grep -E "Fill|tn_reduce|Name" $R/$O/train_cfg4/t_kernel_stats.csv | cut -c1-150
