#!/bin/bash
O=gpurun_out/r04ab; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_context_split.py tests/test_gpu_staging.py tests/test_gpu_torchops.py tests/test_gpu_regressions.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
for i in 1 2; do timeout 300 python bench.py --steps 200 --no-cpu-baseline --no-staged-models --no-train-step 2>/dev/null | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/bench_cfg2 -o t -- python $R/bench.py --steps 20 --no-cpu-baseline --no-staged-models --no-train-step > $R/$O/bench_cfg2.log 2>&1
grep -E "encode_token|skinny|head_kernel|Name" $R/$O/bench_cfg2/t_kernel_stats.csv | cut -c1-150
