#!/bin/bash
O=gpurun_out/r04ac; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_context_split.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
timeout 300 python tools/bench_configs.py --cfg 3 5 --core-precision fp32 --steps 6 2>/dev/null | cut -c1-160
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/cfg5 -o t -- python $R/tools/bench_configs.py --cfg 5 --core-precision fp32 --steps 5 > $R/$O/cfg5.log 2>&1
grep -E "encode|Name" $R/$O/cfg5/t_kernel_stats.csv | cut -c1-150
