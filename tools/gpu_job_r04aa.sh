#!/bin/bash
O=gpurun_out/r04aa; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_backward.py tests/test_gpu_staged.py tests/test_gpu_chain.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
timeout 300 python tools/bench_configs.py --cfg 3 4 5 --steps 10 2>/dev/null | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/fwd_cfg4 -o t -- python $R/tools/bench_configs.py --cfg 4 --core-precision fp32 --steps 20 > $R/$O/fwd_cfg4.log 2>&1
grep -E "merge_explicit|skinny|Name" $R/$O/fwd_cfg4/t_kernel_stats.csv | cut -c1-140
