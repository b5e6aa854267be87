#!/bin/bash
O=gpurun_out/r04ae; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -k "bf16" > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
timeout 300 python tools/bench_configs.py --cfg 4 5 --steps 10 2>/dev/null | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/cfg4 -o t -- python $R/tools/bench_configs.py --cfg 4 --core-precision bf16 --steps 20 > $R/$O/cfg4.log 2>&1
grep -E "gemm_bf16|Name" $R/$O/cfg4/t_kernel_stats.csv | cut -c1-170
