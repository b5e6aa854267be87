#!/bin/bash
# bf16 K/V projection of patch bags: parity tests, cfg4 / cfg5 timing with and without it, kernel stats of cfg5 with the bf16 core
# usage: bash tools/gpu_job_bf16proj.sh <tag>
tag=${1:-bf16proj}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_bf16proj.py tests/test_gpu_bf16.py -q -m gpu -x > $out/tests.log 2>&1; tail -3 $out/tests.log
timeout 300 python tools/bench_configs.py --cfg 4 5 --steps 20 --json $out/configs.json > $out/configs.log 2>&1; cat $out/configs.log
HN_NO_BF16_PROJ=1 timeout 300 python tools/bench_configs.py --cfg 4 5 --core-precision bf16 --steps 20 > $out/configs_noproj.log 2>&1; cat $out/configs_noproj.log
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o cfg5 -- python tools/bench_configs.py --cfg 5 --core-precision bf16 --steps 20 > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/cfg5_b4_bf16_kernel_stats.csv && head -8 $f | cut -c1-160
