#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fullsize.py tests/test_gpu_staging.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
python tools/bench_configs.py --cfg 2 3 5 --core-precision bf16 --steps 30 2>/dev/null | grep -o '"cfg": [0-9]*\|"ms_per_forward": [0-9.]*' | paste - - | tr '\n' ' '
echo
python tools/roofline_configs.py --out gpurun_out/r03k --tag r03_k --cfg 3 --no-pmc > gpurun_out/r03k.log 2>&1; cut -c1-120 gpurun_out/r03k/r03_k_cfg3_b16_bf16_kernel_stats.csv | head -9
