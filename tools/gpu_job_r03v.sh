#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_regressions.py tests/test_gpu_backward.py tests/test_gpu_dropout.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
python tools/bench_tuned.py --json gpurun_out/r03_v_tuned_configs_b8.json 2>/dev/null | cut -c1-200
