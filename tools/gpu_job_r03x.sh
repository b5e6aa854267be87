#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_staging.py tests/test_gpu_regressions.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "cfg4 or cfg5" 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head
python tools/bench_configs.py --cfg 4 --core-precision fp32 --steps 50 2>/dev/null | cut -c1-160
python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1
python tools/bench_tuned.py --configs blca ucec 2>/dev/null | cut -c1-160
