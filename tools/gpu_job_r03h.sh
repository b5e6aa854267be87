#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
for w16 in 2 3 4; do echo "w16=$w16: $(HN_BF16_WAVES16=$w16 python tools/bench_configs.py --cfg 2 3 5 --core-precision bf16 --steps 20 2>/dev/null | grep -o '"cfg": [0-9]*\|"ms_per_forward": [0-9.]*' | paste - - | tr '\n' ' ')"; done
