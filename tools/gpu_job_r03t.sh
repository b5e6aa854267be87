#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_graph.py tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
for b in 1 2 4 8 16 32; do echo "$(python tools/quick_cfg2.py $b 100 2>/dev/null| tail -1)"; done
