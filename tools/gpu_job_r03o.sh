#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_graph.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
for b in 1 2 4 8 12 16; do echo "b=$b cluster: $(python tools/quick_cfg2.py $b 100 2>/dev/null| tail -1)  | off: $(HN_NO_CHAIN_CLUSTER=1 python tools/quick_cfg2.py $b 100 2>/dev/null | tail -1)"; done
