#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_graph.py tests/test_gpu_model.py tests/test_gpu_fuzz.py tests/test_gpu_regressions.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
python tools/small_batch.py --n 300 2>&1 | grep '"case": "cfg1"' | cut -c1-200
HN_NO_CHAIN_CLUSTER=1 python tools/small_batch.py --n 300 --batches 1 4 2>&1 | grep '"case": "cfg1"' | cut -c1-130
for b in 1 2 4 8 12 16 32; do echo "b=$b cluster: $(python tools/quick_cfg2.py $b 100 | tail -1)  | off: $(HN_NO_CHAIN_CLUSTER=1 python tools/quick_cfg2.py $b 100 | tail -1)"; done
