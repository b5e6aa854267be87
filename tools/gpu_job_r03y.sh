#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_bchain.py tests/test_gpu_backward.py tests/test_gpu_dist.py tests/test_gpu_rccl.py tests/test_gpu_train.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "cfg4" 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head
timeout 600 python tools/fuzz_forward.py --n 40 --seed 302 --scale chain --backward 2>&1 | tail -1
timeout 600 python tools/fuzz_forward.py --n 8 --seed 303 --scale medium --backward 2>&1 | tail -1
for i in 1 2; do python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1; HN_NO_CHAIN_CLUSTER=1 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1; done
