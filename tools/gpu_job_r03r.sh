#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_bchain.py tests/test_gpu_backward.py tests/test_gpu_dist.py tests/test_gpu_rccl.py tests/test_gpu_torchops.py tests/test_gpu_dropout.py tests/test_gpu_train.py tests/test_gpu_chain.py -q -m gpu -x 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
timeout 600 python tools/fuzz_forward.py --n 40 --seed 202 --scale chain --backward 2>&1 | tail -1
timeout 600 python tools/fuzz_forward.py --n 30 --seed 203 --backward 2>&1 | tail -1
timeout 600 python tools/fuzz_forward.py --n 20 --seed 204 --backward --dropout 2>&1 | tail -1
for c in cfg2 cfg4; do timeout 200 python tools/train_step.py --config $c --steps 30 2>/dev/null | tail -1; done
python tools/bench_configs.py --cfg 4 --core-precision fp32 --steps 50 2>/dev/null | cut -c1-200
