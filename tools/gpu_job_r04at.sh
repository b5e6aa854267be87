#!/bin/bash
# round 4, call at: training forward -- the one-token block's broadcast add + feed-forward + next projections on one chain (head 2); tests, A/B
O=gpurun_out/r04at; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_train.py tests/test_gpu_chain.py tests/test_gpu_graph.py tests/test_gpu_dist.py tests/test_gpu_dropout.py tests/test_gpu_torchops.py tests/test_gpu_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -4 $O/tests.log
for i in 1 2; do
  timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-170
  HN_NO_TAB_CHAIN=1 timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-170
  timeout 200 python tools/train_step.py --config cfg2 --steps 20 2>/dev/null | tail -1 | cut -c1-170
  HN_NO_TAB_CHAIN=1 timeout 200 python tools/train_step.py --config cfg2 --steps 20 2>/dev/null | tail -1 | cut -c1-170
done | tee $O/r04_at_tab_chain_ab.log
timeout 300 python tools/fuzz_forward.py --scale small --n 16 --seed 21 --backward 2>&1 | tail -1
