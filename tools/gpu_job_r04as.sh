#!/bin/bash
# round 4, call as: latent self-attention backward -- dQ and dK/dV bodies side by side in one launch; training tests, step timings A/B
O=gpurun_out/r04as; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_train.py tests/test_gpu_chain.py tests/test_gpu_graph.py tests/test_gpu_dist.py tests/test_gpu_streams.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -4 $O/tests.log
for i in 1 2; do
  timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-170
  HN_NO_SELF_BWD_PAIR=1 timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-170
  timeout 200 python tools/train_step.py --config cfg2 --steps 20 2>/dev/null | tail -1 | cut -c1-170
  HN_NO_SELF_BWD_PAIR=1 timeout 200 python tools/train_step.py --config cfg2 --steps 20 2>/dev/null | tail -1 | cut -c1-170
done | tee $O/r04_as_self_pair_ab.log
