#!/bin/bash
# round 4, call ap: LN(x) of every attention block on the tape (written by the chain that projects for it) -- training tests, step timings
O=gpurun_out/r04ap; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_train.py tests/test_gpu_staged.py tests/test_gpu_graph.py tests/test_gpu_dropout.py tests/test_gpu_dist.py tests/test_gpu_reference_suite.py tests/test_gpu_chain.py tests/test_gpu_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -4 $O/tests.log
for i in 1 2; do
  timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-170
  HN_NO_XHAT_TAPE=1 timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-170
done | tee $O/r04_ap_ztape_ab.log
timeout 200 python tools/bench_tuned.py 2>/dev/null | tee $O/r04_ap_tuned.log
HN_NO_XHAT_TAPE=1 timeout 200 python tools/bench_tuned.py 2>/dev/null | tee -a $O/r04_ap_tuned.log
