#!/bin/bash
# round 4, call ah: packed context under dropout -- dropout tests, staged tests, dropout cost
O=gpurun_out/r04ah; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dropout.py tests/test_gpu_staged.py tests/test_gpu_graph.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -5 $O/tests.log
timeout 300 python tools/bench_dropout.py 2>/dev/null | tee $O/r04_ah_dropout_cost.txt
timeout 600 python tools/fuzz_forward.py --scale staged --n 12 --seed 9 --backward --dropout 2>&1 | tail -1 | tee $O/fuzz.log
timeout 600 python tools/fuzz_forward.py --scale small --n 12 --seed 3 --backward --dropout 2>&1 | tail -1 | tee -a $O/fuzz.log
