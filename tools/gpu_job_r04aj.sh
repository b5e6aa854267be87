#!/bin/bash
# round 4, call aj: cluster_stream_guard -- concurrent small-batch forwards on several streams; chain / small-batch / graph tests
O=gpurun_out/r04aj; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_streams.py tests/test_gpu_chain.py tests/test_gpu_graph.py tests/test_gpu_dist.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -12 $O/tests.log
timeout 120 python tools/two_stream.py 32 2>&1 | tail -3 | tee $O/r04_aj_two_stream.log
timeout 200 python tools/small_batch.py 2>/dev/null | tail -8 | tee $O/small_batch.log
