#!/bin/bash
O=gpurun_out/r04x; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_context_split.py tests/test_gpu_dist.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -5 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
