#!/bin/bash
O=gpurun_out/r04v; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_context_split.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -25 $O/tests.log
