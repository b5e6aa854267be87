#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m2
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fullsize.py > gpurun_out/m2/tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/m2/tests.log | cut -c1-300
python tools/bench_tuned.py 2>&1 | grep config
