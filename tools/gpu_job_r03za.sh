#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03za
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_chain.py tests/test_gpu_bchain.py tests/test_gpu_rccl.py -q -m gpu -x --durations=5 > gpurun_out/r03za/tests.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r03za/tests.log
bash tools/gpu_job_tuned_prof.sh
