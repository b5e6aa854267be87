#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m7
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fullsize.py > gpurun_out/m7/tests.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/m7/tests.log | cut -c1-250
python tools/host_profile_tuned.py 2>&1 | grep "host-only"
python tools/bench_tuned.py 2>&1 | grep config
for c in cfg2 cfg4; do timeout 200 python tools/train_step.py --config $c --steps 30 2>/dev/null | tail -1; done
python tools/small_batch.py --n 200 2>&1 | grep '"case": "cfg1"' | cut -c1-200
