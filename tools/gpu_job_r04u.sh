#!/bin/bash
O=gpurun_out/r04u; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_train.py tests/test_gpu_staged.py tests/test_gpu_model.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
for i in 1 2 3; do timeout 200 python tools/train_step.py --config cfg4 --steps 40 2>/dev/null | tail -1 | cut -c60-140; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_cfg4 -o t -- python $R/tools/train_step.py --config cfg4 --steps 20 > $R/$O/train_cfg4.log 2>&1
grep -E "smallk|head_bwd|gemm_tn_kernel|Name" $R/$O/train_cfg4/t_kernel_stats.csv | cut -c1-160
