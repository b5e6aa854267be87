#!/bin/bash
# round 4, call f: dK/dV LDS kernel with progressive waits (A/B), PMC of the training-step kernels
O=gpurun_out/r04f; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_bchain.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -2 $O/tests.log
for i in 1 2; do
timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 > $O/train_new_$i.log
HN_NO_DKV_LDS=1 timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 > $O/train_nodkvlds_$i.log
done
cat $O/train_new_*.log $O/train_nodkvlds_*.log | cut -c1-200
timeout 900 python tools/pmc_kernels.py --match gemm_nt_glds gemm_tn_glds attn_bwd_dkv attn_bwd_dq "attn_core_kernel<4" --timeout 240 --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" -- python $R/tools/train_step.py --config cfg4 --steps 4 --warmup 2 > $O/r04_f_pmc_train_cfg4.json 2> $O/pmc.err; head -c 3000 $O/r04_f_pmc_train_cfg4.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_cfg4 -o t -- python $R/tools/train_step.py --config cfg4 --steps 20 > $R/$O/train_cfg4.log 2>&1
head -8 $R/$O/train_cfg4/t_kernel_stats.csv | cut -c1-150
