#!/bin/bash
# round 4, call e: dK/dV kernel with the query side in LDS (A/B), staged-model regression check, PMC of the glds GEMMs
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_bchain.py tests/test_gpu_ops.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
for i in 1 2; do
timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 > $O/train_new_$i.log
HN_NO_DKV_LDS=1 timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 > $O/train_nodkvlds_$i.log
done
cat $O/train_new_*.log $O/train_nodkvlds_*.log | cut -c1-200
timeout 300 python tools/bench_tuned.py > $O/tuned_new.log 2>&1; tail -6 $O/tuned_new.log | cut -c1-300
HN_NO_GLDS_GEMM=1 timeout 300 python tools/bench_tuned.py > $O/tuned_noglds.log 2>&1; tail -6 $O/tuned_noglds.log | cut -c1-300
timeout 900 python tools/pmc_kernels.py --match gemm_nt_glds gemm_tn_glds attn_bwd_dkv attn_bwd_dq "attn_core_kernel<4" --timeout 240 --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" -- python tools/train_step.py --config cfg4 --steps 4 --warmup 2 > $O/r04_e_pmc_train_cfg4.json 2> $O/pmc.err; head -c 1500 $O/r04_e_pmc_train_cfg4.json; tail -3 $O/pmc.err
