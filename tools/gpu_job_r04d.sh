#!/bin/bash
# round 4, call d: bench line, training-step kernel stats + PMC of the two glds GEMMs, full GPU test suite
O=gpurun_out/r04d; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/r04_d_bench_n1.json 2> $O/bench.err; echo "bench exit=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_cfg4 -o t -- python $R/tools/train_step.py --config cfg4 --steps 20 > $R/$O/train_cfg4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/fwd_cfg4 -o t -- python $R/tools/bench_configs.py --cfg 4 --core-precision fp32 --steps 20 > $R/$O/fwd_cfg4.log 2>&1
cd $R
timeout 900 python tools/pmc_kernels.py --match gemm_nt_glds gemm_tn_glds attn_bwd_dkv attn_bwd_dq "attn_core_kernel<4" --timeout 240 --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" -- python tools/train_step.py --config cfg4 --steps 4 --warmup 2 > $O/r04_d_pmc_train_cfg4.json 2> $O/pmc.err
find $O -name "*kernel_stats.csv" | head
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/gpu_tests.log
cut -c1-600 $O/r04_d_bench_n1.json
