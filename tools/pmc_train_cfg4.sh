#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs, no other trace domain) over the cfg4 b=8 training step: MFMA busy / wait counters and
# FETCH / WRITE per launch for the step's main kernels.   bash tools/pmc_train_cfg4.sh <outdir>
O=${1:-gpurun_out/pmc_train}; mkdir -p $O; R=${GRAFT_REPO_ROOT:-$(pwd)}
python tools/pmc_kernels.py --match gemm_nt_glds gemm_tn_glds attn_bwd_dkv attn_bwd_dq "attn_core_kernel<4" latent_chain latent_bchain gemm_tn_lds_multi \
  --timeout 240 --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" \
  -- python $R/tools/train_step.py --config cfg4 --steps 4 --warmup 2 > $O/pmc_train_cfg4.json 2> $O/pmc.err
tail -2 $O/pmc.err; wc -c $O/pmc_train_cfg4.json
