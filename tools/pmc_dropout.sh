#!/bin/bash
# PMC passes over the cfg2 b=32 training step with and without dropout: where the dropping image kernels spend their cycles.
#   bash tools/pmc_dropout.sh <outdir>
O=${1:-gpurun_out/pmc_dropout}; mkdir -p $O; R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "0.25 0.25:drop" "0.0 0.0:none"; do
  a=${v%%:*}; n=${v##*:}
  python tools/pmc_kernels.py --match "attn_core_kernel<1, 4" "attn_bwd_dq_kernel<1, 4" --timeout 300 \
    --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32" \
    -- python $R/tools/dropout_breakdown.py $a > $O/pmc_$n.json 2> $O/pmc_$n.err
  tail -1 $O/pmc_$n.err
done
wc -c $O/*.json
