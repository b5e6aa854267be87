#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/m6
for kc in 208 64; do
  HN_TALL_KC=$kc timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/m6/kc$kc -o t -- python $R/tools/bench_tuned.py --configs blca kirp > $R/gpurun_out/m6/kc$kc.log 2>&1
  grep config $R/gpurun_out/m6/kc$kc.log
  grep -E "tall_narrow|gemm_tn_lds_kernel|splitk_reduce_wide|splitk_reduce_alpha" $R/gpurun_out/m6/kc$kc/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
find $R/gpurun_out/m6 -name '*kernel_trace.csv' -delete
