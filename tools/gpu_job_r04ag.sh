#!/bin/bash
# round 4, call ag: where dropout costs at cfg2 b = 32 (kernel stats with / without dropout)
O=gpurun_out/r04ag; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for p in "0.25 0.25" "0.0 0.0" "0.25 0.0"; do
  tag=$(echo $p | tr ' .' '__')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/d_$tag -o t -- python $R/tools/dropout_breakdown.py $p > $R/$O/d_$tag.log 2>&1
  tail -1 $R/$O/d_$tag.log
  f=$(find $R/$O/d_$tag -name '*kernel_stats.csv' | head -1); cp $f $R/$O/r04_ag_dropout_${tag}_kernel_stats.csv; rm -rf $R/$O/d_$tag
done
