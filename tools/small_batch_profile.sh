#!/bin/bash
# per-kernel profile of the cfg1 forward at b = 1 and b = 4 (chain route and HN_NO_CHAIN=1), and graph-replay wall times of both routes
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for b in 1 4; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/b$b -o fw -- python $GRAFT_REPO_ROOT/tools/quick_cfg2.py $b 100 > $out/b$b.log 2>&1
  HN_NO_CHAIN=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/b${b}_nochain -o fw -- python $GRAFT_REPO_ROOT/tools/quick_cfg2.py $b 100 > $out/b${b}_nochain.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/small_batch.py --n 300 > $out/chain.log 2>&1
HN_NO_CHAIN=1 python tools/small_batch.py --n 300 > $out/nochain.log 2>&1
grep cfg1 $out/chain.log | cut -c1-260; echo; grep cfg1 $out/nochain.log | cut -c1-260
for b in 1 4; do echo "== b=$b chain"; cut -c1-110 $out/b$b/fw_kernel_stats.csv | head -14; done
