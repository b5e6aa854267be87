#!/bin/bash
# kernel stats of cfg4 / cfg5 with the bf16 core + bf16 K/V projection
tag=${1:-bf16proj}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in 4 5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$c -o cfg$c -- python tools/bench_configs.py --cfg $c --core-precision bf16 --steps 20 > $out/prof$c.log 2>&1
  f=$(find $out/prof$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/cfg${c}_bf16_kernel_stats.csv && head -7 $f | cut -c1-150
  rm -rf $out/prof$c
done
