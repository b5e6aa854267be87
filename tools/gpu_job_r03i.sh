#!/bin/bash
cd $GRAFT_REPO_ROOT
for np in 0 1; do for w16 in 2 3; do
  if [ $np = 1 ]; then export HN_BF16_NO_PIPELINE=1; else unset HN_BF16_NO_PIPELINE; fi
  echo "no_pipeline=$np w16=$w16: $(HN_BF16_WAVES16=$w16 python tools/bench_configs.py --cfg 2 3 5 --core-precision bf16 --steps 30 2>/dev/null | grep -o '"cfg": [0-9]*\|"ms_per_forward": [0-9.]*' | paste - - | tr '\n' ' ')"
done; done
unset HN_BF16_NO_PIPELINE
python tools/bench_configs.py --cfg 3 --core-precision bf16x3 --steps 10 2>/dev/null | grep -o '"ms_per_forward": [0-9.]*'
