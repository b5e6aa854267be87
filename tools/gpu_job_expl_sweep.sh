#!/bin/bash
# explicit bf16 core: query tiles per wave x split size, cfg4 / cfg5 forward time with core_precision = bf16
for nq in 2 4; do for ck in 0 512 1024 2048; do
  echo "== NQ=$nq CHUNK=$ck"
  HN_BF16_EXPL_NQ=$nq HN_BF16_EXPL_CHUNK=$ck timeout 200 python tools/bench_configs.py --cfg 4 5 --core-precision bf16 --steps 30 2>/dev/null | cut -c1-140
done; done
