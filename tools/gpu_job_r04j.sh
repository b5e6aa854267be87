#!/bin/bash
# A/B of the training step between the evidence commit (a773ac3, in _ab_old/) and HEAD on ONE box
O=gpurun_out/r04j; mkdir -p $O
for i in 1 2 3; do
(cd _ab_old && timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-130 | sed 's/^/old  /')
timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-130 | sed 's/^/head /'
HN_NO_ATTN_LDS=1 timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1 | cut -c1-130 | sed 's/^/head-noattnlds /'
done
