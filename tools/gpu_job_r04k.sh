#!/bin/bash
for i in 1 2 3 4 5 6; do
timeout 200 python tools/train_step.py --config cfg4 --steps 60 2>/dev/null | tail -1 | cut -c60-112 | sed 's/^/head /'
(cd _ab_old && timeout 200 python tools/train_step.py --config cfg4 --steps 60 2>/dev/null | tail -1 | cut -c60-112 | sed 's/^/old  /')
done
