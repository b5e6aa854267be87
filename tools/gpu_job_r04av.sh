#!/bin/bash
# round 4, call av: wider randomised parity sweep at HEAD (new seeds), forward + backward, with / without dropout, staged and tuned scales
O=gpurun_out/r04av; mkdir -p $O
( timeout 500 python tools/fuzz_forward.py --scale small --n 40 --seed 101 --backward --attn 2>&1 | tail -1
  timeout 500 python tools/fuzz_forward.py --scale small --n 30 --seed 202 --backward --dropout 2>&1 | tail -1
  timeout 500 python tools/fuzz_forward.py --scale medium --n 12 --seed 303 --backward 2>&1 | tail -1
  timeout 500 python tools/fuzz_forward.py --scale staged --n 24 --seed 404 --backward --dropout 2>&1 | tail -1
  timeout 500 python tools/fuzz_forward.py --scale tuned --n 16 --seed 505 --backward 2>&1 | tail -1 ) | tee $O/r04_av_fuzz.log
