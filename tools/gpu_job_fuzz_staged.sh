#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz_staged
run() { echo "== $*"; timeout 900 python tools/fuzz_forward.py "$@" 2>&1 | tail -4 | cut -c1-300; }
{
run --n 100 --seed 201 --scale staged
run --n 60 --seed 202 --scale staged --backward
run --n 40 --seed 203 --scale staged --attn
run --n 50 --seed 204 --scale staged --backward --dropout
run --n 30 --seed 205 --scale staged --core-precision bf16
HN_POISON_WS=1 run --n 40 --seed 206 --scale staged --backward
} 2>&1 | tee gpurun_out/fuzz_staged/fuzz.log
