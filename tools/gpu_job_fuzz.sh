#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 900 python tools/fuzz_forward.py "$@" 2>&1 | tail -4 | cut -c1-300; }
run --n 120 --seed 101 --scale chain
run --n 50 --seed 102 --scale chain --backward
run --n 40 --seed 103 --scale chain --attn
run --n 40 --seed 104 --scale chain --core-precision bf16
run --n 40 --seed 106 --core-precision bf16
run --n 12 --seed 105 --scale medium --backward
run --n 30 --seed 107 --backward --dropout
