#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O
timeout 900 python tools/fuzz_forward.py --scale tuned --n 24 --backward > $O/fuzz_tuned.log 2>&1; echo "tuned exit=$?"; tail -4 $O/fuzz_tuned.log
timeout 600 python tools/fuzz_forward.py --scale tuned --n 12 --seed 5 --backward --dropout > $O/fuzz_tuned_drop.log 2>&1; echo "tuned dropout exit=$?"; tail -3 $O/fuzz_tuned_drop.log
timeout 600 python tools/fuzz_forward.py --scale staged --n 30 --seed 3 --backward > $O/fuzz_staged.log 2>&1; echo "staged exit=$?"; tail -3 $O/fuzz_staged.log
timeout 600 python tools/fuzz_forward.py --scale medium --n 16 --seed 2 --backward > $O/fuzz_medium.log 2>&1; echo "medium exit=$?"; tail -3 $O/fuzz_medium.log
timeout 600 python tools/fuzz_forward.py --scale small --n 40 --seed 7 --backward --attn > $O/fuzz_small.log 2>&1; echo "small exit=$?"; tail -3 $O/fuzz_small.log
