#!/bin/bash
O=gpurun_out/r04w; mkdir -p $O
timeout 600 python tools/bench_context_split.py --json $O/r04_w_context_split_compute.json 2>&1 | tail -12
timeout 600 python tools/bench_context_split.py --batch 2 --json $O/r04_w_context_split_compute_b2.json 2>&1 | tail -12
