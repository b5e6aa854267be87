#!/bin/bash
for a in 0 1 2 3 4 5 7; do echo "abl=$a: $(HN_BF16_ABL=$a timeout 120 tools/ubench/gemm_bf16_bench 32768 1024 773 50 2>&1 | grep 'stage + gemm' | cut -c24-60)"; done
