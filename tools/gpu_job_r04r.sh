#!/bin/bash
O=gpurun_out/r04r; mkdir -p $O
timeout 400 tools/ubench/gemm_f32_bench 32768 3072 773 3 > $O/gemm3072.log 2>&1; echo "bench exit=$?"; grep -E "^variant|TN variant [01]|RACE" $O/gemm3072.log
timeout 400 tools/ubench/gemm_f32_bench 32768 1024 773 3 > $O/gemm1024.log 2>&1; echo "bench exit=$?"; grep -E "^variant|TN variant [01]|RACE" $O/gemm1024.log
