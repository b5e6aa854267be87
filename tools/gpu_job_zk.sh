#!/bin/bash
# evidence run after the bf16 K/V projection: full GPU suite, bench line, rooflines of the patch-bag configs (fp32 + bf16), all
# configs fp32 vs bf16, LDS conflict counters of the projection kernel, smoke
# usage: bash tools/gpu_job_zk.sh <tag> <git head>
cd $GRAFT_REPO_ROOT
tag=${1:-r03_zk}; out=gpurun_out/$tag; mkdir -p $out
export HN_GIT_HEAD=$2
timeout 1500 python -m pytest tests -q -m gpu > $out/${tag}_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $out/${tag}_gpu_tests.log
timeout 600 python bench.py > $out/${tag}_bench_n1.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/${tag}_bench_n1.json
timeout 1500 python tools/roofline_configs.py --out $out --tag $tag --cfg 4 5 > $out/roofline.log 2>&1; echo "roofline rc=$?"; tail -5 $out/roofline.log
python tools/bench_configs.py --json $out/${tag}_configs_fp32_vs_bf16core.json 2>/dev/null | cut -c1-220
timeout 120 python tools/pmc_kernels.py --timeout 100 --match gemm_bf16_kernel --sets "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" -- $PWD/tools/ubench/gemm_bf16_bench 32768 1024 773 5 > $out/${tag}_gemm_bf16_lds_pmc.json 2>/dev/null; cat $out/${tag}_gemm_bf16_lds_pmc.json
timeout 60 tools/ubench/gemm_bf16_bench 32768 1024 773 50 | tee $out/${tag}_gemm_bf16_ubench.txt
timeout 60 tools/ubench/gemm_bf16_bench 16384 1024 773 50 | tee -a $out/${tag}_gemm_bf16_ubench.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
