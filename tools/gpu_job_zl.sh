#!/bin/bash
# evidence run after the bf16 explicit core: full GPU suite, medium-scale bf16 fuzz, bench line, rooflines of the patch-bag configs
# (fp32 + bf16), all configs fp32 vs bf16, smoke
# usage: bash tools/gpu_job_zl.sh <tag> <git head>
cd $GRAFT_REPO_ROOT
tag=${1:-r03_zl}; out=gpurun_out/$tag; mkdir -p $out
export HN_GIT_HEAD=$2
timeout 1500 python -m pytest tests -q -m gpu > $out/${tag}_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $out/${tag}_gpu_tests.log
timeout 900 python tools/fuzz_forward.py --scale medium --core-precision bf16 --n 60 > $out/${tag}_fuzz_medium_bf16.log 2>&1; tail -1 $out/${tag}_fuzz_medium_bf16.log
timeout 600 python bench.py > $out/${tag}_bench_n1.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/${tag}_bench_n1.json
timeout 1500 python tools/roofline_configs.py --out $out --tag $tag --cfg 4 5 > $out/roofline.log 2>&1; echo "roofline rc=$?"; tail -5 $out/roofline.log
python tools/bench_configs.py --json $out/${tag}_configs_fp32_vs_bf16core.json 2>/dev/null | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
