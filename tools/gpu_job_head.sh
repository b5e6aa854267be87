#!/bin/bash
# verification of a HEAD: full GPU suite, bench line, smoke
# usage: bash tools/gpu_job_head.sh <tag> <git head>
cd $GRAFT_REPO_ROOT
tag=${1:-head}; out=gpurun_out/$tag; mkdir -p $out
export HN_GIT_HEAD=$2
timeout 1500 python -m pytest tests -q -m gpu > $out/${tag}_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $out/${tag}_gpu_tests.log
timeout 600 python bench.py > $out/${tag}_bench_n1.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/${tag}_bench_n1.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
