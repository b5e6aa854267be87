#!/bin/bash
# last verification of the round: full GPU suite, bench line, dropout cost, tuned table on the final build
cd $GRAFT_REPO_ROOT
tag=${1:-r03_zg}; out=gpurun_out/$tag; mkdir -p $out
export HN_GIT_HEAD=$2
timeout 1500 python -m pytest tests -q -m gpu > $out/${tag}_gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $out/${tag}_gpu_tests.log
timeout 600 python bench.py > $out/${tag}_bench_n1.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/${tag}_bench_n1.json
python tools/bench_dropout.py 2>&1 | grep -v Warn | tee $out/${tag}_dropout_cost.txt
python tools/bench_tuned.py --json $out/${tag}_tuned_configs_b8.json 2>&1 | grep config
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
