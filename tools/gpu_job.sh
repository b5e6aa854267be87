#!/bin/bash
# ONE script for every GPU call (replaces the per-call tools/gpu_job_r*.sh of rounds 1-4):
#   gpurun --timeout T -- 'bash tools/gpu_job.sh <tag> <stage> [<stage> ...]'
# Output goes to gpurun_out/<tag>/ (merged back by gpurun); copy what is to be judged into profiles/ as <tag>_<what>.
# Stages:
#   tests[=<pytest args>]   python -m pytest <args: default "tests"> -x -q -m gpu
#   bench[=<bench args>]    python bench.py <args>                      -> bench_n1.json
#   stats=<name>:<command>  rocprofv3 --kernel-trace --stats of <command> -> <name>/ (csv), <name>.log
#   pmc_cfg2                tools/pmc_collect.py                         -> pmc_cfg2_b32.json
#   configs / tuned / dropout / trainsteps / smoke                      the tools of the same name
#   cmd=<shell command>     anything else (stdout+stderr -> cmd_<n>.log, tail printed)
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O; R=${GRAFT_REPO_ROOT:-$(pwd)}
n=0
for st in "$@"; do
  key=${st%%=*}; val=""; [[ "$st" == *=* ]] && val=${st#*=}
  case $key in
    tests) timeout 3000 python -m pytest ${val:-tests} -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests exit=$?"; tail -4 $O/gpu_tests.log ;;
    bench) timeout 1200 python bench.py $val > $O/bench_n1.json 2> $O/bench.err; echo "bench exit=$?"; cut -c1-400 $O/bench_n1.json ;;
    stats) name=${val%%:*}; cmd=${val#*:}
           (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$name -o t -- $cmd > $R/$O/$name.log 2>&1); echo "stats $name exit=$?" ;;
    pmc_cfg2) timeout 900 python tools/pmc_collect.py --out $O/pmc --json $O/pmc_cfg2_b32.json > $O/pmc_collect.log 2>&1; tail -2 $O/pmc_collect.log ;;
    configs) timeout 600 python tools/bench_configs.py --cfg 2 3 4 5 --steps 10 2>/dev/null | cut -c1-200 | tee $O/configs.log ;;
    tuned) timeout 300 python tools/bench_tuned.py 2>/dev/null | tee $O/tuned.log ;;
    dropout) timeout 400 python tools/bench_dropout.py 2>/dev/null | tee $O/dropout_cost.txt ;;
    trainsteps) for i in 1 2; do timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1; timeout 200 python tools/train_step.py --config cfg2 --steps 20 2>/dev/null | tail -1; done | cut -c1-200 | tee $O/train_steps.log ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log ;;
    cmd) n=$((n+1)); timeout 1500 bash -c "$val" > $O/cmd_$n.log 2>&1; echo "cmd_$n exit=$?"; tail -6 $O/cmd_$n.log | cut -c1-600 ;;
    *) echo "unknown stage $st" ;;
  esac
done
