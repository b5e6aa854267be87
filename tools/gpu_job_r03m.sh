#!/bin/bash
# final-evidence run of round 3: full GPU suite, bench line, per-config rooflines (+PMC), training profiles, small-batch table
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03zi; mkdir -p $out
export HN_GIT_HEAD=$1
timeout 1200 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $out/gpu_tests.log
timeout 600 python tools/pmc_collect.py --out $out/pmc --json $out/r03_zi_pmc_cfg2_b32.json > $out/pmc.log 2>&1; tail -1 $out/pmc.log
cp $out/r03_zi_pmc_cfg2_b32.json profiles/ 2>/dev/null   # (bench.py reads the PMC traffic of the CURRENT attention.hip from profiles/)
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-400 $out/bench.json
timeout 1800 python tools/roofline_configs.py --out $out --tag r03_zi > $out/roofline.log 2>&1; echo "roofline rc=$?"; tail -6 $out/roofline.log
python tools/small_batch.py --n 300 --json $out/r03_zi_small_batch.json 2>&1 | grep '"case"' | cut -c1-200
cd /tmp; export TMPDIR=/tmp
for c in cfg2 cfg4; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/train_$c -o t -- python $GRAFT_REPO_ROOT/tools/train_step.py --config $c --steps 20 > $GRAFT_REPO_ROOT/$out/train_$c.log 2>&1; done
cd $GRAFT_REPO_ROOT
for c in cfg2 cfg4; do timeout 200 python tools/train_step.py --config $c --steps 30 2>/dev/null | tail -1; done
python tools/bench_configs.py --json $out/r03_zi_configs_fp32_vs_bf16core.json 2>/dev/null | cut -c1-200
bash tools/gpu_job_tuned_evidence.sh r03_zi 2>&1 | grep -v "rocprofv3\|^W2026\|^E2026" | tail -16
