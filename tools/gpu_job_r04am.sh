#!/bin/bash
# round 4, call am: side-stream token encode + query fold on more blocks -- graph / chain / model / stream / small-batch tests, A/B timing
O=gpurun_out/r04am; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_chain.py tests/test_gpu_model.py tests/test_gpu_streams.py tests/test_gpu_context_split.py tests/test_gpu_staging.py tests/test_gpu_torchops.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -4 $O/tests.log
for i in 1 2; do
  timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-staged-models --no-train-step 2>/dev/null | cut -c1-150
  HN_NO_SIDE_ENCODE=1 timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-staged-models --no-train-step 2>/dev/null | cut -c1-150
  HN_NO_SIDE_ENCODE=1 HN_NO_QFOLD_CHAIN=1 timeout 200 python bench.py --steps 100 --no-cpu-baseline --no-staged-models --no-train-step 2>/dev/null | cut -c1-150
done | tee $O/r04_am_side_qfold_ab.log
timeout 200 python tools/small_batch.py 2>/dev/null | head -4 | cut -c1-230
HN_NO_SIDE_ENCODE=1 timeout 200 python tools/small_batch.py 2>/dev/null | head -4 | cut -c1-230
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/bench_cfg2 -o t -- python $R/bench.py --steps 20 --no-cpu-baseline --no-staged-models --no-train-step > $R/$O/bench_cfg2.log 2>&1
cd $R; f=$(find $O/bench_cfg2 -name '*kernel_stats.csv' | head -1); cp $f $O/r04_am_bench_cfg2_b32_kernel_stats.csv; rm -rf $O/bench_cfg2; head -10 $O/r04_am_bench_cfg2_b32_kernel_stats.csv | cut -c1-130
