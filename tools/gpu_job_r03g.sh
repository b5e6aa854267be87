#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03g; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bchain.py tests/test_gpu_bf16.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|rel err" | head -20
echo "== cfg3 bf16 geometry sweep"
for w16 in 3 4; do for w32 in 0 3; do echo "w16=$w16 w32=$w32: $(HN_BF16_WAVES16=$w16 HN_BF16_WAVES32=$w32 python tools/bench_configs.py --cfg 3 5 --core-precision bf16 --steps 20 2>/dev/null | grep -o '"cfg": [0-9]*\|"ms_per_forward": [0-9.]*' | paste - - | tr '\n' ' ')"; done; done
echo "== cfg2 bf16"
python tools/bench_configs.py --cfg 2 --core-precision bf16 --steps 50 2>/dev/null | grep -o '"ms_per_forward": [0-9.]*'
cd /tmp; export TMPDIR=/tmp
for c in cfg2 cfg4; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_$c -o t -- python $GRAFT_REPO_ROOT/tools/train_step.py --config $c --steps 20 > $out/train_$c.log 2>&1; done
cd $GRAFT_REPO_ROOT
for c in cfg2 cfg4; do timeout 200 python tools/train_step.py --config $c --steps 30 2>/dev/null | tail -1; done
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
