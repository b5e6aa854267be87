#!/bin/bash
# round 4, call l: GraphedStep (tests, bench record), dropout tests after the hn_rng change
O=gpurun_out/r04l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_dropout.py tests/test_gpu_staged.py tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_gpu_bench_contract.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests exit=$?"; tail -15 $O/tests.log | cut -c1-200
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04l/bench.json'))
t=j['train_step']
print('headline',j['value'],'train',t['ms_per_step'],'graphed',t.get('graphed_ms_per_step'),'frac',t['roofline']['frac'])
print({k:(v['fwd_bwd_ms'],v.get('fwd_bwd_graphed_ms')) for k,v in j['staged_models']['configs'].items()})
PY
