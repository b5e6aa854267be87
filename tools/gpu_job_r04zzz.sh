#!/bin/bash
# round 4, last call: the full GPU suite and the bench line at the round's final HEAD (kernel stats / PMC: r04_zz, same kernels on the bench path)
O=gpurun_out/r04zzz; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu > $O/r04_zzz_gpu_tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/r04_zzz_gpu_tests.log
timeout 900 python bench.py > $O/r04_zzz_bench_n1.json 2> $O/bench.err; echo "bench exit=$?"; cut -c1-260 $O/r04_zzz_bench_n1.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
