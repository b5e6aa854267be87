cd $GRAFT_REPO_ROOT; out=gpurun_out/r03_zm; mkdir -p $out; export HN_GIT_HEAD=d8bdf43
timeout 600 python -m pytest tests/test_gpu_bf16proj.py tests/test_gpu_bf16.py tests/test_gpu_graph.py tests/test_gpu_model.py -q -m gpu > $out/r03_zm_gpu_tests_bf16.log 2>&1; tail -1 $out/r03_zm_gpu_tests_bf16.log
timeout 600 python tools/fuzz_forward.py --scale medium --core-precision bf16 --n 60 > $out/r03_zm_fuzz_medium_bf16.log 2>&1; tail -1 $out/r03_zm_fuzz_medium_bf16.log
timeout 1500 python tools/roofline_configs.py --out $out --tag r03_zm --cfg 4 5 > $out/roofline.log 2>&1; tail -4 $out/roofline.log
python tools/bench_configs.py --json $out/r03_zm_configs_fp32_vs_bf16core.json 2>/dev/null | cut -c1-220
