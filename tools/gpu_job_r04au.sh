#!/bin/bash
# round 4, call au: evidence run at HEAD (+ self-attention backward pair launch, one-token block on a chain in the training forward) -- full GPU suite, bench line, kernel stats (cfg4 forward / training, cfg2 forward), PMC, rooflines, tuned shapes, dropout cost
O=gpurun_out/r04au; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -x -q -m gpu > $O/r04_au_gpu_tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/r04_au_gpu_tests.log
timeout 900 python bench.py > $O/r04_au_bench_n1.json 2> $O/bench.err; echo "bench exit=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/train_cfg4 -o t -- python $R/tools/train_step.py --config cfg4 --steps 20 > $R/$O/train_cfg4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/fwd_cfg4 -o t -- python $R/tools/bench_configs.py --cfg 4 --core-precision fp32 --steps 20 > $R/$O/fwd_cfg4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/bench_cfg2 -o t -- python $R/bench.py --steps 20 --no-cpu-baseline --no-staged-models --no-train-step > $R/$O/bench_cfg2.log 2>&1
cd $R
timeout 900 python tools/pmc_kernels.py --match gemm_nt_glds gemm_tn_glds attn_bwd_dkv attn_bwd_dq "attn_core_kernel<4" --timeout 240 --sets "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" -- python $R/tools/train_step.py --config cfg4 --steps 4 --warmup 2 > $O/r04_au_pmc_train_cfg4.json 2> $O/pmc.err
timeout 600 python tools/pmc_collect.py --out $O/pmc --json $O/r04_au_pmc_cfg2_b32.json > $O/pmc_collect.log 2>&1; tail -2 $O/pmc_collect.log
timeout 900 python tools/roofline_configs.py --out $O/roof --tag r04_au --cfg 2 4 > $O/roofline.log 2>&1; tail -2 $O/roofline.log
for i in 1 2; do timeout 200 python tools/train_step.py --config cfg4 --steps 30 2>/dev/null | tail -1; timeout 200 python tools/train_step.py --config cfg2 --steps 20 2>/dev/null | tail -1; done | cut -c1-170 | tee $O/r04_au_train_steps.log
timeout 300 python tools/bench_configs.py --cfg 2 3 4 5 --steps 10 2>/dev/null | cut -c1-200 | tee $O/r04_au_configs.log
timeout 200 python tools/bench_tuned.py 2>/dev/null | tee $O/r04_au_tuned.log
timeout 300 python tools/bench_dropout.py 2>/dev/null | tee $O/r04_au_dropout_cost.txt
cut -c1-300 $O/r04_au_bench_n1.json
timeout 600 python tools/fuzz_forward.py --scale tuned --n 16 --backward 2>&1 | tail -1 | tee $O/r04_au_fuzz.log
timeout 600 python tools/fuzz_forward.py --scale small --n 30 --seed 11 --backward --attn 2>&1 | tail -1 | tee -a $O/r04_au_fuzz.log
timeout 600 python tools/fuzz_forward.py --scale medium --n 12 --seed 4 --backward 2>&1 | tail -1 | tee -a $O/r04_au_fuzz.log
timeout 600 python tools/fuzz_forward.py --scale staged --n 20 --seed 9 --backward --dropout 2>&1 | tail -1 | tee -a $O/r04_au_fuzz.log
timeout 300 python tools/bench_context_split.py --json $O/r04_au_context_split_compute.json 2>&1 | tail -7
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
