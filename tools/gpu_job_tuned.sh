#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== per-param grads (round-2 protocol)"; python tools/bench_tuned.py --per-param-grads 2>/dev/null | cut -c1-200
echo "== flat grads"; python tools/bench_tuned.py 2>/dev/null | cut -c1-200
echo "== flat + gemm_big from N>=96"; HN_GEMM_BIG_MIN_N=96 python tools/bench_tuned.py 2>/dev/null | cut -c1-200
echo "== flat + gemm_big N>=96 + tn_lds from 32"; HN_GEMM_BIG_MIN_N=96 HN_TN_LDS_MIN=32 python tools/bench_tuned.py 2>/dev/null | cut -c1-200
echo "== flat + gemm_big N>=32 + tn_lds from 32"; HN_GEMM_BIG_MIN_N=32 HN_TN_LDS_MIN=32 python tools/bench_tuned.py 2>/dev/null | cut -c1-200
