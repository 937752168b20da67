#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 300 python bench.py --tp 2 --steps 64 --warmup 8 ) > gpurun_out/r05_bench_tp2.json 2> gpurun_out/r05_bench_tp2.err
( time timeout 200 python bench.py --model llama3-70b-tp8-rank --steps 64 --warmup 8 --no-extras ) > gpurun_out/r05_bench_70b_rank.json 2> gpurun_out/r05_bench_70b_rank.err
true
