#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
true
