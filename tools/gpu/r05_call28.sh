#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( time timeout 1100 python -m pytest tests/test_gpu_x_frag.py tests/test_gpu_engine.py tests/test_gpu_reproducible.py -q -s ) > gpurun_out/r05_c28_tests.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c28_tests.txt
true
