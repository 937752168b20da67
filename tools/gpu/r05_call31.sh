#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( time timeout 1500 python -m pytest tests -q -m gpu ) > gpurun_out/r05_c31_pytest_gpu.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c31_pytest_gpu.txt
true
