#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( time timeout 900 python -m pytest tests/test_gpu_reproducible.py -q -k "invariance_sweep" -s ) > gpurun_out/r05_c27_invariance.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c27_invariance.txt
true
