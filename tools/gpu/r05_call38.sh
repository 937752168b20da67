#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( time timeout 400 python tools/invariance_sweep.py ) > gpurun_out/r05_invariance_sweep.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_invariance_sweep.txt
( time timeout 400 python tools/repro_sweep.py ) > gpurun_out/r05_repro_sweep.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_repro_sweep.txt
true
