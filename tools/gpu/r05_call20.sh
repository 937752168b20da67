#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( echo "== default"; timeout 120 python tools/invariance_sweep.py llama3_8b tinyllama_q; echo "== VRA_X_FRAG=0"; VRA_X_FRAG=0 timeout 120 python tools/invariance_sweep.py llama3_8b tinyllama_q ) > gpurun_out/r05_c20_invariance.txt 2>&1
true
