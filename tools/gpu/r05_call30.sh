#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( timeout 400 python tools/pre_dbg.py 5 ) > gpurun_out/r05_c30_pre_dbg.txt 2>&1
true
