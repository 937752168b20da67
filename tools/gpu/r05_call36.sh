#!/bin/bash
# the oversubscribed TP=8 stand-in as MANY short sessions (2 prefill + 2 decode forwards each, like tests/test_gpu_tp.py's three): how many in a row pass
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( time VRA_COMM_TIMEOUT_S=30 timeout 470 python tools/tp8_stress.py 2 50 400 ) > gpurun_out/r05_tp8_repro.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_tp8_repro.txt
true
