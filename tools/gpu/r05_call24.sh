#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( echo "== default"; timeout 100 python tools/inv_dbg.py llama3_8b 16 2>&1 | grep -v "^row"
  echo "== VRA_X_FRAG=0"; VRA_X_FRAG=0 timeout 100 python tools/inv_dbg.py llama3_8b 16 2>&1 | grep -v "^row"
  echo "== tinyllama_q 17"; timeout 100 python tools/inv_dbg.py tinyllama_q 17 2>&1 | grep -v "^row"
) > gpurun_out/r05_c24_inv_stages.txt 2>&1
true
