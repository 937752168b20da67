#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( echo "== default"; timeout 100 python tools/inv_dbg.py llama3_8b 16
  echo "== VRA_GS_SKEW=0"; VRA_GS_SKEW=0 timeout 100 python tools/inv_dbg.py llama3_8b 16 | head -3
  echo "== no splits"; VRA_ATTN_SPLIT_TILES=100000 timeout 100 python tools/inv_dbg.py llama3_8b 16
  echo "== X_FRAG=0"; VRA_X_FRAG=0 timeout 100 python tools/inv_dbg.py llama3_8b 16 | head -20
) > gpurun_out/r05_c22_inv_dbg.txt 2>&1
true
