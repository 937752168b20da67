#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( echo "== default"; timeout 100 python tools/inv_dbg.py llama3_8b 16 > /tmp/o.txt 2>&1; grep -v "^row" /tmp/o.txt
  echo "== VRA_NO_GEMV_S=1"; VRA_NO_GEMV_S=1 timeout 100 python tools/inv_dbg.py llama3_8b 16 > /tmp/o.txt 2>&1; grep -v "^row  *[0-689]\|^row 1" /tmp/o.txt
  echo "== VRA_NO_GEMV_W=1"; VRA_NO_GEMV_W=1 timeout 100 python tools/inv_dbg.py llama3_8b 16 > /tmp/o.txt 2>&1; grep -v "^row  *[0-689]\|^row 1" /tmp/o.txt
  echo "== VRA_NO_GEMV_S=1 VRA_NO_GEMV_W=1"; VRA_NO_GEMV_S=1 VRA_NO_GEMV_W=1 timeout 100 python tools/inv_dbg.py llama3_8b 16 > /tmp/o.txt 2>&1; grep -v "^row  *[0-689]\|^row 1" /tmp/o.txt
  echo "== tinyllama_q 17"; timeout 100 python tools/inv_dbg.py tinyllama_q 17 > /tmp/o.txt 2>&1; cat /tmp/o.txt
) > gpurun_out/r05_c23_inv_dbg.txt 2>&1
true
