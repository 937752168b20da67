#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( VRA_X_FRAG=0 timeout 150 python tools/pre_dbg.py 31 ) > gpurun_out/r05_c18_pre_dbg_rowmajor.txt 2>&1
true
