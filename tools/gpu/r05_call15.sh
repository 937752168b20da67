#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 500 python tools/ab_libs.py 2 default default@VRA_W_DBG=2 ) > gpurun_out/r05_c15_ab_w_skipnorm.txt 2>&1
true
