#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 500 python tools/ab_libs.py 2 default default@VRA_ATTN_WG_PER_CU=2 default@VRA_ATTN_WG_PER_CU=4 ) > gpurun_out/r05_c32_ab_attn_wg_per_cu.txt 2>&1
true
