#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 560 python tools/ab_longctx.py 2 - VRA_ATTN_MAX_WG=1024 VRA_ATTN_MAX_WG=2048 ) > gpurun_out/r05_c40_ab_attn_max_wg.txt 2>&1
true
