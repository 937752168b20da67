#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 500 python tools/ab_libs.py 2 libvra_base.so default default@VRA_ATTN_LAT=0 default@VRA_ATTN_SPLIT_TILES=4 default@VRA_ATTN_LAT=0,VRA_ATTN_SPLIT_TILES=4 default@VRA_ATTN_SPLIT_TILES=2 default@VRA_ATTN_SPLIT_TILES=16 ) > gpurun_out/r05_c4_ab_attn.txt 2>&1
true
