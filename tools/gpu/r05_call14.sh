#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
export VRA_LIB=$PWD/vllm_rs_amd/libvra_ts.so
( timeout 200 python tools/gemv_s_ts.py 0 1 2 3 ) > gpurun_out/r05_timeline_kernel_e.txt 2>&1
( for c in "1 150" "1 384" "1 1024" "1 8000" "32 150"; do echo "== $c"; timeout 100 python tools/attn_ts.py $c; done ) > gpurun_out/r05_timeline_attn_decode.txt 2>&1
( timeout 200 python tools/gemv_w_ts.py 32 0 1 2 3 ) > gpurun_out/r05_timeline_kernel_w.txt 2>&1
true
