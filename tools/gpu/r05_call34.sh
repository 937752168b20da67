#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 500 python tools/ab_libs.py 3 default libvra_dw4.so ) > gpurun_out/r05_c34_ab_dw_ring.txt 2>&1
( for l in libvllm_rs_amd.so libvra_dw4.so; do echo "== $l"; VRA_LIB=$PWD/vllm_rs_amd/$l timeout 100 python tools/lm_head_times.py; done ) > gpurun_out/r05_c34_lm_head_times.txt 2>&1
true
