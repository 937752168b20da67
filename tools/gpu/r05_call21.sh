#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( for l in libvra_ne.so libvra_nb.so; do echo "== $l"; VRA_LIB=$PWD/vllm_rs_amd/$l timeout 100 python tools/invariance_sweep.py llama3_8b; done ) > gpurun_out/r05_c21_invariance_bisect.txt 2>&1
true
