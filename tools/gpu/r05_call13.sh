#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 500 python tools/ab_libs.py 3 default libvra_rax.so ) > gpurun_out/r05_c13_ab_rax.txt 2>&1
true
