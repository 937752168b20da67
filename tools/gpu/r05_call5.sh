#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp8kv.py -q -x -k "attention or decode or fp8" ) > gpurun_out/r05_c5_pytest_attn.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c5_pytest_attn.txt
( time timeout 500 python tools/ab_libs.py 3 libvra_base.so default default@VRA_ATTN_LAT=0 ) > gpurun_out/r05_c5_ab_attn.txt 2>&1
true
