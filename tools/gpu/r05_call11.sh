#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp8kv.py tests/test_gpu_longctx.py -q -x -k "attention or decode or fp8 or long" ) > gpurun_out/r05_c11_pytest_attn.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c11_pytest_attn.txt
( time timeout 500 python tools/ab_libs.py 3 libvra_prev.so default ) > gpurun_out/r05_c11_ab_attn_blkvec.txt 2>&1
true
