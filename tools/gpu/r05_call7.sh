#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_gemv_s.py tests/test_gpu_kernels.py -q -x -k "gemv_s or qk_rms or rms_norm" ) > gpurun_out/r05_c7_pytest.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c7_pytest.txt
( time timeout 500 python tools/ab_libs.py 3 default@VRA_GS_SKEW=0 default ) > gpurun_out/r05_c7_ab_skew.txt 2>&1
true
