#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_gemv_s.py -q -x -k "gemv_s" ) > gpurun_out/r05_c8_pytest.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c8_pytest.txt
( time timeout 500 python tools/ab_libs.py 3 default@VRA_GS_SKEW=0 default@VRA_GS_SKEW=1 default ) > gpurun_out/r05_c8_ab_skew.txt 2>&1
true
