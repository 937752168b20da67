#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_gemv_s.py tests/test_gpu_engine.py tests/test_gpu_x_frag.py tests/test_gpu_full_depth.py tests/test_gpu_kernels.py -q -x ) > gpurun_out/r05_c9_pytest.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c9_pytest.txt
( time timeout 500 python tools/ab_libs.py 3 libvra_prev.so default ) > gpurun_out/r05_c9_ab_w.txt 2>&1
true
