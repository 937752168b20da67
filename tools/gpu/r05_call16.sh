#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_x_frag.py tests/test_gpu_engine.py tests/test_gpu_reproducible.py tests/test_gpu_full_depth.py -q -x ) > gpurun_out/r05_c16_pytest.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c16_pytest.txt
( time timeout 500 python tools/ab_libs.py 3 libvra_prev.so default ) > gpurun_out/r05_c16_ab_pre.txt 2>&1
true
