#!/bin/bash
# round 5, GPU call 3: decode attention LAT form (parity incl. new split boundaries), kernel E deferred norm + ring variants A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp8kv.py -q -x -k "attention or decode or fp8" ) > gpurun_out/r05_c3_pytest_attn.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c3_pytest_attn.txt
( time timeout 400 python tools/ab_libs.py 2 libvra_base.so default libvra_re.so libvra_d4.so libvra_red4.so ) > gpurun_out/r05_c3_ab.txt 2>&1
true
