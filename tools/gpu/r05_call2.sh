#!/bin/bash
# round 5, GPU call 2: kernel E without the norm barrier (parity + A/B on one box), TP=8 stall A/B (hardware queues per runner)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_gpu_gemv_s.py tests/test_gpu_engine.py tests/test_gpu_full_depth.py tests/test_gpu_reproducible.py -q -x ) > gpurun_out/r05_c2_pytest_kernel_e.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c2_pytest_kernel_e.txt
( time timeout 300 python tools/ab_libs.py 2 libvra_base.so default libvra_ring_early.so ) > gpurun_out/r05_c2_ab_kernel_e.txt 2>&1
( time VRA_COMM_TIMEOUT_S=20 timeout 400 python tools/tp8_stress.py 40 4 330 1,4 ) > gpurun_out/r05_c2_tp8_stress.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c2_tp8_stress.txt
( time timeout 200 python -m pytest tests/test_gpu_tp.py -q -x -k tp8 ) > gpurun_out/r05_c2_pytest_tp8.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c2_pytest_tp8.txt
true
