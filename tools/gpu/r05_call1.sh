#!/bin/bash
# round 5, GPU call 1: TP=8 stress with stage localisation, forced non-co-residency (CU mask), baseline bench on this box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time VRA_COMM_TIMEOUT_S=60 timeout 330 python tools/tp8_stress.py 25 2 290 ) > gpurun_out/r05_tp8_stress.txt 2>&1
echo "== tp8_stress rc $?" >> gpurun_out/r05_tp8_stress.txt
# forced non-co-residency in ONE process: every queue of this process restricted to 32 of the 256 CUs
( time HSA_CU_MASK=0:0-31 timeout 220 python -m pytest tests/test_gpu_gemv_s.py -q -x -k "k_slices or row_blocks or gemv_w_gptq or bitwise" ) > gpurun_out/r05_cumask_gemv_s.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_cumask_gemv_s.txt
( time HSA_CU_MASK=0:0-31 timeout 200 python tools/repro_sweep.py llama3_70b_tp8_rank ) > gpurun_out/r05_cumask_repro_tp8_rank.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_cumask_repro_tp8_rank.txt
( time timeout 400 python -m pytest tests/test_gpu_tp.py tests/test_gpu_kernels.py -q -x -k "tp or 143_to_221" ) > gpurun_out/r05_pytest_tp.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_pytest_tp.txt
( time timeout 300 python bench.py --no-cpu --no-parity ) > gpurun_out/r05_bench_call1.json 2> gpurun_out/r05_bench_call1.err
echo "== rc $?" >> gpurun_out/r05_bench_call1.err
tail -3 gpurun_out/r05_tp8_stress.txt gpurun_out/r05_cumask_gemv_s.txt gpurun_out/r05_cumask_repro_tp8_rank.txt gpurun_out/r05_pytest_tp.txt
