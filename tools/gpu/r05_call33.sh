#!/bin/bash
# final evidence of the round with the final library: stamp timelines, rocprofv3 traces + counter passes, driver-shaped bench line, full GPU suite, smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
( export VRA_LIB=$R/vllm_rs_amd/libvra_ts.so
  ( timeout 200 python tools/gemv_s_ts.py 0 1 2 3 ) > gpurun_out/r05_timeline_kernel_e.txt 2>&1
  ( for c in "1 150" "1 384" "1 1024" "1 8000" "32 150"; do echo "== $c"; timeout 100 python tools/attn_ts.py $c; done ) > gpurun_out/r05_timeline_attn_decode.txt 2>&1
  ( timeout 200 python tools/gemv_w_ts.py 32 0 1 2 3 ) > gpurun_out/r05_timeline_kernel_w.txt 2>&1 )
bash tools/collect_profiles.sh > gpurun_out/r05_collect.log 2>&1
cd $R
( time timeout 600 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
( time timeout 1500 python -m pytest tests -q -m gpu ) > gpurun_out/r05_final_pytest_gpu.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_final_pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/r05_final_smoke.txt 2>&1
true
