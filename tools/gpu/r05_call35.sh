#!/bin/bash
# re-collection for the final library (host + dense-W ring changed): traces + counters, bench line, full GPU suite, smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp OMP_NUM_THREADS=16
bash tools/collect_profiles.sh > gpurun_out/r05_collect.log 2>&1
cd $R
( time timeout 600 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
( time timeout 1500 python -m pytest tests -q -m gpu ) > gpurun_out/r05_final_pytest_gpu.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_final_pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/r05_final_smoke.txt 2>&1
true
