#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/ -q -m gpu ) > gpurun_out/r05_c19_pytest_gpu.txt 2>&1
echo "== rc $?" >> gpurun_out/r05_c19_pytest_gpu.txt
true
