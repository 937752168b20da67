#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel traces of the decode bench at bs 1 and bs 32 (eager launches: kernel tracing and hipGraph replay do not mix on
#   ROCm 7.2), and SEPARATE counter passes (never combined with a trace domain): FETCH_SIZE, WRITE_SIZE, MFMA / busy cycles.
# plus (round 4) a kernel trace of a 128-token prefill (TTFT 1 x 128) and the bs 1 / bs 32 traces of BASELINE config 3 (Qwen2-7B AWQ).
# Outputs land in gpurun_out/prof_*/ ; tools/summarise_profiles.py turns them into profiles/r06_*.txt + r06_pmc.json.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 32 --warmup 4 --no-graph --no-extras"
rocprofv3 -L 2>/dev/null | grep -iE "MFMA|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|FETCH_SIZE|WRITE_SIZE" | head -60 > $OUT/prof_counters_available.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace_bs1 -- $B --batch 1 > $OUT/prof_trace_bs1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace_bs32 -- $B --batch 32 > $OUT/prof_trace_bs32.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch_bs1 -- $B --batch 1 > $OUT/prof_pmc_fetch_bs1.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_pmc_write_bs1 -- $B --batch 1 > $OUT/prof_pmc_write_bs1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/prof_pmc_mfma_bs1 -- $B --batch 1 > $OUT/prof_pmc_mfma_bs1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/prof_pmc_mfma_bs32 -- $B --batch 32 > $OUT/prof_pmc_mfma_bs32.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch_bs32 -- $B --batch 32 > $OUT/prof_pmc_fetch_bs32.log 2>&1
# prefill (kernel D): TTFT-shaped run, 2 prompts of 4096 tokens
P="python $R/tools/prefill_once.py"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace_prefill -- $P > $OUT/prof_trace_prefill.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/prof_pmc_mfma_prefill -- $P > $OUT/prof_pmc_mfma_prefill.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace_prefill128 -- $P 128 > $OUT/prof_trace_prefill128.log 2>&1
# BASELINE config 3: Qwen2-7B AWQ (same kernels, AWQ zero points + q/k/v bias)
Q="python $R/bench.py --model qwen2-7b-awq --steps 32 --warmup 4 --no-graph --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace_qwen2_bs1 -- $Q --batch 1 > $OUT/prof_trace_qwen2_bs1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace_qwen2_bs32 -- $Q --batch 32 > $OUT/prof_trace_qwen2_bs32.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch_qwen2_bs1 -- $Q --batch 1 > $OUT/prof_pmc_fetch_qwen2_bs1.log 2>&1
cd $R
for d in prof_trace_bs1 prof_trace_bs32 prof_trace_prefill prof_trace_prefill128 prof_trace_qwen2_bs1 prof_trace_qwen2_bs32; do
  db=$(find $OUT/$d -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db 25 > $OUT/$d.summary.txt 2>&1
done
for d in prof_pmc_fetch_bs1 prof_pmc_write_bs1 prof_pmc_mfma_bs1 prof_pmc_mfma_bs32 prof_pmc_fetch_bs32 prof_pmc_mfma_prefill prof_pmc_fetch_qwen2_bs1; do
  db=$(find $OUT/$d -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_pmc.py $db 40 > $OUT/$d.summary.txt 2>&1
done
python -c "
import hashlib,sys
sys.path.insert(0,'$R')
from vllm_rs_amd import _lib
import bench
print(hashlib.sha256(open(_lib.LIB_PATH,'rb').read()).hexdigest()[:16])
print(bench.src_sha16())" > $OUT/prof_lib_sha16.txt
# drop the raw databases (tens of MB): the summaries are what is committed
find $OUT -name "*.db" -path "*prof_*" -delete
ls $OUT
