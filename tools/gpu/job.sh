#!/bin/bash
# One parameterised GPU job script (replaces the per-call scripts of round 5):  gpurun -- bash tools/gpu/job.sh <job> [args...]
# Several jobs in one call:  bash tools/gpu/job.sh multi "probe" "bench --no-cpu" ...
# Every job writes under gpurun_out/<round>_<job>*; summaries that are evidence get copied into profiles/ afterwards.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
RN=${ROUND:-r06}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp OMP_NUM_THREADS=16
job=$1; shift
case "$job" in
  multi)      # each argument is one job line (word-split)
    for j in "$@"; do bash $R/tools/gpu/job.sh $j; done ;;
  probe)      # tools/prologue_probe: where a launch's first microseconds go, by kernarg placement
    for v in unset 0 1; do
      ( if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi; timeout 120 tools/prologue_probe ) > $O/${RN}_prologue_probe_kernarg_$v.txt 2>&1
    done ;;
  bench)      # bench <tag> [bench.py flags]
    tag=$1; shift
    ( time timeout 900 python bench.py "$@" ) > $O/${RN}_bench_$tag.json 2> $O/${RN}_bench_$tag.err ;;
  ab)         # ab <tag> rounds libA libB ...   (tools/ab_libs.py)
    tag=$1; shift
    ( timeout ${T:-1200} python tools/ab_libs.py "$@" ) > $O/${RN}_ab_$tag.txt 2>&1 ;;
  pytest)     # pytest <tag> [pytest args]: GPU tests
    tag=$1; shift
    ( time timeout ${T:-2400} python -m pytest tests -q -m gpu "$@" ) > $O/${RN}_pytest_$tag.txt 2>&1
    echo "== rc $?" >> $O/${RN}_pytest_$tag.txt ;;
  py)         # py <tag> tools/x.py args...
    tag=$1; shift
    ( time timeout ${T:-600} python "$@" ) > $O/${RN}_$tag.txt 2>&1 ;;
  timelines)  # stamp timelines of kernels E / W and decode attention (needs libvra_ts.so: make B=build_ts EXTRA=-DVRA_GEMV_TS OUT=../libvra_ts.so)
    ( export VRA_LIB=$R/vllm_rs_amd/libvra_ts.so
      ( timeout 200 python tools/gemv_s_ts.py 0 1 2 3 ) > $O/${RN}_timeline_kernel_e.txt 2>&1
      ( for c in "1 150" "1 384" "1 1024" "1 8000" "32 150"; do echo "== $c"; timeout 100 python tools/attn_ts.py $c; done ) > $O/${RN}_timeline_attn_decode.txt 2>&1
      ( timeout 200 python tools/gemv_w_ts.py 32 0 1 2 3 ) > $O/${RN}_timeline_kernel_w.txt 2>&1 ) ;;
  trace)      # trace <tag> [bench flags]: rocprofv3 kernel trace of the eager decode bench (VRA_LIB / env vars pass through) -> per-kernel summary
    tag=$1; shift
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -- python $R/bench.py --steps 32 --warmup 4 --no-graph --no-extras "$@" > $O/prof_$tag.log 2>&1 )
    db=$(find $O/prof_$tag -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db 14 > $O/${RN}_trace_$tag.txt 2>&1
    rm -rf $O/prof_$tag ;;
  trace_prefill)  # trace_prefill <tag> [tokens]: rocprofv3 kernel trace of tools/prefill_once.py (two prompts of `tokens`, default 4096) -> per-kernel summary
    tag=$1; shift
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -- python $R/tools/prefill_once.py "$@" > $O/prof_$tag.log 2>&1 )
    db=$(find $O/prof_$tag -name "*.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py $db 20 > $O/${RN}_trace_$tag.txt 2>&1
    rm -rf $O/prof_$tag ;;
  profiles)   # rocprofv3 traces + counter passes (tools/collect_profiles.sh)
    ROUND=$RN bash tools/collect_profiles.sh > $O/${RN}_collect.log 2>&1 ;;
  final)      # the round's closing evidence: bench line as the driver runs it, full GPU suite, smoke
    ( time timeout 900 python bench.py ) > $O/${RN}_bench_default.json 2> $O/${RN}_bench_default.err
    ( time timeout 2400 python -m pytest tests -q -m gpu ) > $O/${RN}_final_pytest_gpu.txt 2>&1
    echo "== rc $?" >> $O/${RN}_final_pytest_gpu.txt
    ( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/${RN}_final_smoke.txt 2>&1 ;;
  *) echo "unknown job $job"; exit 2 ;;
esac
true
