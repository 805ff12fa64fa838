#!/bin/bash
# ONE GPU-box visit, parameterised (replaces the per-experiment gpu_r04[a-k].sh / gpu_profile_r0N.sh copies).
#   tools/gpu_round.sh <tag> <stage> [<stage> ...]
# Logs -> gpurun_out/<tag>/ (copy what is to be judged to profiles/).  Stages, run in the order given:
#   test[:<pytest args>]  pytest -m gpu (default: all of tests/)        smoke        __graft_entry__.smoke()
#   bench                 default bench.py (+ bench_detail.json)         benchprof    rocprofv3 --kernel-trace --stats of the same command
#   step[:<graph 0|1>]    rocprofv3 kernel stats + trace gaps of the train step alone (tools/step_only.py)
#   seq                   ordered kernel list of one train step (tools/step_sequence.py)
#   hip                   rocprofv3 --hip-trace --stats of the eager step
#   ab:"ENV=a|ENV=b"      un-profiled A/B of the train step under the given environments ('|' separated, '' = defaults)
#   egraph                eager vs whole-step hipGraph, alternating
#   pmc                   tools/gpu_pmc.sh (separate --pmc passes; never combined with trace domains beside --kernel-trace)
#   sup                   tools/bench_supervised.py           clocks   tools/clock_watch.py 600
#   py:<script args>      python <script args>  (log: py_<n>.log)       sh:<command>   bash -c <command>
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > $OUT/gpu.txt; nproc >> $OUT/gpu.txt; lscpu | grep "Model name" >> $OUT/gpu.txt
n=0
for stage in "$@"; do
  n=$((n+1))
  name=${stage%%:*}; arg=""; [ "$name" != "$stage" ] && arg=${stage#*:}
  echo "== [$n] $name $arg"
  case $name in
    test)
      timeout 2700 python -m pytest ${arg:-tests} -m gpu -q --tb=short --maxfail=20 > $OUT/pytest_gpu_$n.log 2>&1; echo "pytest rc=$?"
      grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu_$n.log | tail -12 ;;
    smoke)
      timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log ;;
    bench)
      timeout 1500 python bench.py $arg > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
      cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
      tail -n 1 $OUT/bench.json | wc -c; python tools/bench_summary.py $OUT/bench.json $OUT/bench_detail.json | head -60; tail -n 3 $OUT/bench.err ;;
    benchprof)
      cd /tmp
      timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- python $ROOT/bench.py --no-cpu-baseline --no-128 $arg > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof.err; echo "rocprof rc=$?"
      cd $ROOT
      find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $OUT/bench_kernel_stats.csv
      rm -rf $OUT/prof; head -n 14 $OUT/bench_kernel_stats.csv | cut -c1-160 ;;
    step)
      mode=${arg:-0}
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof$mode -- python $ROOT/tools/step_only.py 14 $mode 0 > $ROOT/$OUT/step_only_$mode.log 2>&1; echo "rc=$?"
      cd $ROOT
      f=$(find $OUT/prof$mode -name "*kernel_trace.csv" | head -1)
      python tools/trace_gaps.py $f 0.4 > $OUT/step_trace_gaps_$mode.txt 2>&1
      find $OUT/prof$mode -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $OUT/step_kernel_stats_$mode.csv
      rm -rf $OUT/prof$mode; tail -n 1 $OUT/step_only_$mode.log; head -n 24 $OUT/step_kernel_stats_$mode.csv | cut -c1-160 ;;
    seq)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/profseq -- python $ROOT/tools/step_only.py 6 0 0 > $ROOT/$OUT/step_only_seq.log 2>&1; echo "rc=$?"
      cd $ROOT
      f=$(find $OUT/profseq -name "*kernel_trace.csv" | head -1)
      python tools/step_sequence.py $f > $OUT/step_sequence.txt 2>&1
      rm -rf $OUT/profseq; wc -l $OUT/step_sequence.txt ;;
    hip)
      cd /tmp
      timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $ROOT/$OUT/hip -- python $ROOT/tools/step_only.py 10 0 0 > $ROOT/$OUT/step_only_hip.log 2>&1
      cd $ROOT
      f=$(find $OUT/hip -name "*hip_api_stats*.csv" | head -1); cp $f $OUT/hip_api_stats.csv 2>/dev/null; rm -rf $OUT/hip; head -n 8 $OUT/hip_api_stats.csv ;;
    ab)
      IFS='|' read -ra cfgs <<< "$arg"
      for rep in 1 2; do for cfg in "${cfgs[@]}"; do echo -n "[$cfg] "; env $cfg timeout 400 python tools/step_only.py 30 0 0 2>&1 | tail -1; done; done | tee $OUT/ab_$n.txt ;;
    egraph)
      for i in 1 2; do timeout 300 python tools/step_only.py 30 0 0 2>&1 | tail -1; timeout 300 python tools/step_only.py 30 1 0 2>&1 | tail -2 | head -1; done | tee $OUT/eager_vs_graph.txt ;;
    pmc)   bash tools/gpu_pmc.sh $TAG/pmc ;;
    sup)   timeout 300 python tools/bench_supervised.py 2>&1 | tail -1 | tee $OUT/supervised.txt ;;
    clocks) timeout 900 python tools/clock_watch.py 600 2>&1 | tee $OUT/clock_watch.txt | tail -12 ;;
    py)    timeout 1500 python $arg > $OUT/py_$n.log 2>&1; echo "rc=$?"; tail -n 40 $OUT/py_$n.log ;;
    sh)    timeout 1500 bash -c "$arg" > $OUT/sh_$n.log 2>&1; echo "rc=$?"; tail -n 40 $OUT/sh_$n.log ;;
    *)     echo "unknown stage $name" ;;
  esac
done
