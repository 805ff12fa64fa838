#!/bin/bash
# Round-4 evidence run on one GPU box: full test suite, smoke, default bench, rocprofv3 kernel stats of the SAME bench
# command, kernel stats + trace gaps of the eager train step alone and of the whole-step hipGraph, HIP API stats, clocks.
# Logs -> gpurun_out/<tag>/ (copy what is to be judged to profiles/).
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > $OUT/gpu.txt; nproc >> $OUT/gpu.txt; lscpu | grep "Model name" >> $OUT/gpu.txt
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q --tb=short --maxfail=20 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -8
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/smoke.log
echo "== bench"; timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python tools/bench_summary.py $OUT/bench.json | head -16
ROOT=$PWD
cd /tmp
echo "== rocprofv3 (bench)"; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- python $ROOT/bench.py --no-cpu-baseline > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof.err; echo "rocprof rc=$?"
find $ROOT/$OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $ROOT/$OUT/bench_kernel_stats.csv
rm -rf $ROOT/$OUT/prof
for mode in 0 1; do
  echo "== rocprofv3 (train step only, graph_step=$mode)"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof$mode -- python $ROOT/tools/step_only.py 10 $mode 0 > $ROOT/$OUT/step_only_$mode.log 2>&1; echo "rc=$?"
  f=$(find $ROOT/$OUT/prof$mode -name "*kernel_trace.csv" | head -1)
  python $ROOT/tools/trace_gaps.py $f 0.4 > $ROOT/$OUT/step_trace_gaps_$mode.txt 2>&1
  find $ROOT/$OUT/prof$mode -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $ROOT/$OUT/step_kernel_stats_$mode.csv
  rm -rf $ROOT/$OUT/prof$mode
done
echo "== rocprofv3 hip-trace (eager)"; timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $ROOT/$OUT/hip -- python $ROOT/tools/step_only.py 10 0 0 > $ROOT/$OUT/step_only_hip.log 2>&1
f=$(find $ROOT/$OUT/hip -name "*hip_api_stats*.csv" | head -1); cp $f $ROOT/$OUT/hip_api_stats.csv 2>/dev/null; rm -rf $ROOT/$OUT/hip
cd $ROOT
echo "== un-profiled A/B"; for i in 1 2; do timeout 300 python tools/step_only.py 30 0 0 2>&1 | tail -1; timeout 300 python tools/step_only.py 30 1 0 2>&1 | tail -2 | head -1; done | tee $OUT/eager_vs_graph.txt
echo "== supervised"; timeout 300 python tools/bench_supervised.py 2>&1 | tail -1 | tee $OUT/supervised.txt
echo "== clocks"; timeout 900 python tools/clock_watch.py 600 2>&1 | tee $OUT/clock_watch.txt | tail -12
head -n 12 $OUT/bench_kernel_stats.csv | cut -c1-150
