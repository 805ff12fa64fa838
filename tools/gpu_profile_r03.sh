#!/bin/bash
# Round-3 evidence run on one GPU box: full test suite, smoke, default bench, rocprofv3 kernel stats of the SAME bench
# command, kernel stats of the graphed train step alone, PMC passes.  Logs -> gpurun_out/<tag>/ (copy to profiles/).
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > $OUT/gpu.txt; nproc >> $OUT/gpu.txt; lscpu | grep "Model name" >> $OUT/gpu.txt
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short --maxfail=20 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 8 $OUT/pytest_gpu.log | head -3
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/smoke.log
echo "== bench"; timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python tools/bench_summary.py $OUT/bench.json | head -14
ROOT=$PWD
cd /tmp
echo "== rocprofv3 (bench)"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- python $ROOT/bench.py --no-cpu-baseline > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof.err; echo "rocprof rc=$?"
find $ROOT/$OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $ROOT/$OUT/bench_kernel_stats.csv
rm -rf $ROOT/$OUT/prof
echo "== rocprofv3 (train step only, as bench.py runs it: eager encoder)"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof2 -- python $ROOT/tools/step_only.py 10 0 0 > $ROOT/$OUT/step_only.log 2>&1; echo "rc=$?"
f=$(find $ROOT/$OUT/prof2 -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_gaps.py $f 0.4 > $ROOT/$OUT/step_trace_gaps.txt 2>&1
find $ROOT/$OUT/prof2 -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $ROOT/$OUT/step_kernel_stats.csv
rm -rf $ROOT/$OUT/prof2
cd $ROOT
head -n 12 $OUT/bench_kernel_stats.csv | cut -c1-150
