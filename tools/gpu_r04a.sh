#!/bin/bash
# Round-4 check on one GPU box: full GPU suite, eager vs whole-step-graph A/B of the train step, kernel trace of the eager step.
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --tb=short --maxfail=10 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 15 $OUT/pytest_gpu.log
echo "== eager"; timeout 300 python tools/step_only.py 20 0 0 2>&1 | tail -2
echo "== step graph"; timeout 300 python tools/step_only.py 20 1 0 2>&1 | tail -3
ROOT=$PWD
cd /tmp
echo "== rocprofv3 (train step only, eager)"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof2 -- python $ROOT/tools/step_only.py 10 0 0 > $ROOT/$OUT/step_only.log 2>&1; echo "rc=$?"
f=$(find $ROOT/$OUT/prof2 -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_gaps.py $f 0.4 > $ROOT/$OUT/step_trace_gaps.txt 2>&1
find $ROOT/$OUT/prof2 -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $ROOT/$OUT/step_kernel_stats.csv
rm -rf $ROOT/$OUT/prof2
cd $ROOT
cat $OUT/step_only.log | tail -2
head -30 $OUT/step_trace_gaps.txt
