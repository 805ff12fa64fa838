#!/bin/bash
TAG=${1:-r04h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
echo "== rocprofv3 (train step only, eager)"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof2 -- python $ROOT/tools/step_only.py 10 0 0 > $ROOT/$OUT/step_only.log 2>&1; echo "rc=$?"
f=$(find $ROOT/$OUT/prof2 -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_gaps.py $f 0.4 > $ROOT/$OUT/step_trace_gaps.txt 2>&1
find $ROOT/$OUT/prof2 -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $ROOT/$OUT/step_kernel_stats.csv
python $ROOT/tools/step_sequence.py $f > $ROOT/$OUT/step_sequence.txt 2>&1
rm -rf $ROOT/$OUT/prof2
cd $ROOT
tail -2 $OUT/step_only.log; head -5 $OUT/step_trace_gaps.txt
