#!/bin/bash
# decoder tests, then eager vs step-graph A/B with kernel traces of both (per-kernel durations under graph replay vs eager)
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== decoder tests"; timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_encoder.py -q --tb=short -x 2>&1 | tail -8
echo "== eager"; timeout 300 python tools/step_only.py 20 0 0 2>&1 | tail -2
echo "== step graph"; timeout 300 python tools/step_only.py 20 1 0 2>&1 | tail -3
ROOT=$PWD
cd /tmp
for mode in 0 1; do
  echo "== rocprofv3 (train step only, graph_step=$mode)"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof$mode -- python $ROOT/tools/step_only.py 10 $mode 0 > $ROOT/$OUT/step_only_$mode.log 2>&1; echo "rc=$?"
  f=$(find $ROOT/$OUT/prof$mode -name "*kernel_trace.csv" | head -1)
  python $ROOT/tools/trace_gaps.py $f 0.4 > $ROOT/$OUT/step_trace_gaps_$mode.txt 2>&1
  find $ROOT/$OUT/prof$mode -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $ROOT/$OUT/step_kernel_stats_$mode.csv
  rm -rf $ROOT/$OUT/prof$mode
  tail -2 $ROOT/$OUT/step_only_$mode.log
  head -8 $ROOT/$OUT/step_trace_gaps_$mode.txt
done
cd $ROOT
