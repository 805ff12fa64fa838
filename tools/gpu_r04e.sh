#!/bin/bash
TAG=${1:-r04e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_fullsize.py tests/test_gpu_actor_extra.py tests/test_gpu_actor.py -m gpu -q --tb=short -s 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|relative L2|trunk bs=64|^E  " | tail -30
ROOT=$PWD
cd /tmp
echo "== rocprofv3 hip-trace (eager)"; timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $ROOT/$OUT/hip -- python $ROOT/tools/step_only.py 10 0 0 > $ROOT/$OUT/step_only_hip.log 2>&1; echo "rc=$?"
f=$(find $ROOT/$OUT/hip -name "*hip_api_stats*.csv" | head -1); echo $f
cp $f $ROOT/$OUT/hip_api_stats.csv 2>/dev/null
head -40 $ROOT/$OUT/hip_api_stats.csv
rm -rf $ROOT/$OUT/hip
cd $ROOT
echo "== A/B overlap lang"; timeout 300 python tools/step_only.py 20 0 0 2>&1 | tail -1
T2O_NO_OVERLAP_LANG=1 timeout 300 python tools/step_only.py 20 0 0 2>&1 | tail -1
