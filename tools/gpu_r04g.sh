#!/bin/bash
TAG=${1:-r04g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_actor.py tests/test_gpu_actor_extra.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -s -x 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|relative L2|trunk bs=64|^E  " | tail -30
echo "== A/B arena"
for i in 1 2; do
timeout 300 python tools/step_only.py 30 0 0 2>&1 | tail -1
T2O_NO_ARENA=1 timeout 300 python tools/step_only.py 30 0 0 2>&1 | tail -1
done
echo "== graph"; timeout 300 python tools/step_only.py 30 1 0 2>&1 | tail -2
