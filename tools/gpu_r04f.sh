#!/bin/bash
TAG=${1:-r04f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q --tb=short --maxfail=8 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|relative L2|trunk bs=64" $OUT/pytest_gpu.log | tail -30
echo "== A/B BN fold"; 
for i in 1 2; do
T2O_BN_FOLD=1 timeout 300 python tools/step_only.py 30 0 0 2>&1 | tail -1
T2O_BN_FOLD=0 timeout 300 python tools/step_only.py 30 0 0 2>&1 | tail -1
done
echo "== graph"; T2O_BN_FOLD=1 timeout 300 python tools/step_only.py 30 1 0 2>&1 | tail -2
