#!/bin/bash
TAG=${1:-r04c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q --tb=short --maxfail=8 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|relative L2|trunk bs=64" $OUT/pytest_gpu.log | tail -30
echo "== clocks"; timeout 600 python tools/clock_watch.py 300 2>&1 | tee $OUT/clock_watch.txt | tail -30
