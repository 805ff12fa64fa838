#!/bin/bash
TAG=${1:-r04d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_fullsize.py tests/test_gpu_actor_extra.py -m gpu -q --tb=short -s 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|relative L2|trunk bs=64|^E  " | tail -30
echo "== clocks"; ls /sys/class/drm/ 2>&1 | head; ls /sys/class/drm/card*/device/hwmon/ 2>&1 | head -5; rocm-smi --showclocks --showpower 2>&1 | tail -12
timeout 900 python tools/clock_watch.py 600 2>&1 | tee $OUT/clock_watch.txt | tail -30
