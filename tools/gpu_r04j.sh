#!/bin/bash
export TMPDIR=/tmp
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_actor.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -x 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|^E  " | tail -20
echo "== A/B dual bn"
python - <<'PY'
import re
p='tools/step_only.py'
PY
for i in 1 2; do
timeout 300 python tools/step_only.py 30 0 0 2>&1 | tail -1
T2O_NO_DUAL_BN=1 timeout 300 python tools/step_only.py 30 0 0 2>&1 | tail -1
done
