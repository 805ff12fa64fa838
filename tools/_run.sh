export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -q --tb=short 2>&1 | tail -8
