export TMPDIR=/tmp
timeout 300 python tools/_dbg.py 2>&1 | grep -v amdgpu.ids | head -5
timeout 900 python -m pytest tests/test_gpu_conv.py -q --tb=short 2>&1 | tail -15
timeout 600 python tools/bench_conv.py 64 fdw 2>&1 | grep -v amdgpu.ids
