export TMPDIR=/tmp
python tools/bench_conv.py 64 fdws 2>&1 | grep -v amdgpu.ids
