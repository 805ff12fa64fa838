export TMPDIR=/tmp
timeout 300 python tools/_dbg.py 2>&1 | grep -v amdgpu.ids
