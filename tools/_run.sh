mkdir -p gpurun_out/r03k; export TMPDIR=/tmp; O=gpurun_out/r03k
timeout 900 python -m pytest tests/test_gpu_actor_extra.py -q --tb=short > $O/pytest.log 2>&1; echo pytest rc=$?; grep -n "passed\|failed\|FAILED\|Mismatched\|Max abs\|^E   [a-z_.0-9]*$\|assert" $O/pytest.log | head -30
