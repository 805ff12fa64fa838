mkdir -p gpurun_out/r03e; export TMPDIR=/tmp; O=gpurun_out/r03e
timeout 1200 python -m pytest tests -m gpu -q --tb=short > $O/pytest.log 2>&1; echo pytest rc=$?; grep -n "passed\|failed\|FAILED\|Mismatched\|Max abs\|err_msg" $O/pytest.log | head -20
