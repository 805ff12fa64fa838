mkdir -p gpurun_out/r03f; export TMPDIR=/tmp; O=gpurun_out/r03f
timeout 1200 python -m pytest tests -m gpu -q --tb=short > $O/pytest.log 2>&1; echo pytest rc=$?; grep -n "passed\|failed\|FAILED\|Mismatched\|Max abs\|Error" $O/pytest.log | head -20
bash tools/ab_train.sh "T2O_NHWC=1" > $O/ab.txt 2>&1; cat $O/ab.txt
