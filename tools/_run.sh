mkdir -p gpurun_out/r03c; export TMPDIR=/tmp; O=gpurun_out/r03c
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x > $O/pytest.log 2>&1; echo pytest rc=$?; tail -12 $O/pytest.log | cut -c1-250
bash tools/ab_train.sh "T2O_NHWC=1" "T2O_NHWC=1 T2O_OWN_WGRAD=1" > $O/ab.txt 2>&1; cat $O/ab.txt
