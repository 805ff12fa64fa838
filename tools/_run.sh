mkdir -p gpurun_out/r03d; export TMPDIR=/tmp; O=gpurun_out/r03d
timeout 900 python -m pytest tests/test_gpu_planner.py tests/test_gpu_actor_extra.py -q --tb=short > $O/pytest.log 2>&1; echo pytest rc=$?; tail -12 $O/pytest.log | cut -c1-250
timeout 300 python tools/bench_planner.py > $O/planner.txt 2>&1; cat $O/planner.txt
