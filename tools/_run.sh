mkdir -p gpurun_out/r02u; export TMPDIR=/tmp; O=gpurun_out/r02u
timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_actor.py -q --tb=short -x > $O/pytest.log 2>&1; echo pytest rc=$?; tail -4 $O/pytest.log | cut -c1-250
python bench.py --no-train --no-cpu-baseline --exec-steps 100 --exec-warmup 10 > $O/bench.json 2>$O/err.txt; python tools/bench_summary.py $O/bench.json | grep -i "_chain\|fused"
