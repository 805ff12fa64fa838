mkdir -p gpurun_out/r03l; export TMPDIR=/tmp; O=gpurun_out/r03l
timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_planner.py -q --tb=short > $O/pytest.log 2>&1; echo pytest rc=$?; tail -3 $O/pytest.log | cut -c1-200
for s in 1 0; do T2O_CHAIN_STATIC=$s python bench.py --no-train --no-cpu-baseline --exec-steps 100 --exec-warmup 10 > $O/bench_static$s.json 2>$O/err$s.txt; echo "== static=$s"; python tools/bench_summary.py $O/bench_static$s.json | grep -i "_chain\|fused"; done
