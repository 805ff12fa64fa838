#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats.  Logs -> gpurun_out/<tag>/.
# usage: tools/gpu_check.sh [tag] [extra]     extra: "sweep" also sweeps the per-thread work knobs
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > $OUT/gpu.txt; nproc >> $OUT/gpu.txt; lscpu | grep "Model name" >> $OUT/gpu.txt
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -q --tb=short --maxfail=20 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 25 $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 30 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python tools/bench_summary.py $OUT/bench.json; tail -n 5 $OUT/bench.err
if [ "$2" == "sweep" ]; then
  for it in 1 2 4 8; do echo "== T2O_ITERS=$it T2O_CHAIN_ITERS=$it"; T2O_ITERS=$it T2O_CHAIN_ITERS=$it timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/sweep_$it.json 2>> $OUT/bench.err; python tools/bench_summary.py $OUT/sweep_$it.json; done
fi
if [ "$2" == "train" ]; then
echo "== train step"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 5 > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "train rc=$?"; python -c "import json,sys; d=json.loads(open('$OUT/bench_train.json').read().strip().splitlines()[-1]); print(d.get('train_step'))"; tail -n 3 $OUT/bench_train.err
fi
echo "== rocprofv3"; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r head -n 30 | cut -c1-200
