#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel stats.  Logs -> gpurun_out/<tag>/.
# usage: tools/gpu_check.sh [tag] [extra]     extra: "notest" skips pytest
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > $OUT/gpu.txt; nproc >> $OUT/gpu.txt; lscpu | grep "Model name" >> $OUT/gpu.txt
if [ "$2" != "notest" ]; then
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -q --tb=short --maxfail=20 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 25 $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
fi
echo "== bench"; timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python tools/bench_summary.py $OUT/bench.json; tail -n 5 $OUT/bench.err
echo "== rocprofv3"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python bench.py --quick --no-cpu-baseline --exec-steps 20 --exec-warmup 5 --steps 5 --warmup 3 > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?"
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/prof
head -n 40 $OUT/kernel_stats.csv | cut -c1-200
