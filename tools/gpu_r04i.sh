#!/bin/bash
TAG=${1:-r04i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== supervised"; timeout 300 python tools/bench_supervised.py 2>&1 | tail -1 | tee $OUT/supervised.txt
echo "== tune A/B"; 
T2O_TUNE_GEMMS=1 python bench.py --quick --no-cpu-baseline --exec-steps 5 --exec-warmup 2 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train_step']; print('tune=1', t['ms_per_step'], t.get('supervised_step',{}).get('ms_per_step'), t.get('alternating_pair',{}).get('ms_per_pair'))"
T2O_TUNE_GEMMS=0 python bench.py --quick --no-cpu-baseline --exec-steps 5 --exec-warmup 2 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train_step']; print('tune=0', t['ms_per_step'], t.get('supervised_step',{}).get('ms_per_step'), t.get('alternating_pair',{}).get('ms_per_pair'))"
echo "== clocks"; timeout 900 python tools/clock_watch.py 600 2>&1 | tee $OUT/clock_watch.txt | tail -12
echo "== pmc"; bash tools/gpu_pmc.sh $TAG/pmc 2>&1 | tail -8
