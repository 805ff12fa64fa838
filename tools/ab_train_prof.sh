#!/bin/bash
# GPU-busy time of the train step with / without the fused batch-norm kernels (rocprofv3 kernel stats)
export TMPDIR=/tmp
for f in 1 0; do
  export T2O_FUSED_BN=$f
  rm -rf gpurun_out/abprof$f; mkdir -p gpurun_out/abprof$f
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abprof$f -- python bench.py --steps 1 --warmup 0 --train-steps 6 --train-warmup 2 --no-cpu-baseline > gpurun_out/abprof$f/log.txt 2>&1
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/abprof$f/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
bn=sum(float(r['TotalDurationNs']) for r in rows if 'BatchNorm' in r['Name'] or 'k_bn_' in r['Name'])
print('T2O_FUSED_BN=$f GPU busy total %.1f ms over 8 train steps (+ small bench part); batch-norm kernels %.1f ms'%(tot/1e6,bn/1e6))
PY
  grep -o '"train_step": {[^}]*}' gpurun_out/abprof$f/log.txt | head -1 | cut -c1-160
  rm -f gpurun_out/abprof$f/*/*kernel_trace.csv
done
