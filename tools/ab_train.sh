#!/bin/bash
# A/B of the episode train step on ONE box (timings vary a few % between boxes):
# fused batch-norm kernels on / off x image-encoder hipGraphs on / off
for cfg in "1 1" "1 0" "0 0" "1 1"; do
  set -- $cfg
  T2O_FUSED_BN=$1 T2O_GRAPH_ENCODER=$2 python bench.py --steps 5 --warmup 2 --train-steps 8 --train-warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train_step']; print('fused_bn=$1 graph_encoder=$2', t.get('ms_per_step'), 'host', t.get('host_enqueue_ms_per_step'), t.get('images_per_sec'), t.get('loss'), t.get('error',''))"
done
