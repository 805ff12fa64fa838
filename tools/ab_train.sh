#!/bin/bash
# A/B of the episode train step on ONE box (timings vary a few % between boxes).
# usage: tools/ab_train.sh "VAR=a VAR2=b" "VAR=c" ...   (each argument = one environment setting; '' = defaults)
for cfg in "$@"; do
  env $cfg python bench.py --quick --no-cpu-baseline --exec-steps 5 --exec-warmup 2 --steps 10 --warmup 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('train_step',{}); print('[$cfg]', t.get('ms_per_step'), 'host', t.get('host_enqueue_ms_per_step'), t.get('images_per_sec'), t.get('loss'), 'graphs', t.get('encoder_hipgraphs'), t.get('step_hipgraphs'), d.get('error',''))"
done
