#!/bin/bash
# sanity of bench.py's legs: plain run, watchdog path (--train-timeout 1), torchrun launch with one rank
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['n_gpus'], d.get('train_step'), 'cpu_baseline' in d)"; }
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | show plain
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-timeout 1 2>/dev/null | show watchdog; echo "rc=${PIPESTATUS[0]}"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | show torchrun
