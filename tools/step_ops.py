"""Which framework operators make up the small-launch tail of the episode train step: torch.profiler over ONE step
(after warm-up + graph capture), device time by (operator, input shapes).  Diagnostic, not part of the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import t2onet_amd  # noqa: E402
from t2onet_amd.actor import Actor  # noqa: E402
from t2onet_amd.train import Trainer  # noqa: E402
from bench import synthetic_requests  # noqa: E402

dev = torch.device('cuda:0')
B, H, W = 64, 256, 256
opt = t2onet_amd.default_options()
torch.manual_seed(10)
model = Actor(opt).to(dev).train()
model.use_channels_last()
tr = Trainer(model, opt)                      # eager, as bench.py runs it
g = torch.Generator().manual_seed(10)
img = torch.rand(B, 3, H, W, generator=g).to(dev)
tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
x = synthetic_requests(B, g)
lengths = (x != 0).sum(1)
x = x.to(dev)
for _ in range(4):
    tr.episode_step(x, img, tgt, lengths=lengths)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.episode_step(x, img, tgt, lengths=lengths)
    torch.cuda.synchronize()
rows = {}
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0 and 'cpu' in str(e.device_type).lower():          # operator rows only: kernel rows would count the time twice
        site = next((fr for fr in e.stack if '/t2onet_amd/' in fr and 'torch/' not in fr), e.stack[0] if e.stack else '?')
        site = site.split('/t2onet_amd/')[-1][:60]
        k = (e.key, str(e.input_shapes)[:70], site)
        r = rows.setdefault(k, [0.0, 0])
        r[0] += dt
        r[1] += e.count
rows = sorted(((v[0], v[1]) + k for k, v in rows.items()), reverse=True)
tot = sum(r[0] for r in rows)
print('device time in the step: %.2f ms over %d (operator, shape, call site) groups; launches %d' % (tot / 1e3, len(rows), sum(r[1] for r in rows)))
by = {}
for dt, n, k, sh, site in rows:
    if not k.startswith('t2o') and 'Fn' not in k:
        by[site] = by.get(site, 0) + n
print('framework operator calls with device time, by call site:')
for site, n in sorted(by.items(), key=lambda kv: -kv[1])[:60]:
    print('  %4d  %s' % (n, site))
for dt, n, k, sh, site in rows[:int(os.environ.get('ROWS', '120'))]:
    print('%9.1f us %5d  %-34s %-70s %s' % (dt, n, k[:34], sh, site))
