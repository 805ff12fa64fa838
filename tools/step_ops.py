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
tr = Trainer(model, opt, graph_encoder=True)
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.episode_step(x, img, tgt, lengths=lengths)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0 and 'cpu' in str(e.device_type).lower():          # operator rows only: kernel rows would count the time twice
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('device time in the step: %.2f ms over %d (operator, shape) groups' % (tot / 1e3, len(rows)))
for dt, n, k, sh in rows[:int(os.environ.get('ROWS', '90'))]:
    print('%9.1f us %5d  %-42s %s' % (dt, n, k[:42], sh))
