"""The OTHER half of train_seq2seqL1.py's alternation: the teacher-forced step (:51-65) at the benchmark's shape
(bs=64, 256x256, 5 operators + END).  bench.py's headline is the episode/L1 step (BASELINE.json's metric); this prints
the supervised step and the alternating pair beside it.  Synthetic FiveK-shaped batch, random-init weights."""
import os
import sys
import time

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
img_y = torch.rand(B, 6, 3, H, W, generator=g).to(dev)
x = synthetic_requests(B, g)
lengths = (x != 0).sum(1)
x = x.to(dev)
# operator sequences: START, a permutation of the 6 FiveK operators cut to 5, END (vocabulary ids: executor index + 3)
ops = torch.stack([torch.randperm(6, generator=g)[:5] for _ in range(B)])
vocab = torch.tensor([3, 4, 5, 6, 8, 9])[ops]
y = torch.cat([torch.full((B, 1), 1), vocab, torch.full((B, 1), 2)], 1).to(dev)
gt = (torch.rand(B, 5, 24, generator=g) * 2 - 1).to(dev)


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


sup = timed(lambda: tr.supervised_step(x, y, img, img_y, gt, lengths=lengths))
epi = timed(lambda: tr.episode_step(x, img, img_y[:, -1], lengths=lengths))
both = timed(lambda: (tr.supervised_step(x, y, img, img_y, gt, lengths=lengths), tr.episode_step(x, img, img_y[:, -1], lengths=lengths)), n=6, warm=1)
print('bs=%d %dx%d: supervised step %.2f ms (%.0f img/s), episode step %.2f ms (%.0f img/s), alternating pair %.2f ms'
      % (B, H, W, sup, B / sup * 1e3, epi, B / epi * 1e3, both))
