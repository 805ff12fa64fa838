"""The host's own time for one eager episode train step: the same launches at batch size 1 (GPU work ~1/64), where the step time IS
the host's enqueue time.  python tools/host_floor.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import t2onet_amd
from t2onet_amd.actor import Actor
from t2onet_amd.train import Trainer
import bench

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt = t2onet_amd.default_options()
torch.manual_seed(10)
model = Actor(opt).to(dev).train()
model.use_channels_last()
tr = Trainer(model, opt)
g = torch.Generator().manual_seed(10)
H = W = 256
img = torch.rand(B, 3, H, W, generator=g).to(dev)
tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
x = bench.synthetic_requests(max(B, 2), g)[:B]
lengths = (x != 0).sum(1)
x = x.to(dev)
for _ in range(5):
    tr.episode_step(x, img, tgt, lengths=lengths)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    tr.episode_step(x, img, tgt, lengths=lengths)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('B=%d: host enqueue %.2f ms per step, until the GPU is done %.2f ms per step' % (B, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
