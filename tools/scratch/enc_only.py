import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import t2onet_amd
from t2onet_amd.actor import Actor
from t2onet_amd.train import Trainer
import bench
dev = torch.device('cuda:0')
opt = t2onet_amd.default_options()
torch.manual_seed(10)
model = Actor(opt).to(dev).train()
model.use_channels_last()
g = torch.Generator().manual_seed(10)
B, H, W = 64, 256, 256
img = torch.rand(B, 3, H, W, generator=g).to(dev); tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
x = bench.synthetic_requests(B, g); lengths = (x != 0).sum(1); x = x.to(dev)
tr = Trainer(model, opt, graph_encoder=True)
for _ in range(4): tr.episode_step(x, img, tgt, lengths=lengths)
torch.cuda.synchronize()
slots = model.__dict__['_graphed_encoders'].slots[:5]
def run():
    for s in slots: s.fwd.replay()
    for s in reversed(slots): s.bwd.replay()
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize()
print('5 x (encoder forward + backward graphs): %.2f ms per step-equivalent' % ((time.perf_counter() - t0) / 10 * 1e3))
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
for s in slots: s.fwd.replay()
e[1].record()
for s in reversed(slots): s.bwd.replay()
e[2].record(); torch.cuda.synchronize()
print('forward graphs %.2f ms, backward graphs %.2f ms' % (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
