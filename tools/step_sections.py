"""GPU-clock sections of the graphed episode step (HIP events on the launch stream): request encoder forward, static-input
copies, the whole-step hipGraph replay, request encoder backward, Adam.  usage: python tools/step_sections.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
tr = Trainer(model, opt, graph_encoder=True, graph_step=True)
g = torch.Generator().manual_seed(10)
B, H, W = 64, 256, 256
img = torch.rand(B, 3, H, W, generator=g).to(dev)
tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
x = bench.synthetic_requests(B, g)
lengths = (x != 0).sum(1)
x = x.to(dev)
for _ in range(4):
    tr.episode_step(x, img, tgt, lengths=lengths)
torch.cuda.synchronize()
sg = next(iter(tr._step_graphs.values()))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
names = ['lang_fwd', 'copies', 'replay', 'lang_bwd', 'update']
acc = {n: 0.0 for n in names}
t0 = time.perf_counter()
for _ in range(steps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    enc_out, enc_hidden, _ = model.lang_encoder(x, lengths)
    hidden = model.decoder._init_state(enc_hidden)
    ev[1].record()
    sg.s_img.copy_(img); sg.s_target.copy_(tgt)
    with torch.no_grad():
        sg.s_enc.copy_(enc_out); sg.s_h.copy_(hidden[0]); sg.s_c.copy_(hidden[1])
    ev[2].record()
    sg.graph.replay()
    ev[3].record()
    torch.autograd.backward([enc_out, hidden[0], hidden[1]], [sg.s_enc.grad, sg.s_h.grad, sg.s_c.grad])
    ev[4].record()
    tr._update()
    ev[5].record()
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        acc[n] += ev[i].elapsed_time(ev[i + 1])
wall = (time.perf_counter() - t0) / steps * 1e3
print('per step (ms, synchronised after every step): wall %.2f ' % wall + ' '.join('%s %.3f' % (n, acc[n] / steps) for n in names))
