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
tr = Trainer(model, opt, graph_encoder=True, graph_step=True)
g = torch.Generator().manual_seed(10)
B, H, W = 64, 256, 256
img = torch.rand(B, 3, H, W, generator=g).to(dev); tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
x = bench.synthetic_requests(B, g); lengths = (x != 0).sum(1); x = x.to(dev)
for _ in range(4): tr.episode_step(x, img, tgt, lengths=lengths)
torch.cuda.synchronize()
sg = next(iter(tr._step_graphs.values()))
names = ['lang_fwd', 'copies+A', 'wait+copies', 'B', 'lang_bwd', 'C', 'update']
acc = {n: 0.0 for n in names}
steps = 10
main = torch.cuda.current_stream(dev); side = sg.lang_stream
t00 = time.perf_counter()
for _ in range(steps):
    t = [time.perf_counter()]
    side.wait_stream(main)
    with torch.cuda.stream(side):
        enc_out, enc_hidden, _ = model.lang_encoder(x, lengths)
        hidden = model.decoder._init_state(enc_hidden)
    t.append(time.perf_counter())
    sg.s_img.copy_(img); sg.s_target.copy_(tgt); sg.graphs[0].replay()
    t.append(time.perf_counter())
    main.wait_stream(side)
    with torch.no_grad():
        sg.s_enc.copy_(enc_out); sg.s_h.copy_(hidden[0]); sg.s_c.copy_(hidden[1])
    t.append(time.perf_counter())
    sg.graphs[1].replay()
    t.append(time.perf_counter())
    side.wait_stream(main)
    with torch.cuda.stream(side):
        torch.autograd.backward([enc_out, hidden[0], hidden[1]], [sg.s_enc.grad, sg.s_h.grad, sg.s_c.grad])
    t.append(time.perf_counter())
    sg.graphs[2].replay()
    t.append(time.perf_counter())
    main.wait_stream(side)
    tr._update()
    t.append(time.perf_counter())
    for i, n in enumerate(names): acc[n] += t[i + 1] - t[i]
torch.cuda.synchronize()
print('ms/step %.2f;  host ms: ' % ((time.perf_counter() - t00) / steps * 1e3) + ' '.join('%s %.2f' % (n, acc[n] / steps * 1e3) for n in names))
