import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import t2onet_amd
from t2onet_amd.actor import Actor
from t2onet_amd.train import Trainer
import bench
dev = torch.device('cuda:0')
opt = t2onet_amd.default_options()
for mode in ('normal', 'no_lang', 'normal', 'no_lang'):
    torch.manual_seed(10)
    model = Actor(opt).to(dev).train()
    model.use_channels_last()
    g = torch.Generator().manual_seed(10)
    B, H, W = 64, 256, 256
    img = torch.rand(B, 3, H, W, generator=g).to(dev); tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
    x = bench.synthetic_requests(B, g); lengths = (x != 0).sum(1); x = x.to(dev)
    if mode == 'no_lang':
        with torch.no_grad():
            eo, (h, c), emb = model.lang_encoder(x, lengths)
        cached = (eo.detach(), (h.detach(), c.detach()), emb.detach())
        model.lang_encoder.forward = lambda *a, **k: cached
    tr = Trainer(model, opt, graph_encoder=True)
    for _ in range(4): tr.episode_step(x, img, tgt, lengths=lengths)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.episode_step(x, img, tgt, lengths=lengths)
    torch.cuda.synchronize()
    print(mode, 'ms/step %.2f' % ((time.perf_counter() - t0) / 10 * 1e3))
