"""Where the HOST time of an eager episode train step goes (cProfile over a few steps): python tools/host_profile.py [steps]"""
import cProfile, os, pstats, sys, time
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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tr = Trainer(model, opt, graph_encoder=False, graph_step=False)
g = torch.Generator().manual_seed(10)
B = int(os.environ.get('T2O_BATCH', '64'))
H = W = int(os.environ.get('T2O_SIZE', '256'))      # T2O_BATCH=2: the GPU work is negligible, what is timed is the host's own work
img = torch.rand(B, 3, H, W, generator=g).to(dev)
tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
x = bench.synthetic_requests(B, g)
lengths = (x != 0).sum(1)
x = x.to(dev)
for _ in range(4):
    tr.episode_step(x, img, tgt, lengths=lengths)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    tr.episode_step(x, img, tgt, lengths=lengths)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(40)
