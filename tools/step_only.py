"""N graphed episode train steps and nothing else (for rocprofv3 --kernel-trace): python tools/step_only.py [steps] [graph_step 0|1] [graph_encoder 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import t2onet_amd
from t2onet_amd.actor import Actor
from t2onet_amd.train import Trainer
import bench
if os.environ.get('T2O_NO_OVERLAP_LANG'):
    import t2onet_amd.actor as _A
    _A._OVERLAP_LANG = False

dev = torch.device('cuda:0')
opt = t2onet_amd.default_options()
torch.manual_seed(10)
model = Actor(opt).to(dev).train()
model.use_channels_last()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if os.environ.get('T2O_NO_DUAL_BN'):
    import t2onet_amd.encoder as _E
    _E._DUAL_BN = False
if os.environ.get('T2O_NO_BN_SUMS_EPI'):
    import t2onet_amd.encoder as _E2
    _E2._BN_SUMS_EPILOGUE = False
if os.environ.get('T2O_NO_ARENA'):
    Trainer._arena = lambda self, img, passes: None
tr = Trainer(model, opt, graph_encoder=(sys.argv[3] != '0') if len(sys.argv) > 3 else True, graph_step=(sys.argv[2] != '0') if len(sys.argv) > 2 else True)
g = torch.Generator().manual_seed(10)
B = int(os.environ.get('T2O_BATCH', '64'))
H = W = int(os.environ.get('T2O_SIZE', '256'))           # T2O_SIZE=128: the reference's own training size (datasets/FiveKdataset.py:68)
img = torch.rand(B, 3, H, W, generator=g).to(dev)
tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
x = bench.synthetic_requests(B, g)
lengths = (x != 0).sum(1)
x = x.to(dev)
for _ in range(4):
    tr.episode_step(x, img, tgt, lengths=lengths)
torch.cuda.synchronize()
t0 = time.perf_counter()
host = 0.0
for _ in range(steps):
    h0 = time.perf_counter()
    tr.episode_step(x, img, tgt, lengths=lengths)
    host += time.perf_counter() - h0
torch.cuda.synchronize()
print('ms/step %.2f host enqueue %.2f' % ((time.perf_counter() - t0) / steps * 1e3, host / steps * 1e3))
if tr._step_graphs:
    sg = next(iter(tr._step_graphs.values()))
    torch.cuda.synchronize()
    h0 = time.perf_counter(); sg.graph.replay(); h1 = time.perf_counter(); torch.cuda.synchronize(); h2 = time.perf_counter()
    print('graph: host launch %.2f ms, until done %.2f ms, memset nodes replaced %d' % ((h1 - h0) * 1e3, (h2 - h0) * 1e3, sg.memsets_replaced))
