import os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2onet_amd.actor_resnet import ResNet
from t2onet_amd.train import FlatGradients
from t2onet_amd.graphs import GraphedEncoder
dev = torch.device('cuda:0')
torch.manual_seed(0)
base = ResNet().to(dev).train()
if 'nhwc' in sys.argv:
    base = base.to(memory_format=torch.channels_last)
B, S, calls = 4, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 3
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # eager calls after the graphed ones
imgs = [torch.rand(B, 3, S, S, device=dev) for _ in range(calls + extra)]
ws = [torch.randn(B, 512, device=dev) for _ in range(calls + extra)]

def run(graph):
    m = copy.deepcopy(base)
    fg = FlatGradients(m.parameters())
    g = GraphedEncoder(m, imgs[0], calls) if graph else None
    print('after capture flat max', float(fg.flat.abs().max()))
    for rep in range(2):
        fg.zero()
        outs = []
        for k in range(calls + extra):
            x = imgs[k].clone().requires_grad_(k > 0)
            outs.append(((g(x, k) if graph and k < calls else m(x)) * ws[k]).sum())
        loss = sum(outs)
        loss.backward()
    return loss.item(), fg.flat.clone(), {n: p.grad.clone() for n, p in m.named_parameters()}

l0, f0, g0 = run(False)
l1, f1, g1 = run(True)
print('loss', l0, l1)
for n in g0:
    d = (g0[n] - g1[n]).abs().max().item(); s = g0[n].abs().max().item()
    if d > 1e-4 * s or d != d:
        print('%-40s max %.3e diff %.3e ratio %.3f' % (n, s, d, (g1[n].flatten()[0] / g0[n].flatten()[0]).item()))
print('flat rel diff', ((f0 - f1).abs().max() / f0.abs().max()).item())
