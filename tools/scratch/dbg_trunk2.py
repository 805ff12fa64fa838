import sys, torch
sys.path.insert(0, '.')
from oracle import synth
import t2onet_amd.actor_resnet as R
import t2onet_amd.encoder as E
from tests.test_gpu_encoder import _encoder
DEV='cuda:0'
N,H,W=4,64,256
img=synth.images(N,H,W,31); gout=synth.uniform((N,512),32,-1.0,1.0)
def rel(a,b): return float((a.cpu().double()-b.cpu().double()).norm()/b.cpu().double().norm())

def run_layers(net, x, gout):
    """per-layer path with hooks on every block output gradient"""
    grads = {}
    x = x.clone().requires_grad_(True)
    h = x.contiguous(memory_format=torch.channels_last)
    y, st = R._conv(net.conv1, h, net.bn1)
    h = R._bn_relu(net.bn1, y, None, False, st)
    h.register_hook(lambda g: grads.__setitem__('stem', g.clone()))
    k = 0
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        for b in layer:
            h = b(h)
            h.register_hook(lambda g, k=k: grads.__setitem__('b%d' % k, g.clone()))
            k += 1
    out = net.fc(h.mean((2, 3)))
    out.backward(gout)
    grads['dimg'] = x.grad
    return grads

R._TRUNK = False
res = []
for rep in range(3):
    net=_encoder().to(DEV).to(memory_format=torch.channels_last).train()
    res.append(run_layers(net, img.to(DEV), gout.to(DEV)))
cpu=_encoder().double().train()
ref=run_layers(cpu, img.double(), gout.double())
for rep, g in enumerate(res):
    print(rep, ' '.join('%s %.1e' % (k, rel(g[k], ref[k])) for k in ['b7','b6','b5','b4','b3','b2','b1','b0','stem','dimg']))
# the trunk: stash d per block
E._DEBUG = []
R._TRUNK = True
net=_encoder().to(DEV).to(memory_format=torch.channels_last).train()
x = img.to(DEV).clone().requires_grad_(True)
net(x).backward(gout.to(DEV))
names = ['b7','b6','b5','b4','b3','b2','b1','b0','stem']
print('trunk', ' '.join('%s %.1e' % (k, rel(d.permute(0,3,1,2), ref[k])) for k, d in zip(names, E._DEBUG)), 'dimg %.1e' % rel(x.grad, ref['dimg']))
