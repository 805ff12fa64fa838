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
    grads = {}
    x = x.clone().requires_grad_(True)
    h = x.contiguous(memory_format=torch.channels_last) if x.is_cuda else x
    y, st = R._conv(net.conv1, h, net.bn1)
    h = R._bn_relu(net.bn1, y, None, False, st)
    k = 0
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        for b in layer:
            xin = h
            y1, st = R._conv(b.conv1, xin, b.bn1)
            a1 = R._bn_relu(b.bn1, y1, None, False, st)
            sc = R._bn_plain(b.shortcut[1], b.shortcut[0](xin), False) if len(b.shortcut) else xin
            y2, st = R._conv(b.conv2, a1, b.bn2)
            h = R._bn_relu(b.bn2, y2, sc, False, st)
            for nm, t in (('dy2', y2), ('da1', a1), ('dsc', sc), ('d', h)):
                if t.requires_grad: t.register_hook(lambda g, k=k, nm=nm: grads.__setitem__('%s_%d' % (nm, k), g.clone()))
            k += 1
    out = net.fc(h.mean((2, 3)))
    out.backward(gout)
    return grads

R._TRUNK = False
net=_encoder().to(DEV).to(memory_format=torch.channels_last).train()
gl=run_layers(net, img.to(DEV), gout.to(DEV))
cpu=_encoder().double().train()
ref=run_layers(cpu, img.double(), gout.double())
for k in (7,6,5,4,3):
    print('layers', k, ' '.join('%s %.1e' % (nm, rel(gl['%s_%d'%(nm,k)], ref['%s_%d'%(nm,k)])) for nm in ('d','dy2','dsc','da1') if '%s_%d'%(nm,k) in gl))
E._DEBUG = []
R._TRUNK = True
net=_encoder().to(DEV).to(memory_format=torch.channels_last).train()
x = img.to(DEV).clone().requires_grad_(True)
net(x).backward(gout.to(DEV))
k = 8
for item in E._DEBUG:
    if torch.is_tensor(item):
        k -= 1
        if k >= 0: print('trunk', k, 'd %.1e' % rel(item.permute(0,3,1,2), ref['d_%d' % k]), end=' ')
    else:
        print(' '.join('%s %.1e' % (item[i], rel(item[i+1].permute(0,3,1,2), ref['%s_%d' % (item[i], k)])) for i in (0,2,4)))
