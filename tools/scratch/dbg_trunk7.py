import sys, torch
sys.path.insert(0, '.')
from oracle import synth
import t2onet_amd.actor_resnet as R
import t2onet_amd.encoder as E
from tests.test_gpu_encoder import _encoder
DEV='cuda:0'
N,H,W=4,64,256
img=synth.images(N,H,W,31); gout=synth.uniform((N,512),32,-1.0,1.0)
def fwd_acts(net, x):
    acts = {}
    h = x
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
            pre = b.bn2(y2) + sc if not x.is_cuda else None
            h = R._bn_relu(b.bn2, y2, sc, False, st)
            acts[k] = dict(y1=y1, a1=a1, y2=y2, out=h, sc=sc, pre=pre)
            k += 1
    return acts
cpu=_encoder().double().train()
with torch.no_grad():
    ref=fwd_acts(cpu, img.double())
net=_encoder().to(DEV).to(memory_format=torch.channels_last).train()
plan=net.trunk_plan()
x = img.to(DEV).clone().requires_grad_(True)
y=E.trunk_forward(plan, x)
ctx=y.grad_fn
net.fc(y.mean((2,3))).backward(gout.to(DEV))
torch.cuda.synchronize()
for bi in range(8):
    rec=ctx.saved[bi]
    for k in ('y1','a1','y2','out'):
        got=rec[k].permute(0,3,1,2).double().cpu(); r=ref[bi][k]
        diff=(got-r).abs()
        mm=int(((got>0)!=(r>0)).sum())
        print('block %d %-3s max|diff| %.2e  mask mismatches %d of %d   min|ref pre| %s' % (bi,k,float(diff.max()),mm,r.numel(), ('%.2e'%float(ref[bi]['pre'].abs().min())) if k=='out' else ''))
