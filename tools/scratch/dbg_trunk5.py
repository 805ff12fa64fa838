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
E._DEBUG = []
net=_encoder().to(DEV).to(memory_format=torch.channels_last).train()
plan=net.trunk_plan()
x = img.to(DEV).clone().requires_grad_(True)
y=E.trunk_forward(plan, x)
ctx=y.grad_fn
net.fc(y.mean((2,3))).backward(gout.to(DEV))
dbg=E._DEBUG
# entries: d7,(blk7), d6,(blk6), d5,(..), d4,(dy2,dsc,da1) ...
ds=[t for t in dbg if torch.is_tensor(t)]; inner=[t for t in dbg if not torch.is_tensor(t)]
for bi in (5, 4):
    rec=ctx.saved[bi]; b=plan.blocks[bi]
    d=ds[7-bi].double().cpu(); dy2=inner[7-bi][1].double().cpu(); dsc=inner[7-bi][3].double().cpu()
    y2=rec['y2'].double().cpu(); out=rec['out'].double().cpu(); m=rec['m2'].double().cpu(); i=rec['i2'].double().cpu()
    C=y2.shape[-1]; Y=y2.reshape(-1,C); M=Y.shape[0]
    mean=Y.mean(0); var=Y.var(0,unbiased=False); invstd=1/torch.sqrt(var+b.bn2.eps)
    print('block',bi,'mean err %.1e invstd err %.1e'%(rel(m,mean), rel(i,invstd)))
    w=b.bn2.weight.double().cpu(); bias=b.bn2.bias.double().cpu()
    sc_in = (rec['x'] if not len(b.shortcut) else None)
    g=(d.reshape(-1,C))*(out.reshape(-1,C)>0)
    xhat=(Y-m)*i
    dx=(w*i)*(g-g.mean(0)-xhat*(g*xhat).mean(0))
    print('   kernel dy2 vs formula(saved stats) %.1e   dsc vs g %.1e   frac out>0 %.3f  frac |pre|<1e-6: %.4f' % (rel(dy2.reshape(-1,C),dx), rel(dsc.reshape(-1,C),g), float((out>0).double().mean()), 0.0))
