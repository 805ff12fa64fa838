import sys, torch
sys.path.insert(0, '.')
from oracle import synth
import t2onet_amd.actor_resnet as R
import t2onet_amd.encoder as E
from tests.test_gpu_encoder import _encoder
DEV='cuda:0'
N,H,W=4,64,256
img=synth.images(N,H,W,31); gout=synth.uniform((N,512),32,-1.0,1.0)
net=_encoder().to(DEV).to(memory_format=torch.channels_last).train()
plan=net.trunk_plan()
x=img.to(DEV).clone().requires_grad_(True)
y=E.trunk_forward(plan, x)
ctx=y.grad_fn
snap=[{k:(v.clone() if torch.is_tensor(v) else v) for k,v in rec.items()} for rec in ctx.saved]
stem_snap=[t.clone() for t in ctx.stem]
torch.cuda.synchronize()
out=net.fc(y.mean((2,3)))
torch.cuda.synchronize()
def check(tag):
    for bi,(rec,s) in enumerate(zip(ctx.saved,snap)):
        for k,v in rec.items():
            if torch.is_tensor(v) and not torch.equal(v, s[k]):
                d=(v-s[k]).abs(); print(tag,'block',bi,k,'changed: n=%d max=%g shape=%s first idx=%s'%(int((d>0).sum()),float(d.max()),tuple(v.shape),(d.flatten()>0).nonzero()[:4].flatten().tolist()))
    for i,(a,b) in enumerate(zip(ctx.stem,stem_snap)):
        if not torch.equal(a,b): print(tag,'stem',i,'changed')
check('after fwd')
out.backward(gout.to(DEV))
torch.cuda.synchronize()
check('after bwd')
print('done')
