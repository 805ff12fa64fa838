import sys, torch
sys.path.insert(0, '.')
from oracle import synth
import t2onet_amd.actor_resnet as R
import t2onet_amd.encoder as E
import t2onet_amd.functional as T
from tests.test_gpu_encoder import _encoder
DEV='cuda:0'
N,H,W=4,64,256
img=synth.images(N,H,W,31); gout=synth.uniform((N,512),32,-1.0,1.0)
GUARD=2048   # floats
CANARY=-12345.0
allocs=[]
import inspect
class Proxy:
    def __getattr__(self, k): return getattr(torch, k)
    def empty(self, *size, dtype=torch.float32, device=None, **kw):
        if len(size)==1 and isinstance(size[0], (tuple, list, torch.Size)): size=tuple(size[0])
        n=1
        for s in size: n*=int(s)
        es=torch.empty((), dtype=dtype).element_size()
        g=GUARD*4//es
        raw=torch.full((n+2*g,), CANARY if dtype==torch.float32 else 77, dtype=dtype, device=device)
        fr=inspect.stack()[1]
        allocs.append((raw, g, '%s:%d' % (fr.function, fr.lineno)))
        return raw[g:g+n].view(*size) if size else raw[g:g+n].view(())
    def empty_like(self, t, **kw):
        out=self.empty(*t.shape, dtype=t.dtype, device=t.device)
        return out
E.torch=Proxy()
net=_encoder().to(DEV).to(memory_format=torch.channels_last).train()
plan=net.trunk_plan()
x = img.to(DEV).clone().requires_grad_(True)
y=E.trunk_forward(plan, x)
torch.cuda.synchronize()
def check(tag):
    bad=0
    for raw,g,where in allocs:
        c = CANARY if raw.dtype==torch.float32 else 77
        lo=(raw[:g]!=c).nonzero().flatten(); hi=(raw[-g:]!=c).nonzero().flatten()
        if len(lo) or len(hi):
            bad+=1
            print(tag, where, 'numel', raw.numel()-2*g, 'before:', len(lo), lo[:3].tolist(), 'after:', len(hi), hi[:3].tolist(), hi[-3:].tolist() if len(hi) else '')
    print(tag, 'buffers', len(allocs), 'overruns', bad)
check('fwd')
net.fc(y.mean((2,3))).backward(gout.to(DEV))
torch.cuda.synchronize()
check('bwd')
