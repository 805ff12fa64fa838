import sys, torch
sys.path.insert(0, '.')
from oracle import synth
import t2onet_amd.actor_resnet as R
from tests.test_gpu_encoder import _encoder, _run
DEV='cuda:0'
N,H,W=4,64,256
img=synth.images(N,H,W,31); gout=synth.uniform((N,512),32,-1.0,1.0)
def rel(a,b): return float((a.cpu().double()-b).norm()/b.norm())
res={}
for rep in range(2):
  for trunk in (True, False):
    R._TRUNK=trunk
    net=_encoder().to(DEV).to(memory_format=torch.channels_last).train()
    res[(rep,trunk)]=_run(net,img.to(DEV),gout.to(DEV))
cpu=_encoder().double().train()
ref=_run(cpu,img.double(),gout.double())
for k,v in res.items():
    bad=[(n, '%.1e' % rel(v[2][n], ref[2][n])) for n in v[2] if rel(v[2][n], ref[2][n]) > 1e-4]
    print(k, 'out %.1e dimg %.1e' % (rel(v[0],ref[0]), rel(v[1],ref[1])), 'bad params:', [b[0] for b in bad], len(bad))
