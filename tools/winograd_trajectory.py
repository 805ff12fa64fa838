"""Do the Winograd layers change training?  The same 40 deterministic (arg-max) episode train steps at bs=64 256x256 with the
direct kernels everywhere (T2O_WINOGRAD=0) and with Winograd F(2x2,3x3) on the 256- / 512-channel layers (default), each in
its own process; prints both loss curves and their deviation.  python tools/winograd_trajectory.py [steps]"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
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
tr = Trainer(model, opt)
g = torch.Generator().manual_seed(10)
B, H, W = 64, 256, 256
out = []
for step in range(int(sys.argv[1])):
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    tgt = (img * 0.8 + 0.1 * torch.rand(B, 3, H, W, generator=g).to(dev)).clamp(0, 1)
    x = bench.synthetic_requests(B, g)
    out.append(float(tr.episode_step(x.to(dev), img, tgt, reinforce_sample=0, lengths=(x != 0).sum(1))))
print(json.dumps(out))
''' % ROOT

steps = sys.argv[1] if len(sys.argv) > 1 else '40'
curves = {}
for name, env in (('direct', {'T2O_WINOGRAD': '0'}), ('winograd', {})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, '-c', WORKER, steps], env=e, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr[-2000:]); sys.exit(1)
    curves[name] = json.loads(r.stdout.strip().splitlines()[-1])
d, w = curves['direct'], curves['winograd']
print('step   direct      winograd    rel.dev')
for i, (a, b) in enumerate(zip(d, w)):
    print('%4d  %.6f   %.6f   %.2e' % (i, a, b, abs(a - b) / abs(a)))
n = len(d)
print('mean loss, last 10 steps: direct %.6f  winograd %.6f' % (sum(d[-10:]) / 10, sum(w[-10:]) / 10))
print('first step relative deviation %.2e; max over the first 5 steps %.2e' % (abs(d[0] - w[0]) / abs(d[0]), max(abs(a - b) / abs(a) for a, b in zip(d[:5], w[:5]))))
