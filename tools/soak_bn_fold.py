"""Long comparison of the folded batch-norm finalize (t2o_bn_set_sync_region: the finalize step inside the apply kernels,
agent-scope release / acquire across the XCDs) with the separate finalize launches: TWO trainers from the same
initialisation step through the same batches with the same random streams, one in each mode; after every step the losses must be
the same BITS and, every `window` steps, so must a checksum over all parameters.  A stale coefficient read once in 3,000 steps x
170 batch-norm calls would show as a first differing step.   python tools/soak_bn_fold.py [steps=3000] [window=250] [size=256] [control 0|1]
(control = 1: BOTH trainers with the separate launches -- that the comparison itself is bit-stable.  The framework's index_add_ /
embedding gradients use atomics by default; the run asks for its deterministic algorithms.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import t2onet_amd
import t2onet_amd.functional as T
from t2onet_amd.actor import Actor
from t2onet_amd.train import Trainer
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
window = int(sys.argv[2]) if len(sys.argv) > 2 else 250
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
control = len(sys.argv) > 4 and sys.argv[4] != '0'
torch.use_deterministic_algorithms(True, warn_only=True)
dev = torch.device('cuda:0')
opt = t2onet_amd.default_options()


def make():
    torch.manual_seed(10)
    m = Actor(opt).to(dev).train()
    m.use_channels_last()
    return m, Trainer(m, opt)


(m0, t0), (m1, t1) = make(), make()
g = torch.Generator().manual_seed(10)
B = 64
batches = []
for _ in range(3):
    img = torch.rand(B, 3, size, size, generator=g).to(dev)
    tgt = torch.rand(B, 3, size, size, generator=g).to(dev)
    x = bench.synthetic_requests(B, g)
    batches.append((x.to(dev), img, tgt, (x != 0).sum(1)))


def checksum(model):
    return float(sum(p.detach().double().sum() for p in model.parameters())), float(sum(p.detach().double().abs().sum() for p in model.parameters()))


bad = None
tic = time.perf_counter()
for s in range(steps):
    x, img, tgt, lengths = batches[s % 3]
    losses = []
    for fused, tr in ((False, t0), (not control, t1)):
        T.bn_fused_finalize(dev, fused)
        torch.manual_seed(1000 + s)                           # the same operator draws and dropout masks in both
        losses.append(tr.episode_step(x, img, tgt, lengths=lengths))
    a, b = float(losses[0]), float(losses[1])
    if a != b and bad is None:
        bad = (s, a, b)
        print('step %d: loss %r (separate finalize) vs %r (folded): DIFFERENT' % (s, a, b), flush=True)
        break
    if (s + 1) % window == 0:
        c0, c1 = checksum(m0), checksum(m1)
        same = c0 == c1
        print('steps %5d  loss %.9f  parameter checksums %s  (%.1f s)' % (s + 1, a, 'equal %r' % (c0,) if same else 'DIFFERENT %r %r' % (c0, c1),
                                                                        time.perf_counter() - tic), flush=True)
        if not same and bad is None:
            bad = (s, c0, c1)
            break
T.bn_fused_finalize(dev, None)
blk = T._bn_sync[0][0]
torch.cuda.synchronize()
print('counter block after the run:', blk.tolist()[:2])
print('mode: %s' % ('CONTROL (separate finalize in both)' if control else 'separate vs folded'))
print('RESULT: %s' % ('bit-identical over %d steps' % steps if bad is None else 'DIFFERENT at %r' % (bad,)))
sys.exit(0 if bad is None else 1)
