"""Time the actor's image encoder (ResNet-18 variant, fp32) forward+backward under a few
MIOpen/PyTorch settings: which configuration the train step should use."""
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2onet_amd.actor_resnet import ResNet  # noqa: E402

B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda')


def run(tag, benchmark, channels_last):
    torch.backends.cudnn.benchmark = benchmark
    net = ResNet().to(dev).train()
    x = torch.rand(B, 3, S, S, device=dev, requires_grad=True)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
        x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for _ in range(3):
        net(x).sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        net(x).sum().backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    flops = 3 * 4.52e9 * B * (S / 256.0) ** 2
    print('%-28s %7.2f ms  %6.1f TFLOP/s' % (tag, dt * 1e3, flops / dt / 1e12), flush=True)


run('default', False, False)
run('benchmark', True, False)
run('channels_last', False, True)
run('benchmark+channels_last', True, True)
