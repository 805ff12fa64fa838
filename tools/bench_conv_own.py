"""Event timings of the encoder's data-gradient kernels at the bs = 64, 256 x 256 stage shapes: the stride-2 data gradient
of the four stage-entry convolutions, and the stride-1 data gradient with and without the fused addend."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import t2onet_amd.functional as T

dev = torch.device('cuda:0')
B = 64


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for name, ci, co, ho in (('l1.0.conv1', 64, 64, 64), ('l2.0.conv1', 64, 128, 32), ('l3.0.conv1', 128, 256, 16), ('l4.0.conv1', 256, 512, 8)):
    dy = cl(torch.randn(B, co, ho, ho, device=dev))
    w = cl(torch.randn(co, ci, 3, 3, device=dev))
    us = timeit(lambda: T.conv3x3s2_dgrad(dy, w))
    gf = 2.0 * B * ho * ho * co * ci * 9 / 1e9
    print('s2 dgrad %-11s %7.1f us  %6.1f TF/s' % (name, us, gf / us * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / (us * 1e-6) / 1e3))
for c, h in ((64, 64), (128, 32)):
    dy = cl(torch.randn(B, c, h, h, device=dev))
    w = cl(torch.randn(c, c, 3, 3, device=dev))
    wt = T.conv_weight_transform(w, 9, True)
    add = cl(torch.randn(B, c, h, h, device=dev))
    a = timeit(lambda: T.conv3x3_dgrad_pre(dy, wt, c))
    b = timeit(lambda: T.conv3x3_dgrad_pre(dy, wt, c, addend=add))
    print('s1 dgrad c=%d: %7.1f us, with addend %7.1f us' % (c, a, b))
