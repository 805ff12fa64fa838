import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import t2onet_amd.functional as T
dev = torch.device('cuda:0')
w = (torch.rand(64, 3, 3, 3, device=dev) - 0.5).contiguous(memory_format=torch.channels_last)
dy = (torch.rand(64, 64, 128, 128, device=dev) - 0.5).contiguous(memory_format=torch.channels_last)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('stem data gradient bs=64 256x256: %.1f us' % t(lambda: T.conv3x3s2_dgrad(dy, w)))
