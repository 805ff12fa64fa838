"""Stand-alone timings of the 1x1 stride-2 shortcut kernels at the four encoder stages (bs=64, 256x256 image): python tools/bench_shortcut.py"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import t2onet_amd.functional as T
from t2onet_amd import _lib
dev = torch.device('cuda:0')
lib = _lib.load()
def t(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
st = T._stream(dev)
for (Ci, Co, H) in ((64, 64, 128), (64, 128, 64), (128, 256, 32), (256, 512, 16)):
    N = 64
    x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, Ci, device=dev); wt = w.t().contiguous()
    y = torch.empty(N, H // 2, H // 2, Co, device=dev); dx = torch.zeros_like(x)
    a = t(lambda: lib.t2o_conv1x1s2_fwd_nhwc(x.data_ptr(), w.data_ptr(), y.data_ptr(), N, H, H, Ci, Co, st))
    b = t(lambda: lib.t2o_conv1x1s2_dgrad_acc_nhwc(y.data_ptr(), wt.data_ptr(), dx.data_ptr(), N, H, H, Ci, Co, st))
    print('%3d -> %3d at %3dx%3d: forward %.1f us, data gradient (scatter-add) %.1f us' % (Ci, Co, H, H, a, b))
