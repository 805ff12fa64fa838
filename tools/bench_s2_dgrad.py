"""HIP-event timing of the stride-2 data gradient (t2o_conv3x3s2_dgrad_pre_nhwc) at the four stage entries, bs = 64.
T2O_S2_DGRAD_NARROW=0/1 forces the 8-wave 128x64 / the 4-wave 128x32 tile form.  python tools/bench_s2_dgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import t2onet_amd.functional as T          # noqa: E402
from t2onet_amd import _lib                 # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.load()
N = 64
for Ci, Co, Ho in ((64, 64, 64), (64, 128, 32), (128, 256, 16), (256, 512, 8)):
    dy = torch.rand(N, Ho, Ho, Co, device=dev) - 0.5
    wt = (torch.rand(Ci, 3, 3, Co, device=dev) - 0.5) * 0.05
    dx = torch.empty(N, 2 * Ho, 2 * Ho, Ci, device=dev)
    ws = T._conv_workspace(dev, 64 << 10)
    st = torch.cuda.current_stream().cuda_stream

    def fn():
        _lib.check(lib.t2o_conv3x3s2_dgrad_pre_nhwc(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), ws.data_ptr(), ws.numel(), N, Ho, Ho, Ci, Co, st), 's2 dgrad')
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    flop = 2.0 * 9 * Ci * Co * N * Ho * Ho
    print('Ci=%d Co=%d dy %dx%d: %7.1f us  %6.1f TF/s (%.3f of 157.3)  [T2O_S2_DGRAD_NARROW=%s]' % (
        Ci, Co, Ho, Ho, ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 157.3, os.environ.get('T2O_S2_DGRAD_NARROW', 'default')))
