"""HIP-event timing of the on-chip Winograd weight gradient (t2o_wino_fused_wgrad_nhwc) against the direct kernel it replaces, at
the train step's shapes (five encoder passes of bs = 64 side by side).  python tools/bench_wino_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import t2onet_amd.functional as T          # noqa: E402
from t2onet_amd import _lib                 # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.load()
for N, C, H in ((320, 64, 64), (320, 128, 32), (64, 64, 64), (320, 256, 16)):
    x = torch.rand(N, H, H, C, device=dev) - 0.5
    dy = torch.rand(N, H, H, C, device=dev) - 0.5
    dw = torch.zeros(C, 3, 3, C, device=dev)
    need = lib.t2o_conv3x3_wgrad_workspace_bytes(N, H, H, C, C)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def direct():
        _lib.check(lib.t2o_conv3x3_wgrad_acc_nhwc(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), need, N, H, H, C, C, 1, 1, st), 'direct')

    def wino():
        assert T.wino_fused_wgrad_nhwc(x, dy, dw, N, H, H, True)
    flop = 2.0 * 9 * C * C * N * H * H
    for name, fn in (('direct', direct), ('on-chip winograd', wino)):
        if name == 'direct' and need == 0:
            continue
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        ex = flop * (16.0 / 36.0 if name != 'direct' else 1.0)
        print('N=%d C=%d %dx%d %-18s %8.1f us  %6.1f TF/s algorithmic  %6.1f TF/s executed (%.3f of 157.3)' % (
            N, C, H, H, name, ms * 1e3, flop / ms / 1e9, ex / ms / 1e9, ex / ms / 1e9 / 157.3))
