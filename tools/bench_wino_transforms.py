"""Stand-alone timings of the Winograd transform kernels at the two encoder stages that use them (bs=64):
python tools/bench_wino_transforms.py"""
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
for (N, H, C) in ((64, 8, 512), (64, 16, 256)):
    x = torch.randn(N, H, H, C, device=dev)
    st = T._stream(dev)
    Tp = lib.t2o_wino_padded_tiles(N, H, H)
    V = torch.empty(16, Tp, C, device=dev); Ad = torch.empty(16, Tp, C, device=dev)
    M = torch.randn(16, N * H * H // 4, C, device=dev); y = torch.empty_like(x)
    stats = torch.empty(lib.t2o_wino_stats_rows(N, H, H, C), 2, C, device=dev)
    dU = torch.randn(4, 16, C, C, device=dev); dw = torch.zeros(C, 3, 3, C, device=dev)
    print('C=%d %dx%d: input %.1f us, dy_both %.1f us, output %.1f us, output+stats %.1f us, dw(splits 1) %.1f us, dw(splits 4) %.1f us' % (C, H, H,
          t(lambda: lib.t2o_wino_input_transform(x.data_ptr(), V.data_ptr(), N, H, H, C, st)),
          t(lambda: lib.t2o_wino_dy_transforms(x.data_ptr(), V.data_ptr(), Ad.data_ptr(), N, H, H, C, st)),
          t(lambda: lib.t2o_wino_output_transform(M.data_ptr(), None, y.data_ptr(), None, N, H, H, C, st)),
          t(lambda: lib.t2o_wino_output_transform(M.data_ptr(), None, y.data_ptr(), stats.data_ptr(), N, H, H, C, st)),
          t(lambda: lib.t2o_wino_dw_transform(dU.data_ptr(), dw.data_ptr(), C, C, 1, 1, st)),
          t(lambda: lib.t2o_wino_dw_transform(dU.data_ptr(), dw.data_ptr(), C, C, 4, 1, st))))
