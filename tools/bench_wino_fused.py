"""Event timings of the on-chip Winograd kernel against the direct forward kernel at the bs = 64 stage shapes."""
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


for c, h in ((64, 64), (128, 32), (256, 16)):
    x = torch.randn(B, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(c, c, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    xh = x.permute(0, 2, 3, 1)
    Uc = T.wino_u_chunked(T.wino_weight(w.permute(0, 2, 3, 1).contiguous(), c, c))
    out = torch.empty(B, h, h, c, device=dev)
    d = timeit(lambda: T.conv3x3_forward(x, w, want_stats=True))
    if c >= 256:
        U = T.wino_weight(w.permute(0, 2, 3, 1).contiguous(), c, c)
        xc = xh.contiguous()
        sp = timeit(lambda: T.wino_conv_nhwc(xc, U, B, h, h, None, True))
        print('   separate-pass Winograd pipeline %.1f us (input transform + 16 GEMMs + output transform)' % sp)
    f = timeit(lambda: T.wino_fused_conv_nhwc(xh, Uc, B, h, h, None, True, out=out))
    f2 = timeit(lambda: T.wino_fused_conv_nhwc(xh, Uc, B, h, h, None, False, out=out))
    ref = torch.nn.functional.conv2d(x, w, None, 1, 1)
    err = float((out.permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max())
    print('c=%d %dx%d: direct %.1f us, on-chip winograd %.1f us (no stats %.1f), max err vs library fp32 %.2e' % (c, h, h, d, f, f2, err))
