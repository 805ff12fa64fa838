"""Would the on-chip Winograd kernel pay on the small maps (8 x 8 maps as 4 x 4 tiles x 4 images per
workgroup)?  Measured without writing the kernel form: t2o_wino_fused_conv_nhwc on a batch of N / 4 images of 16 x 16 has EXACTLY
the workgroup count, the chunks per workgroup and the FLOP the 4-images-per-workgroup form would have on N images of 8 x 8 (its
patch DMA would fetch 400 pixel slots per chunk instead of 324: slightly more) -- against the separate-pass pipeline on the real
shape.  python tools/bench_small_maps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import t2onet_amd.functional as T

dev = torch.device('cuda:0')


def timeit(fn, n=20, reps=5):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


# (what, channels, real (N, H), stand-in (N, H) for the on-chip kernel)
for what, c, (n, h), (n2, h2) in (('512 ch, 8x8 maps, bs 64 (the 256x256 step)', 512, (64, 8), (16, 16)),
                                  ('256 ch, 8x8 maps, bs 64 (the 128x128 step)', 256, (64, 8), (16, 16)),
                                  ('512 ch, 4x4 maps, bs 64 (the 128x128 step)', 512, (64, 4), (4, 16))):
    w = (torch.randn(c, 3, 3, c, device=dev) * 0.05)
    U = T.wino_weight(w, c, c)
    Uc = T.wino_u_chunked(U)
    x = torch.randn(n, h, h, c, device=dev)
    sep = timeit(lambda: T.wino_conv_nhwc(x, U, n, h, h, None, True))
    x2 = torch.randn(n2, h2, h2, c, device=dev)
    out = torch.empty_like(x2)
    on = timeit(lambda: T.wino_fused_conv_nhwc(x2, Uc, n2, h2, h2, None, True, out=out))
    wgs = n2 * (h2 // 16) ** 2 * (c // 64)
    print('%-46s separate passes %6.1f us   on-chip kernel on the equivalent problem (%d workgroups x %d chunks) %6.1f us'
          % (what, sep, wgs, c // 8, on))
