"""Own MFMA convolution kernels vs MIOpen's for the encoder's 3x3 stride-1 layers (bs=64, 256x256 input):
forward, data gradient, weight gradient (f, d, w), and the data gradient of the stride-2 layers (s).
usage: tools/bench_conv.py [batch] [directions, e.g. fd]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import t2onet_amd.functional as T  # noqa: E402

dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DIRS = sys.argv[2] if len(sys.argv) > 2 else 'fdws'


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def lib_bwd(gy, x, w, mask):
    return torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, mask)


for name, c, h in (('l1.s1', 64, 64), ('l2.s1', 128, 32), ('l3.s1', 256, 16), ('l4.s1', 512, 8)):
    x = torch.randn(B, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, c, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    gf = 2.0 * B * h * h * c * c * 9 / 1e9
    cases = {
        'f': ('fwd  ', lambda: torch.nn.functional.conv2d(x, w, None, 1, 1), lambda: T.conv3x3_forward(x, w)),
        'd': ('dgrad', lambda: lib_bwd(gy, x, w, [True, False, False])[0], lambda: T.conv3x3_dgrad(gy, w)),
        'w': ('wgrad', lambda: lib_bwd(gy, x, w, [False, True, False])[1], lambda: T.conv3x3_wgrad(x, gy)),
    }
    for d in DIRS:
        if d not in cases:
            continue
        label, lib, own = cases[d]
        t_lib, t_own = timeit(lib), timeit(own)
        ref = lib()
        err = ((own() - ref).abs().max() / ref.abs().max()).item()
        print('%-6s %s %6.2f GFLOP  MIOpen %.3f ms (%6.1f TF/s)   own %.3f ms (%6.1f TF/s)   x%.2f   rel err vs MIOpen %.1e' % (
            name, label, gf, t_lib, gf / t_lib, t_own, gf / t_own, t_lib / t_own, err), flush=True)

if 's' in DIRS:
    # the stride-2 layers (first convolution of each stage): data gradient
    for name, ci, co, ho in (('stem', 3, 64, 128), ('l1.s2', 64, 64, 64), ('l2.s2', 64, 128, 32), ('l3.s2', 128, 256, 16), ('l4.s2', 256, 512, 8)):
        x = torch.randn(B, ci, 2 * ho, 2 * ho, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, co, ho, ho, device=dev).contiguous(memory_format=torch.channels_last)
        gf = 2.0 * B * ho * ho * ci * co * 9 / 1e9
        lib = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        own = lambda: T.conv3x3s2_dgrad(gy, w)
        t_lib, t_own = timeit(lib), timeit(own)
        ref = lib()
        err = ((own() - ref).abs().max() / ref.abs().max()).item()
        print('%-6s dgrad %6.2f GFLOP  MIOpen %.3f ms (%6.1f TF/s)   own %.3f ms (%6.1f TF/s)   x%.2f   rel err vs MIOpen %.1e' % (
            name, gf, t_lib, gf / t_lib, t_own, gf / t_own, t_lib / t_own, err), flush=True)
        if ci % 64 == 0:
            lib = lambda: torch.nn.functional.conv2d(x, w, None, 2, 1)
            own = lambda: T.conv3x3s2_forward(x, w)
            t_lib, t_own = timeit(lib), timeit(own)
            ref = lib()
            err = ((own() - ref).abs().max() / ref.abs().max()).item()
            print('%-6s fwd   %6.2f GFLOP  MIOpen %.3f ms (%6.1f TF/s)   own %.3f ms (%6.1f TF/s)   x%.2f   rel err vs MIOpen %.1e' % (
                name, gf, t_lib, gf / t_lib, t_own, gf / t_own, t_lib / t_own, err), flush=True)
            lib = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
            own = lambda: T.conv3x3s2_wgrad(x, gy)
            t_lib, t_own = timeit(lib), timeit(own)
            ref = lib()
            err = ((own() - ref).abs().max() / ref.abs().max()).item()
            print('%-6s wgrad %6.2f GFLOP  MIOpen %.3f ms (%6.1f TF/s)   own %.3f ms (%6.1f TF/s)   x%.2f   rel err vs MIOpen %.1e' % (
                name, gf, t_lib, gf / t_lib, t_own, gf / t_own, t_lib / t_own, err), flush=True)
