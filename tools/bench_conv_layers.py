"""Per-layer timing of the image encoder's convolutions as PyTorch-ROCm runs them (MIOpen):
forward, data gradient, weight gradient, NCHW and channels_last, bs=64 at 256x256 input.
Tells which layer shapes a hand-written MFMA kernel has to beat, and by how much.

    python tools/bench_conv_layers.py [B] [S]
"""
import sys
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda')
# (name, Cin, Cout, k, stride, Hin, count per encoder pass)
LAYERS = [('stem', 3, 64, 3, 2, S, 1),
          ('l1.s2', 64, 64, 3, 2, S // 2, 1), ('l1.sc', 64, 64, 1, 2, S // 2, 1), ('l1.s1', 64, 64, 3, 1, S // 4, 3),
          ('l2.s2', 64, 128, 3, 2, S // 4, 1), ('l2.sc', 64, 128, 1, 2, S // 4, 1), ('l2.s1', 128, 128, 3, 1, S // 8, 3),
          ('l3.s2', 128, 256, 3, 2, S // 8, 1), ('l3.sc', 128, 256, 1, 2, S // 8, 1), ('l3.s1', 256, 256, 3, 1, S // 16, 3),
          ('l4.s2', 256, 512, 3, 2, S // 16, 1), ('l4.sc', 256, 512, 1, 2, S // 16, 1), ('l4.s1', 512, 512, 3, 1, S // 32, 3)]


def timeit(fn, n=10):
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


tot = {}
print('%-7s %-5s %9s | %21s | %21s | %21s' % ('layer', 'fmt', 'GFLOP', 'fwd ms (TF/s)', 'dgrad ms (TF/s)', 'wgrad ms (TF/s)'))
for name, ci, co, k, s, h, cnt in LAYERS:
    pad = k // 2
    ho = (h + 2 * pad - k) // s + 1
    gf = 2.0 * B * ho * ho * co * ci * k * k / 1e9
    for fmt in ('nchw', 'nhwc'):
        mf = torch.channels_last if fmt == 'nhwc' else torch.contiguous_format
        x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=mf)
        w = torch.randn(co, ci, k, k, device=dev).contiguous(memory_format=mf)
        y = torch.nn.functional.conv2d(x, w, None, s, pad)
        gy = torch.randn_like(y).contiguous(memory_format=mf)
        t_f = timeit(lambda: torch.nn.functional.conv2d(x, w, None, s, pad))
        bw = lambda mask: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, mask)
        t_d = timeit(lambda: bw([True, False, False])) if ci > 3 else float('nan')
        t_w = timeit(lambda: bw([False, True, False]))
        print('%-7s %-5s %9.2f | %9.3f (%7.1f) | %9.3f (%7.1f) | %9.3f (%7.1f)   x%d' % (
            name, fmt, gf, t_f, gf / t_f, t_d, gf / t_d, t_w, gf / t_w, cnt), flush=True)
        a = tot.setdefault(fmt, [0.0, 0.0, 0.0, 0.0])
        a[0] += cnt * t_f
        a[1] += cnt * (0 if t_d != t_d else t_d)
        a[2] += cnt * t_w
        a[3] += cnt * gf
for fmt, (f, d, w, gf) in tot.items():
    print('%s per encoder pass: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms, total %.2f ms for %.1f GFLOP x3 -> %.1f TF/s' % (
        fmt, f, d, w, f + d + w, gf, 3 * gf / (f + d + w)))
