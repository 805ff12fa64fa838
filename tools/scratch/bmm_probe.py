"""How fast is the library's batched fp32 GEMM at the Winograd F(2x2,3x3) shapes of the 256- and 512-channel stages?"""
import torch, time
dev = torch.device('cuda:0')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (T, C) in ((1024, 512), (4096, 256), (16384, 128), (65536, 64)):
    V = torch.randn(16, T, C, device=dev)
    U = torch.randn(16, C, C, device=dev)          # (xi, co, ci)
    out = torch.empty(16, T, C, device=dev)
    fl = 16 * T * C * C * 2
    us = t(lambda: torch.bmm(V, U.transpose(1, 2), out=out))
    print('bmm NT  T=%6d C=%4d  %7.1f us  %6.1f TF/s' % (T, C, us, fl / us / 1e6))
    Un = U.transpose(1, 2).contiguous()
    us = t(lambda: torch.bmm(V, Un, out=out))
    print('bmm NN  T=%6d C=%4d  %7.1f us  %6.1f TF/s' % (T, C, us, fl / us / 1e6))
    V2 = V.reshape(16 * T, C)
    us = t(lambda: torch.mm(V2, Un[0]))
    print('mm one  M=%6d C=%4d  %7.1f us  %6.1f TF/s' % (16 * T, C, us, fl / us / 1e6))
