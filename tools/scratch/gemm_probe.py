import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import t2onet_amd.functional as T
dev = torch.device('cuda:0')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (Tt, C) in ((1024, 512), (4096, 256), (16384, 128)):
    V = torch.randn(16, Tt, C, device=dev); U = torch.randn(16, C, C, device=dev)
    fl = 16 * Tt * C * C * 2
    a = t(lambda: T.gemm_nt_batched(V, U)); b = t(lambda: torch.bmm(V, U.transpose(1, 2)))
    print('T=%6d C=%4d  own %7.1f us %6.1f TF/s   library %7.1f us %6.1f TF/s' % (Tt, C, a, fl / a / 1e6, b, fl / b / 1e6))
for (Tt, C) in ((1024, 512), (4096, 256)):
    Ad = torch.randn(16, Tt, C, device=dev); V = torch.randn(16, Tt, C, device=dev)
    fl = 16 * Tt * C * C * 2
    a = t(lambda: T.gemm_tn_batched(Ad, V)); b = t(lambda: torch.bmm(Ad.transpose(1, 2), V))
    print('TN T=%6d C=%4d  own %7.1f us %6.1f TF/s   library %7.1f us %6.1f TF/s' % (Tt, C, a, fl / a / 1e6, b, fl / b / 1e6))
