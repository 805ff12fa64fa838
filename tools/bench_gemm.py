"""t2o_gemm on the train step's own products against the framework's BLAS (HIP-event timed, groups of 20 launches):
python tools/bench_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import t2onet_amd.functional as T

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
# (what, M, N, K, a_kmajor, b_kmajor)
CASES = [('enc fwd l0  x W^T', 1088, 2048, 300, False, False), ('enc fwd l1  x W^T', 1088, 2048, 512, False, False),
         ('enc dx  l1  dgi W', 1088, 512, 2048, False, True), ('enc dx  l0  dgi W', 1088, 300, 2048, False, True),
         ('enc dWih l0 dgi^T x', 1024, 300, 1088, True, True), ('enc dWih l1', 1024, 512, 1088, True, True),
         ('enc dWhh', 1024, 256, 1024, True, True),
         ('tape W_ih_l0', 2048, 812, 320, True, True), ('tape W_hh', 2048, 512, 320, True, True),
         ('tape vis_linear', 512, 512, 320, True, True), ('tape linear_out half', 512, 512, 320, True, True),
         ('tape out_linear', 11, 512, 320, True, True), ('tape fc', 512, 512, 320, True, True)]


def timed(fn, reps=5, group=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(group):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / group * 1e3)
    return sorted(ts)[len(ts) // 2]


tot_own = tot_lib = 0.0
for what, M, N, K, ak, bk in CASES:
    A = (torch.rand((K, M) if ak else (M, K), generator=g) - 0.5).to(dev)
    B = (torch.rand((K, N) if bk else (N, K), generator=g) - 0.5).to(dev)
    out = torch.zeros(M, N, device=dev)
    own = timed(lambda: T.gemm(A, B, out=out, a_kmajor=ak, b_kmajor=bk, accumulate=True))
    Aop, Bop = (A.t() if ak else A), (B if bk else B.t())
    lib = timed(lambda: torch.addmm(out, Aop, Bop, out=out))
    tot_own += own
    tot_lib += lib
    print('%-24s M %5d N %5d K %5d  own %7.1f us (%5.1f TF/s)   library %7.1f us' % (what, M, N, K, own, 2.0 * M * N * K / own / 1e6, lib))
print('sum: own %.1f us, library %.1f us' % (tot_own, tot_lib))
X = (torch.rand(320, 2048, generator=g) - 0.5).to(dev)
o = torch.zeros(2048, device=dev)
print('colsum 320 x 2048: %.1f us' % timed(lambda: T.colsum(X, out=o, accumulate=True)))
