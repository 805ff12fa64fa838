"""The decoder's M = 64 GEMMs: library default vs torch's TunableOp pick: python tools/scratch/small_gemm_probe.py"""
import torch, time, os
dev = torch.device('cuda:0')
def t(fn, n=40):
    """GPU time per call: the calls are captured in a graph (the host's per-call cost is above the kernels')."""
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 64
shapes = [('lstm0 x@Wih^T', (B, 812), (2048, 812), 'nt'), ('lstm h@Whh^T', (B, 512), (2048, 512), 'nt'), ('linear_out', (B, 1024), (512, 1024), 'nt'),
          ('fc/vis', (B, 512), (512, 512), 'nt'), ('dgates@Wih', (B, 2048), (2048, 812), 'nn'), ('dgates@Whh', (B, 2048), (2048, 512), 'nn'),
          ('dWih += dg^T x', (B, 2048), (B, 812), 'tn'), ('dWhh += dg^T h', (B, 2048), (B, 512), 'tn')]
def build():
    fns = []
    for name, sa, sb, kind in shapes:
        a = torch.randn(*sa, device=dev); b = torch.randn(*sb, device=dev)
        if kind == 'nt': fn = (lambda a=a, b=b: a @ b.t())
        elif kind == 'nn': fn = (lambda a=a, b=b: a @ b)
        else:
            acc = torch.zeros(sa[1], sb[1], device=dev)
            fn = (lambda a=a, b=b, acc=acc: acc.addmm_(a.t(), b))
        fns.append((name, fn))
    return fns
fns = build()
base = [t(fn) for _, fn in fns]
print('default          :', ' '.join('%s %.1f' % (n, v) for (n, _), v in zip(fns, base)))
for lib in ('cublas', 'cublaslt'):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        print('%-17s:' % lib, ' '.join('%.1f' % t(fn) for _, fn in fns))
    except Exception as e:
        print(lib, 'failed', e)
torch.backends.cuda.preferred_blas_library('default')
try:
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(50)
    torch.cuda.tunable.set_filename('/tmp/tunable.csv')
    t0 = time.time()
    for _, fn in fns: fn()
    torch.cuda.synchronize()
    print('tuning took %.1f s' % (time.time() - t0))
    torch.cuda.tunable.tuning_enable(False)
    print('tunable          :', ' '.join('%.1f' % t(fn) for _, fn in fns))
    for r in torch.cuda.tunable.get_results(): print(r)
except Exception as e:
    import traceback; traceback.print_exc()
