"""The batch norms' finalize step inside their apply kernels (t2o_bn_set_sync_region, t2o_norm.hip fin_head / fin_tail: the first
C/4 workgroups of the apply kernel reduce the partial rows, publish the coefficients with an agent-scope release, everybody
acquires them -- across the eight XCDs' L2s) against the separate finalize launch: the SAME bits in every output (normalised
activations, saved statistics, running statistics, input and parameter gradients) over more than 2,000 randomised launches of
every entry point that has the fold, the counter block zero again after every call (a wrong hand-over would be
silently stale statistics)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Mode:
    """with _Mode(fused): the library's fold on / off for cuda:0; .counters = the registered block"""

    def __init__(self, fused):
        self.fused = fused

    def __enter__(self):
        import t2onet_amd.functional as T
        T.bn_fused_finalize(DEV, self.fused)
        self.counters = T._bn_sync[0][0]
        return self

    def __exit__(self, *exc):
        import t2onet_amd.functional as T
        T.bn_fused_finalize(DEV, None)                     # back to the module default


def _fwd(lib, x, res, w, b, relu, partial):
    M, C = x.shape
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, invstd, out = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty_like(x)
    ws = torch.empty(lib.t2o_bn_nhwc_workspace_bytes(M, C), dtype=torch.uint8, device=DEV)
    if partial is None:
        rc = lib.t2o_bn_relu_nhwc_fwd(_p(x), _p(res), _p(w), _p(b), _p(rm), _p(rv), _p(mean), _p(invstd), _p(out), 0.1, 1e-5, relu,
                                      _p(ws), ws.numel(), M, C, _st())
    else:
        rc = lib.t2o_bn_relu_nhwc_fwd_partials(_p(x), _p(res), _p(w), _p(b), _p(rm), _p(rv), _p(mean), _p(invstd), _p(out), 0.1, 1e-5,
                                               relu, _p(partial), partial.shape[0], _p(ws), ws.numel(), M, C, _st())
    assert rc == 0
    return out, mean, invstd, rm, rv


def _bwd(lib, x, y, dy, w, b, mean, invstd, has_res, relu, partial):
    M, C = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if has_res else None
    dw, db = torch.full((C,), 0.25, device=DEV), torch.full((C,), -0.5, device=DEV)         # (accumulate form: added to)
    ws = torch.empty(lib.t2o_bn_nhwc_workspace_bytes(M, C), dtype=torch.uint8, device=DEV)
    if partial is None:
        rc = lib.t2o_bn_relu_nhwc_bwd_acc(_p(x), _p(y), _p(dy), _p(w), _p(b), _p(mean), _p(invstd), _p(dx), _p(dres), _p(dw), _p(db),
                                          has_res, relu, 1, _p(ws), ws.numel(), M, C, _st())
    else:
        rc = lib.t2o_bn_relu_nhwc_bwd_partials_acc(_p(x), _p(dy), _p(w), _p(b), _p(mean), _p(invstd), _p(dx), _p(dw), _p(db), relu, 1,
                                                   _p(partial), partial.shape[0], _p(ws), ws.numel(), M, C, _st())
    assert rc == 0
    return (dx, dw, db) + ((dres,) if has_res else ())


def _dual(lib, x, xs, dy, w, b, w2, b2, partial):
    M, C = x.shape
    z = lambda: torch.zeros(C, device=DEV)
    rm, rv, rms, rvs = z(), z() + 1, z(), z() + 1
    mean, invstd, means, invstds, out = z(), z(), z(), z(), torch.empty_like(x)
    ws = torch.empty(lib.t2o_bn_dual_nhwc_workspace_bytes(M, C), dtype=torch.uint8, device=DEV)
    rc = lib.t2o_bn_dual_relu_nhwc_fwd(_p(x), _p(partial), 0 if partial is None else partial.shape[0], _p(xs), _p(w), _p(b), _p(rm), _p(rv),
                                       _p(mean), _p(invstd), _p(w2), _p(b2), _p(rms), _p(rvs), _p(means), _p(invstds), _p(out), 0.1, 1e-5,
                                       0.1, 1e-5, _p(ws), ws.numel(), M, C, _st())
    assert rc == 0
    dx, dxs = torch.empty_like(x), torch.empty_like(x)
    dw, db, dw2, db2 = z() + 1, z() - 1, z() + 2, z() - 2
    rc = lib.t2o_bn_dual_relu_nhwc_bwd_acc(_p(x), _p(xs), _p(out), _p(dy), _p(w), _p(b), _p(mean), _p(invstd), _p(w2), _p(b2), _p(means),
                                           _p(invstds), _p(dx), _p(dxs), _p(dw), _p(db), _p(dw2), _p(db2), 1, _p(ws), ws.numel(), M, C, _st())
    assert rc == 0
    return out, mean, invstd, means, invstds, rm, rv, rms, rvs, dx, dxs, dw, db, dw2, db2


def _case(lib, rng, fused_counters=None):
    """One random shape run through every folded entry point; returns the list of output tensors (and the launch count)."""
    C = int(rng.choice([64, 128, 256, 512]))
    # (M >= 2048: every entry point's apply grid has at least its C/4 (pair: C/2) finalizer workgroups -- the fold runs; below, it
    # falls back to the separate launch for some: covered, not counted)
    M = int(rng.choice([64, 1000, 2048, 3001, 4096, 16384, 65536 // (C // 64) + 2048]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    r = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(DEV)
    x, res, dy, xs = r(M, C), r(M, C), r(M, C), r(M, C)
    w, b, w2, b2 = r(C) + 1.0, r(C), r(C) + 1.0, r(C)
    rows = int(rng.choice([1, 7, 64, 300]))
    partial = r(rows, 2, C).abs() + 0.5                    # (any positive rows: the finalize body only adds them up)
    partial[:, 1] += partial[:, 0] ** 2 * 4.0 / max(M, 1)
    outs, launches = [], 0
    for relu in (0, 1):
        for rs in (None, res):
            for pt in (None, partial):
                o = _fwd(lib, x, rs, w, b, relu, pt)
                outs += list(o)
                launches += 1
    out, mean, invstd = _fwd(lib, x, res, w, b, 1, None)[:3]
    launches += 1
    for has_res, relu in ((0, 1), (1, 1), (0, 0)):
        outs += list(_bwd(lib, x, out, dy, w, b, mean, invstd, has_res, relu, None))
        launches += 1
    outs += list(_bwd(lib, x, None, dy, w, b, mean, invstd, 0, 1, partial))
    launches += 1
    for pt in (None, partial):
        outs += list(_dual(lib, x, xs, dy, w, b, w2, b2, pt))
        launches += 2
    if fused_counters is not None:
        torch.cuda.synchronize()
        assert int(fused_counters.abs().sum()) == 0, 'the counter block is not zero after the calls'
    return outs, (launches if M >= 2048 else 0)


def test_folded_finalize_is_bit_identical_over_2000_randomised_launches():
    from t2onet_amd import _lib
    lib = _lib.load()
    total = 0
    seed = 0
    while total < 2000:
        with _Mode(False):
            ref, _ = _case(lib, np.random.default_rng(seed))
        with _Mode(True) as m:
            got, n = _case(lib, np.random.default_rng(seed), m.counters)
        assert len(got) == len(ref)
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), 'seed %d output %d differs (max %g)' % (seed, i, float((a - b).abs().max()))
        total += n
        seed += 1
    assert total >= 2000


def test_fold_applies_to_the_trunk_and_removes_its_finalize_launches():
    """One trunk forward + backward in both modes: the same bits everywhere, and -- counted with the profiler -- no
    k_bn_nhwc_finalize* launch in the folded mode."""
    from torch.profiler import profile, ProfilerActivity
    from oracle import synth
    from tests.test_gpu_encoder import _encoder, _run
    img = synth.images(4, 128, 128, 31).to(DEV)
    gout = synth.uniform((4, 512), 32, -1.0, 1.0).to(DEV)
    res = {}
    for fused in (False, True):
        with _Mode(fused):
            net = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                res[fused] = _run(net, img, gout)
                torch.cuda.synchronize()
            n_fin = sum(e.count for e in prof.key_averages() if 'k_bn_nhwc_finalize' in e.key)
            assert (n_fin == 0) == fused, (fused, n_fin)
            res[fused] += ({n: b.clone() for n, b in net.named_buffers()},)
    (o0, d0, g0, b0), (o1, d1, g1, b1) = res[False], res[True]
    assert torch.equal(o0, o1) and torch.equal(d0, d1)
    assert all(torch.equal(g0[n], g1[n]) for n in g0) and all(torch.equal(b0[n], b1[n]) for n in b0)
