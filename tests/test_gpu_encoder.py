"""The encoder's one-node trunk (t2onet_amd/encoder.py) and the kernels only it uses: 1x1 stride-2 shortcut
convolutions (t2o_conv1x1.hip), the planar stem forms, the data gradient with pre-transformed weights and an addend
-- against fp64 conv2d / the fp64 PyTorch ResNet (models/actor_resnet.py:21-44, :98-107) and against the per-layer
autograd path (functional.py)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def _close(got, ref, tol=1e-5):
    ref = ref.float().cpu()
    scale = float(ref.abs().max()) or 1.0
    np.testing.assert_allclose(got.float().cpu().numpy(), ref.numpy(), rtol=tol, atol=tol * scale)


# (N, Ci, Co, H, W): ragged pixel counts (tile tails), odd sizes, one row, every channel combination of the encoder
SC_SHAPES = [(2, 64, 64, 8, 8), (3, 64, 128, 6, 10), (1, 128, 256, 7, 9), (2, 256, 512, 4, 4), (5, 64, 64, 1, 16),
             (2, 128, 64, 33, 18), (1, 64, 64, 64, 66)]


@pytest.mark.parametrize('shape', SC_SHAPES)
def test_shortcut_conv_forward_dgrad_wgrad_vs_fp64(shape):
    import t2onet_amd.functional as T
    N, Ci, Co, H, W = shape
    x = synth.uniform((N, Ci, H, W), 801, -1.0, 1.0)
    w = synth.uniform((Co, Ci, 1, 1), 802, -1.0, 1.0)
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, 2)
    dy = synth.uniform(tuple(y64.shape), 803, -1.0, 1.0)
    y64.backward(dy.double())
    y = T.conv1x1s2_forward(_cl(x), w.to(DEV))
    assert y.shape == y64.shape and y.is_contiguous(memory_format=torch.channels_last)
    _close(y, y64.detach())
    # data gradient: added into an existing tensor, only at the even positions
    base = synth.uniform((N, Ci, H, W), 804, -1.0, 1.0)
    dx = _cl(base).clone(memory_format=torch.channels_last)
    T.conv1x1s2_dgrad_acc(_cl(dy), w.to(DEV), dx)
    _close(dx, base.double() + x64.grad)
    # weight gradient: fresh, then accumulated; deterministic
    dw = T.conv1x1s2_wgrad(_cl(x), _cl(dy))
    _close(dw, w64.grad)
    assert torch.equal(dw, T.conv1x1s2_wgrad(_cl(x), _cl(dy)))
    acc = dw.clone()
    T.conv1x1s2_wgrad(_cl(x), _cl(dy), into=acc)
    _close(acc, 2 * w64.grad)


@pytest.mark.parametrize('shape', [(2, 64, 8, 8), (3, 32, 6, 40), (1, 64, 34, 66)])
def test_stem_planar_matches_channels_last_forms(shape):
    """NCHW image in / NCHW image gradient out: the same numbers as the channels-last kernels, accumulate forms add."""
    import t2onet_amd.functional as T
    N, Co, Ho, Wo = shape
    x = synth.uniform((N, 3, 2 * Ho, 2 * Wo), 811, 0.0, 1.0).to(DEV)
    w = _cl(synth.uniform((Co, 3, 3, 3), 812, -1.0, 1.0))
    dy = _cl(synth.uniform((N, Co, Ho, Wo), 813, -1.0, 1.0))
    y_ref, st_ref = T.stem_forward(x.contiguous(memory_format=torch.channels_last), w, True)
    y, st = T.stem_planar(x, w)
    assert torch.equal(y, y_ref) and torch.equal(st, st_ref)
    ref64 = F.conv2d(x.double().cpu(), w.double().cpu(), None, 2, 1)
    _close(y, ref64)
    dw_ref = T.stem_wgrad(x.contiguous(memory_format=torch.channels_last), dy)
    dw = T.stem_planar(x, w, dy, 'wgrad')
    assert torch.equal(dw, dw_ref)
    acc = dw.clone(memory_format=torch.channels_last)
    T.stem_planar(x, w, dy, 'wgrad', into=acc)
    _close(acc, 2 * dw_ref.double())
    dx_ref = T.conv3x3s2_dgrad(dy, w)                       # (N,3,2Ho,2Wo) channels-last
    dx = T.stem_planar(None, w, dy, 'dgrad')
    assert dx.is_contiguous() and torch.equal(dx, dx_ref.contiguous())
    base = synth.uniform(tuple(dx.shape), 814, -1.0, 1.0).to(DEV)
    acc = base.clone()
    T.stem_planar(None, w, dy, 'dgrad', into=acc)
    _close(acc, base.double() + dx_ref.double())


def test_dgrad_with_transformed_weight_and_addend():
    import t2onet_amd.functional as T
    N, Co, Ci, H, W = 2, 128, 64, 9, 16
    dy = _cl(synth.uniform((N, Co, H, W), 821, -1.0, 1.0))
    w = _cl(synth.uniform((Co, Ci, 3, 3), 822, -1.0, 1.0))
    add = _cl(synth.uniform((N, Ci, H, W), 823, -1.0, 1.0))
    ref = T.conv3x3_dgrad(dy, w)
    wt = T.conv_weight_transform(w, 9, True)
    assert torch.equal(T.conv3x3_dgrad_pre(dy, wt, Ci), ref)
    assert torch.equal(T.conv3x3_dgrad_pre(dy, wt, Ci, add), ref + add)


def _encoder(seed=5, bias=(-0.2, 0.2)):
    from t2onet_amd.actor_resnet import ResNet
    torch.manual_seed(seed)
    net = ResNet(3, 18, 512)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.0 if bias[0] > 1 else 1.5)
                m.bias.uniform_(*bias)
    return net


def _run(net, img, gout):
    img = img.clone().requires_grad_(True)
    out = net(img)
    out.backward(gout)
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    return out.detach(), img.grad.detach(), grads


@pytest.mark.parametrize('size', [(64, 256), (128, 128), (96, 160)])
@pytest.mark.parametrize('bias', [(3.0, 4.0), (-0.2, 0.2)])
def test_trunk_matches_per_layer_path_and_fp64(monkeypatch, bias, size):
    """One-node trunk vs the per-layer autograd path and vs the fp64 PyTorch module.

    Among the 1.6 M ReLU inputs of this small case some lie within fp32 rounding of zero (|pre-activation| < 1e-6): in
    fp32 such an element may land on the other side of the kink, and ONE flipped mask moves the gradients of a 4-image
    batch by ~1 % in L2 (met: layer3.0, |pre| = 7.5e-7).  bias (3, 4) with weights <= 1 keeps the pre-activations 3 sigma
    above the kink (0.1 % still masked, none expected within 1e-6 of it): there everything is compared elementwise.  With the default-like
    biases only the forward is compared elementwise and the gradients in relative L2."""
    import t2onet_amd.actor_resnet as R
    N, (H, W) = 4, size                                     # (64,256): the LDS-DMA kernels everywhere; (128,128): the reference's
    img = synth.images(N, H, W, 31)                         # training size, 4 x 4 last stage; (96,160): odd stages (3 x 5)
    gout = synth.uniform((N, 512), 32, -1.0, 1.0)
    cpu = _encoder(bias=bias).double().train()
    ref_out, ref_dimg, ref_g = _run(cpu, img.double(), gout.double())
    nets = {}
    for trunk in (True, False):
        monkeypatch.setattr(R, '_TRUNK', trunk)
        net = _encoder(bias=bias).to(DEV).to(memory_format=torch.channels_last).train()
        if trunk:
            assert net.trunk_plan().supported(img.to(DEV))
        nets[trunk] = (net,) + _run(net, img.to(DEV), gout.to(DEV))

    def rel_l2(got, ref):
        ref = ref.float().cpu()
        return float((got.float().cpu() - ref).norm() / ref.norm())

    tight = bias[0] > 1
    _, out1, dimg1, g1 = nets[True]
    _, out0, dimg0, g0 = nets[False]
    _close(out1, out0, 2e-5)
    if tight:
        _close(dimg1, dimg0, 1e-4)
        for n in g1:
            _close(g1[n], g0[n], 2e-4)                      # (the trunk's deep stages run Winograd, the per-layer path the direct
                                                            # kernels: two fp32 algorithms, each within 2e-4 of fp64 below)
    for trunk in (True, False):
        net, out, dimg, g = nets[trunk]
        _close(out, ref_out, 2e-4)
        if tight:
            _close(dimg, ref_dimg, 2e-4)
            for n, v in g.items():
                _close(v, ref_g[n], 2e-4)
        else:
            assert rel_l2(dimg, ref_dimg) < 3e-2
            for n, v in g.items():
                assert rel_l2(v, ref_g[n]) < 3e-2, n
        for (n, b), (_, rb) in zip(net.named_buffers(), cpu.named_buffers()):
            _close(b, rb, 1e-4)                              # running statistics and num_batches_tracked


def test_trunk_accumulates_into_persistent_gradients_and_is_deterministic():
    """Every parameter has a dense .grad REGISTERED for in-place accumulation (functional.enable_grad_accumulation -- what
    the Trainer's flat buffer does): the trunk adds into it inside its kernels, hands autograd nothing, two backward passes
    give exactly twice one, and repeated runs are bit-identical.  Unregistered .grad tensors get ordinary autograd
    gradients (a forward, zero_grad(set_to_none), backward loop must work): second half."""
    import t2onet_amd.functional as T
    N, H, W = 2, 32, 256
    img = synth.images(N, H, W, 41).to(DEV)
    gout = synth.uniform((N, 512), 42, -1.0, 1.0).to(DEV)

    def run(passes):
        net = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
        for p in net.parameters():
            p.grad = torch.zeros_like(p)                    # (zeros_like keeps the channels-last strides)
        T.enable_grad_accumulation(net.parameters())
        keep = [p.grad for p in net.parameters()]
        for _ in range(passes):
            x = img.clone().requires_grad_(True)
            net(x).backward(gout)
        assert all(p.grad is k for p, k in zip(net.parameters(), keep))
        return {n: p.grad.clone() for n, p in net.named_parameters()}, x.grad.clone()

    g1, d1 = run(1)
    g1b, d1b = run(1)
    assert torch.equal(d1, d1b) and all(torch.equal(g1[n], g1b[n]) for n in g1)
    fresh = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
    out, dimg, gf = _run(fresh, img, gout)
    assert torch.equal(dimg, d1)
    for n in g1:
        _close(g1[n], gf[n], 1e-6)
    g2, _ = run(2)
    trunk_names = [n for n in g1 if not n.startswith('fc.')]
    for n in trunk_names:                                    # batch statistics do not depend on the running buffers
        _close(g2[n], 2 * g1[n].double(), 1e-5)
    # not registered: .grad present at the forward, set to None before the backward (optimizer.zero_grad()) -> autograd
    # delivers fresh gradients, nothing is written into the orphaned tensors
    net = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    orphans = [p.grad for p in net.parameters()]
    x = img.clone().requires_grad_(True)
    y = net(x)
    for p in net.parameters():
        p.grad = None
    y.backward(gout)
    assert all(float(o.abs().sum()) == 0.0 for o in orphans)
    for n, p in net.named_parameters():
        assert p.grad is not None, n
        _close(p.grad, gf[n], 1e-6)
    # registered, then the gradient tensor replaced between forward and backward: same
    net = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    T.enable_grad_accumulation(net.parameters())
    x = img.clone().requires_grad_(True)
    y = net(x)
    for p in net.parameters():
        p.grad = None
    y.backward(gout)
    for n, p in net.named_parameters():
        _close(p.grad, gf[n], 1e-6)


@pytest.mark.parametrize('size,spare', [((64, 256), 0), ((128, 128), 1), ((64, 256), 1)])
def test_weight_gradients_once_per_step_equal_per_pass_gradients(size, spare):
    """encoder.WgradArena (the Trainer's opt-in): three trunk passes whose direct 3x3 / 1x1 weight gradients are formed by ONE
    launch per layer over the passes' (x, dy) arenas (Winograd layers: one batch of GEMMs over the passes' V and A dY A^T),
    against the same passes with a weight-gradient launch each.  Same kernels over 3 N images instead of 3 x N: equal up to
    the order of the split-K sums; everything else bit-identical.  spare = 1: the arena holds one pass more than is used
    (the Trainer's one arena per batch shape, sized for the teacher-forced step: the Winograd layers' GEMMs then run over the
    first rows of every arena plane -- t2o_gemm_tn_batched_ld)."""
    import t2onet_amd.functional as T
    from t2onet_amd.encoder import WgradArena
    N, (H, W), P = 2, size, 3
    imgs = [synth.images(N, H, W, 51 + p).to(DEV) for p in range(P)]
    gouts = [synth.uniform((N, 512), 61 + p, -1.0, 1.0).to(DEV) for p in range(P)]

    def run(use_arena):
        net = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        T.enable_grad_accumulation(net.parameters())
        plan = net.trunk_plan()
        if use_arena:
            plan.__dict__['arena'] = WgradArena(plan, N, H, W, P + spare, torch.device(DEV))
            assert len(plan.arena.layers) >= 10 and (len(plan.arena.wino) == 6 or size != (64, 256))
        xs = [im.clone().requires_grad_(True) for im in imgs]
        outs = [net(x) for x in xs]
        total = sum((o * g).sum() for o, g in zip(outs, gouts))
        total.backward()
        if use_arena:
            assert plan.arena.done == set(range(P))
            plan.arena.flush(plan)
            assert plan.arena.n_passes == 0
        return {n: p.grad.clone() for n, p in net.named_parameters()}, [x.grad.clone() for x in xs], [o.detach() for o in outs]

    g0, d0, o0 = run(False)
    g1, d1, o1 = run(True)
    assert all(torch.equal(a, b) for a, b in zip(o0, o1)) and all(torch.equal(a, b) for a, b in zip(d0, d1))
    moved = 0
    for n in g0:
        _close(g1[n], g0[n], 2e-5)
        moved += int(not torch.equal(g1[n], g0[n]))
    assert moved >= 10                                       # (the deferred layers did take the other summation order)
    # a pass whose backward never ran (its output unused) is left out; the others still arrive
    net = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    T.enable_grad_accumulation(net.parameters())
    plan = net.trunk_plan()
    plan.__dict__['arena'] = WgradArena(plan, N, H, W, P + 1, torch.device(DEV))
    outs = [net(im.clone().requires_grad_(True)) for im in imgs]
    ((outs[0] * gouts[0]).sum() + (outs[2] * gouts[2]).sum()).backward()
    assert plan.arena.done == {0, 2}
    plan.arena.flush(plan)
    ref = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
    for p in ref.parameters():
        p.grad = torch.zeros_like(p)
    T.enable_grad_accumulation(ref.parameters())
    ((ref(imgs[0].clone().requires_grad_(True)) * gouts[0]).sum() + (ref(imgs[2].clone().requires_grad_(True)) * gouts[2]).sum()).backward()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        _close(p.grad, q.grad, 2e-5)


@pytest.mark.parametrize('size', [(64, 256), (96, 160)])
def test_dual_batch_norm_of_shortcut_blocks_is_bit_identical_to_the_separate_passes(monkeypatch, size):
    """t2o_bn_dual_relu_nhwc_fwd / _bwd_acc (out = relu(bn2(y2) + bn_s(ys)) and both backward passes from one sweep over the
    gated gradient) against the separate kernels they replace: same arithmetic in the same order, so outputs, running
    statistics, image gradient and every parameter gradient must be bit-identical."""
    import t2onet_amd.encoder as E
    N, (H, W) = 3, size
    img = synth.images(N, H, W, 71).to(DEV)
    gout = synth.uniform((N, 512), 72, -1.0, 1.0).to(DEV)
    res = {}
    for dual in (True, False):
        monkeypatch.setattr(E, '_DUAL_BN', dual)
        net = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
        out, dimg, g = _run(net, img, gout)
        res[dual] = (out, dimg, g, {n: b.clone() for n, b in net.named_buffers()})
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for n in res[True][2]:
        assert torch.equal(res[True][2][n], res[False][2][n]), n
    for n in res[True][3]:
        assert torch.equal(res[True][3][n], res[False][3][n]), n


@pytest.mark.parametrize('size,wino', [((64, 256), False), ((96, 160), False), ((256, 256), False), ((256, 256), True)])
def test_batch_norm_backward_sums_from_the_data_gradient_epilogue(monkeypatch, size, wino):
    """bn1's backward sums formed in the epilogue of conv2's data gradient (t2o_conv3x3_dgrad_pre_bnsums_nhwc +
    t2o_bn_relu_nhwc_bwd_partials_acc; the direct-kernel stages) against the batch norm's own sums pass: the same gated terms
    added in another order -- outputs identical, gradients equal to fp32 summation noise."""
    import t2onet_amd.encoder as E
    N, (H, W) = 3, size
    img = synth.images(N, H, W, 81).to(DEV)
    gout = synth.uniform((N, 512), 82, -1.0, 1.0).to(DEV)
    res = {}
    monkeypatch.setattr(E, '_WINO_FUSED', wino)             # (the direct kernels' epilogue, or -- 256 x 256 -- the on-chip Winograd kernel's)
    for fused in (True, False):
        monkeypatch.setattr(E, '_BN_SUMS_EPILOGUE', fused)
        net = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
        res[fused] = _run(net, img, gout)
    assert torch.equal(res[True][0], res[False][0])
    assert not all(torch.equal(res[True][2][n], res[False][2][n]) for n in res[True][2])      # (the fused form did run)
    _close(res[True][1], res[False][1], 2e-5)
    for n in res[True][2]:
        _close(res[True][2][n], res[False][2][n], 2e-5)


@pytest.mark.parametrize('shape', [(2, 64, 64, 16, 24), (64, 64, 64, 64, 64), (8, 128, 128, 32, 32), (3, 128, 64, 8, 40)])
def test_data_gradient_with_batch_norm_sums_epilogue(shape):
    """t2o_conv3x3_dgrad_pre_bnsums_nhwc alone: dx bit-identical to the plain data gradient; the rows add up to the batch
    norm backward's sums (fp64 reference: sum of g and of g * xhat per channel, g = dx gated by bn_x * sc + sh > 0)."""
    import ctypes
    import t2onet_amd.functional as T
    from t2onet_amd import _lib
    N, Ci, Co, H, W = shape
    lib = _lib.load()
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    dy, w = cl(synth.uniform((N, Co, H, W), 91, -1.0, 1.0)), cl(synth.uniform((Co, Ci, 3, 3), 92, -0.2, 0.2))
    bn_x = cl(synth.uniform((N, Ci, H, W), 93, -2.0, 2.0))
    mean, invstd = synth.uniform((Ci,), 94, -0.3, 0.3).to(DEV), synth.uniform((Ci,), 95, 0.5, 2.0).to(DEV)
    gamma, beta = synth.uniform((Ci,), 96, -1.5, 1.5).to(DEV), synth.uniform((Ci,), 97, -0.5, 0.5).to(DEV)
    wt = T.conv_weight_transform(w, 9, True)
    ref_dx = T.conv3x3_dgrad_pre(dy, wt, Ci)
    n_rows = lib.t2o_conv3x3_dgrad_bnsums_rows(N, H, W, Ci, Co)
    assert n_rows > 0
    rows = torch.full((n_rows, 2, Ci), float('nan'), device=DEV)
    dx = torch.empty_like(ref_dx)
    ws = torch.zeros(64 << 10, dtype=torch.uint8, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.t2o_conv3x3_dgrad_pre_bnsums_nhwc(p(dy), p(wt), p(dx), p(bn_x), p(mean), p(invstd), p(gamma), p(beta), p(rows), p(ws), ws.numel(),
                                               N, H, W, Ci, Co, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, 't2o_conv3x3_dgrad_pre_bnsums_nhwc')
    assert torch.equal(dx, ref_dx)
    sc = gamma * invstd
    sh = beta - mean * sc
    v = lambda t: t.view(1, -1, 1, 1)
    gate = (bn_x * v(sc) + v(sh)) > 0
    g = torch.where(gate, dx, torch.zeros_like(dx)).double()
    xhat = ((bn_x - v(mean)) * v(invstd)).double()
    want = torch.stack([g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))])
    got = rows.double().sum(0)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * scale, (float((got - want).abs().max()), scale)


@pytest.mark.parametrize('size', [(256, 256), (128, 384)])
@pytest.mark.parametrize('bias', [(3.0, 4.0), (-0.2, 0.2)])
def test_on_chip_winograd_stages_against_the_direct_kernels(monkeypatch, bias, size):
    """The 64- / 128-channel stride-1 layers on t2o_wino_fused_conv_nhwc (forward and data gradient) against the same trunk
    on the direct kernels.  As in test_trunk_matches_per_layer_path_and_fp64: with pre-activations kept away from the ReLU
    kink (bias 3..4) everything is compared elementwise; with default-like biases a mask flipped by fp32 rounding moves a
    2-image batch's gradients by ~1 %, so those are compared in relative L2."""
    import t2onet_amd.encoder as E
    N, (H, W) = 2, size
    img = synth.images(N, H, W, 85).to(DEV)
    gout = synth.uniform((N, 512), 86, -1.0, 1.0).to(DEV)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(E, '_WINO_FUSED', fused)
        net = _encoder(bias=bias).to(DEV).to(memory_format=torch.channels_last).train()
        res[fused] = _run(net, img, gout)
    assert not torch.equal(res[True][0], res[False][0])      # (two algorithms did run)
    _close(res[True][0], res[False][0], 2e-5)

    def rel_l2(got, ref):
        return float((got.double() - ref.double()).norm() / ref.double().norm())
    if bias[0] > 1:
        _close(res[True][1], res[False][1], 2e-4)
        for n in res[True][2]:
            _close(res[True][2][n], res[False][2][n], 2e-4)
    else:
        assert rel_l2(res[True][1], res[False][1]) < 3e-2
        for n in res[True][2]:
            assert rel_l2(res[True][2][n], res[False][2][n]) < 3e-2, n


class _EntryPointRecorder:
    """Stands in for the ctypes library: every C-ABI call goes through, its name is noted."""

    def __init__(self, lib):
        self._lib, self.calls = lib, []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*args):
            self.calls.append(name)
            return fn(*args)
        return call


@pytest.mark.parametrize('shape', [(2, 256, 256), (4, 128, 128)])
def test_flop_table_names_the_kernels_a_pass_really_calls(monkeypatch, shape):
    """bench.py prices the train step's matrix work from TrunkPlan.flop_table.  Here one forward + backward
    of the trunk runs with every C-ABI call recorded, and the 3x3 layers' entry points -- counted per kernel family and
    direction -- must be exactly the table's rows."""
    from collections import Counter
    import t2onet_amd._lib as L
    N, H, W = shape
    net = _encoder().to(DEV).to(memory_format=torch.channels_last).train()
    plan = net.trunk_plan()
    img = synth.images(N, H, W, 31).to(DEV)
    assert plan.supported(img)
    want = Counter((r[2], r[1]) for r in plan.flop_table(N, H, W) if r[0].startswith('block') and 'shortcut' not in r[0])
    rec = _EntryPointRecorder(L.load())
    monkeypatch.setattr(L, '_lib', rec)
    _run(net, img, synth.uniform((N, 512), 32, -1.0, 1.0).to(DEV))
    monkeypatch.undo()
    family = {'t2o_wino_fused_conv_nhwc': 'wino_fused', 't2o_wino_fused_conv_bnsums_nhwc': 'wino_fused', 't2o_wino_fused_wgrad_nhwc': ('wino_wgrad', 'wgrad'),
              't2o_gemm_nt_batched': 'wino_sep', 't2o_gemm_tn_batched_ld': ('wino_sep', 'wgrad'),
              't2o_conv3x3_fwd_stats_nhwc': ('direct', 'fwd'), 't2o_conv3x3_dgrad_pre_nhwc': ('direct', 'dgrad'),
              't2o_conv3x3s2_dgrad_pre_nhwc': ('direct', 'dgrad'), 't2o_conv3x3_dgrad_pre_bnsums_nhwc': ('direct', 'dgrad'),
              't2o_conv3x3_wgrad_acc_nhwc': ('direct', 'wgrad'), 't2o_conv3x3_any_fwd_nhwc': ('generic', 'fwd'),
              't2o_conv3x3_any_dgrad_nhwc': ('generic', 'dgrad'), 't2o_conv3x3_any_wgrad_nhwc': ('generic', 'wgrad')}
    got = Counter()
    n_fb = Counter()                                        # forward / data-gradient launches of the two-direction entry points, in call order
    for name in rec.calls:
        f = family.get(name)
        if f is None:
            continue
        if isinstance(f, tuple):
            got[f] += 1
        else:
            n_fb[f] += 1
    # the forward runs first: of a family's two-direction launches the table's forward count are forwards, the rest data gradients
    for fam, n in n_fb.items():
        nf = want[(fam, 'fwd')]
        got[(fam, 'fwd')] += nf
        got[(fam, 'dgrad')] += n - nf
    assert +got == +want, (got, want)
