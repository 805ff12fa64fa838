"""Parity tests proper: the HIP kernels, called through the C ABI (ctypes) behind the
reference-shaped Executor/Operator API, against
  (a) the committed golden fixtures = outputs of the reference itself (tools/gen_golden.py),
  (b) the oracle on seeded inputs at sizes it finishes in seconds,
  (c) size-independent properties at BASELINE.json's full sizes.

Tolerance (north_star: <= 1e-5 relative fp32): forward rtol 1e-5 + atol 2e-6 -- a RELATIVE bound with a floor of two fp32
ulps of 1.0 for values at zero (measured with tools/measure_parity.py, round 5: every output of every operator is within
1e-5 relative of the reference / the oracle, no absolute floor needed; the kernels follow the reference's operation order
with -ffp-contract=off), so a regression on dark pixels cannot hide; gradients are closed forms, so
they are compared with the oracle's fp64 autograd at 1e-5 and with the reference's fp32
autograd at 2e-5 + 1e-4 rel (its HSV round trip is itself that noisy: SURVEY.md section 7)."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref, synth

pytestmark = pytest.mark.gpu

OPT = cpu_ref.default_opt()
OPS = [0, 1, 2, 3, 5, 6, 7]


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def executor(dev):
    import t2onet_amd
    ex = t2onet_amd.Executor(t2onet_amd.default_options())
    ex.load_state_dict(synth.fill_state_dict(ex.state_dict(), seed=3))
    return ex.to(dev)


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'operators.npz'))


def test_library_is_loaded_natively():
    from t2onet_amd import _lib
    lib = _lib.load()
    assert lib.t2o_abi_version() == 4
    with open('/proc/self/maps') as f:
        assert 'libt2onet_hip.so' in f.read()


def test_executor_metadata(executor, gold):
    assert list(gold['name_list']) == executor.name_list
    assert list(gold['param_num']) == [executor.get_param_num(i) for i in range(8)]
    for i in range(8):
        np.testing.assert_allclose(gold['param_bnd'][i], executor.get_param_bnd(i), atol=1e-12)


@pytest.mark.parametrize('op', OPS)
def test_golden_reference_outputs(executor, gold, dev, op):
    B, H, W = 2, 24, 20
    img = synth.images(B, H, W, 11)
    gout = synth.uniform((B, 3, H, W), 12, -1.0, 1.0).to(dev)
    masks = {'none': None, 'm1': synth.masks(B, 1, H, W, 14), 'm3': synth.masks(B, 3, H, W, 15, soft=False)}
    for si, setting in enumerate(['mid', 'strong', 'neg']):
        for mname, mask in masks.items():
            key = 'op%d_%s_%s' % (op, setting, mname)
            if key + '_out' not in gold:
                continue
            x = img.to(dev).requires_grad_(True)
            p = synth.op_params(op, B, 100 + 10 * op + si, setting).to(dev).requires_grad_(True)
            out, par = executor.execute(x, op, None if mask is None else mask.to(dev), specified_param=p)
            assert par is p
            out.backward(gout)
            np.testing.assert_allclose(out.detach().cpu().numpy(), gold[key + '_out'], rtol=1e-5, atol=2e-6, err_msg=key)
            np.testing.assert_allclose(x.grad.cpu().numpy(), gold[key + '_gimg'], rtol=1e-4,
                                       atol=2e-5 if op in (0, 2) else 5e-6, err_msg=key)
            scale = max(1.0, float(np.abs(gold[key + '_gparam']).max()))
            np.testing.assert_allclose(p.grad.cpu().numpy(), gold[key + '_gparam'], rtol=1e-4, atol=5e-5 * scale, err_msg=key)


@pytest.mark.parametrize('op', OPS)
def test_golden_learned_parameter_path(executor, gold, dev, op):
    B, H, W = 2, 24, 20
    img = synth.images(B, H, W, 11).to(dev)
    f = synth.uniform((B, 512), 13, -1.0, 1.0).to(dev).requires_grad_(True)
    out, par = executor.execute(img, op, None, features=f)
    np.testing.assert_allclose(par.detach().cpu().numpy(), gold['op%d_feat_param' % op], rtol=1e-5, atol=1e-5)
    # (round 6, tools/measure_parity.py `learned op`: at rtol 1e-5 the outputs need an absolute floor of <= 1.2e-7 for six of the
    # seven operators and 3.2e-6 for the colour curve -- its 24 knots come from the parameter head's two GEMMs, whose 3.6e-7
    # rounding differences the curve's normalisation multiplies; twice that is the bound)
    np.testing.assert_allclose(out.detach().cpu().numpy(), gold['op%d_feat_out' % op], rtol=1e-5, atol=6e-6 if op == 3 else 2e-6)
    if 'op%d_feat_gfeat' % op in gold:
        out.backward(synth.uniform((B, 3, H, W), 12, -1.0, 1.0).to(dev))
        g = gold['op%d_feat_gfeat' % op]
        np.testing.assert_allclose(f.grad.cpu().numpy(), g, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(g).max())))


def test_identity_and_errors(executor, dev):
    img = synth.images(2, 24, 20, 11).to(dev)
    out, par = executor.execute(img, -1, None, features=img)
    assert out is img and par.shape == (2, 24) and float(par.abs().sum()) == 0.0
    with pytest.raises(AssertionError):
        executor.execute(img, 0, None)                                    # neither features nor param
    with pytest.raises(AssertionError):                                   # operators.py:113, on the Operator itself
        executor.brightness_op.execute(img, None, features=torch.zeros(2, 512, device=dev),
                                       specified_param=torch.zeros(2, 1, device=dev))
    with pytest.raises(IndexError):
        executor.execute(img, 8, None, specified_param=torch.zeros(2, 1, device=dev))
    with pytest.raises(RuntimeError):
        executor.execute(img.cpu(), 0, None, specified_param=torch.zeros(2, 1))      # no CPU fallback
    with pytest.raises(RuntimeError):
        executor.execute(img, 4, None, specified_param=torch.zeros(2, 1, device=dev))   # inpaint


def _oracle64(op, img, p, mask, gout):
    from tests.test_block_programs_cpu import oracle_fwd_bwd
    return oracle_fwd_bwd(op, img, p, mask, gout, torch.float64)


@pytest.mark.parametrize('shape', [(2, 23, 19), (1, 40, 150), (3, 17, 68), (2, 128, 128), (1, 397, 600)])
@pytest.mark.parametrize('op', OPS)
def test_vs_oracle_ragged_sizes(executor, dev, op, shape):
    B, H, W = shape
    img = synth.images(B, H, W, 61)
    gout = synth.uniform((B, 3, H, W), 62, -1.0, 1.0)
    for mask in (None, synth.masks(B, 1, H, W, 63)):
        p = synth.op_params(op, B, 400 + op, 'mid')
        o_ref = cpu_ref.operator_apply(op, img, p, mask, OPT)
        _, gi64, gp64 = _oracle64(op, img, p, mask, gout)
        x = img.to(dev).requires_grad_(True)
        pp = p.to(dev).requires_grad_(True)
        out, _ = executor.execute(x, op, None if mask is None else mask.to(dev), specified_param=pp)
        out.backward(gout.to(dev))
        np.testing.assert_allclose(out.detach().cpu().numpy(), o_ref.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(x.grad.cpu().numpy(), gi64.numpy(), rtol=1e-5, atol=5e-6)
        scale = max(1.0, float(gp64.abs().max()))
        np.testing.assert_allclose(pp.grad.cpu().numpy(), gp64.numpy(), rtol=2e-4, atol=1e-4 * scale)


@pytest.mark.parametrize('shape', [(2, 3, 4), (1, 8, 260), (1, 21, 512), (2, 16, 256), (1, 9, 1028), (3, 50, 12)])
@pytest.mark.parametrize('setting', ['mid', 'strong'])
def test_sharpness_strip_kernels(executor, dev, shape, setting):
    """W % 4 == 0: the LDS-free stencil kernels -- segment borders every 256 columns (lane 0 / 63 read
    their outside column from memory), ragged last rows / last segment, fewer rows than one strip,
    clamp active ('strong'); plain gradient and the fused-L1 form (target instead of gout)."""
    import t2onet_amd.functional as T
    B, H, W = shape
    img = synth.images(B, H, W, 71)
    gout = synth.uniform((B, 3, H, W), 72, -1.0, 1.0)
    tgt = synth.images(B, H, W, 73)
    p = synth.op_params(6, B, 470, setting)
    o_ref = cpu_ref.operator_apply(6, img, p, None, OPT)
    _, gi64, gp64 = _oracle64(6, img, p, None, gout)
    x = img.to(dev).requires_grad_(True)
    pp = p.to(dev).requires_grad_(True)
    out, _ = executor.execute(x, 6, None, specified_param=pp)
    out.backward(gout.to(dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), o_ref.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), gi64.numpy(), rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(pp.grad.cpu().numpy(), gp64.numpy(), rtol=2e-4, atol=1e-4 * max(1.0, float(gp64.abs().max())))
    # fused L1 on the stencil pair (t2o_op_fwd_l1 / t2o_op_bwd_l1 through the one-operator sequence)
    x64 = img.double().requires_grad_(True)
    p64 = p.double().requires_grad_(True)
    loss_ref = (cpu_ref.operator_apply(6, x64, p64, None, OPT) - tgt.double()).abs().mean()
    loss_ref.backward()
    x2 = img.to(dev).requires_grad_(True)
    p2 = torch.zeros(1, B, 24, device=dev)
    p2[0, :, :1] = p.to(dev)
    p2.requires_grad_(True)
    loss, _ = T.sequence_l1(x2, [6], p2, tgt.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-6
    n = B * 3 * H * W
    np.testing.assert_allclose(x2.grad.cpu().numpy(), x64.grad.float().numpy(), rtol=1e-5, atol=1e-6 / n * 10 + 1e-9)
    np.testing.assert_allclose(p2.grad[0, :, :1].cpu().numpy(), p64.grad.float().numpy(), rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize('op', OPS)
def test_edge_cases_ties_follow_pytorch(executor, dev, op):
    """Black / white / grey pixels, values on curve knots, flat saturated regions: PyTorch's tie
    conventions (inclusive clamp, first-index max/min, 1/2-1/2 elementwise max/min)."""
    from tests.test_block_programs_cpu import edge_image, oracle_fwd_bwd
    img = edge_image()
    B, _, H, W = img.shape
    gout = synth.uniform((B, 3, H, W), 92, -1.0, 1.0)
    for setting in ['mid', 'strong']:
        p = synth.op_params(op, B, 600 + op, setting)
        o_ref, gi_ref, gp_ref = oracle_fwd_bwd(op, img, p, None, gout)
        x = img.to(dev).requires_grad_(True)
        pp = p.to(dev).requires_grad_(True)
        out, _ = executor.execute(x, op, None, specified_param=pp)
        out.backward(gout.to(dev))
        np.testing.assert_allclose(out.detach().cpu().numpy(), o_ref.numpy(), rtol=0, atol=1e-6)
        gi, gi_ref = x.grad.cpu().numpy(), gi_ref.numpy()
        np.testing.assert_allclose(gi, gi_ref, rtol=1e-4, atol=5e-5 if op in (0, 2) else 2e-6)
        np.testing.assert_allclose(pp.grad.cpu().numpy(), gp_ref.numpy(), rtol=1e-4,
                                   atol=1e-4 * max(1.0, float(gp_ref.abs().max())))


@pytest.mark.parametrize('shape', [(1, 1, 1), (1, 1, 7), (2, 5, 1), (1, 3, 5), (1, 2, 130)])
def test_tiny_and_ragged_shapes(executor, dev, shape):
    B, H, W = shape
    img = synth.images(B, H, W, 93)
    gout = synth.uniform((B, 3, H, W), 94, -1.0, 1.0)
    for op in OPS:
        p = synth.op_params(op, B, 700 + op, 'mid')
        o_ref = cpu_ref.operator_apply(op, img, p, None, OPT)
        _, gi64, gp64 = _oracle64(op, img, p, None, gout)
        x = img.to(dev).requires_grad_(True)
        pp = p.to(dev).requires_grad_(True)
        out, _ = executor.execute(x, op, None, specified_param=pp)
        out.backward(gout.to(dev))
        np.testing.assert_allclose(out.detach().cpu().numpy(), o_ref.numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(x.grad.cpu().numpy(), gi64.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(pp.grad.cpu().numpy(), gp64.numpy(), rtol=1e-4, atol=2e-5 * max(1.0, float(gp64.abs().max())))


def test_per_sample_operators_one_launch(executor, dev):
    B, H, W = 9, 40, 72
    img = synth.images(B, H, W, 71)
    gout = synth.uniform((B, 3, H, W), 72, -1.0, 1.0)
    ops = [0, 1, 2, 3, 5, 6, 7, -1, 6]
    params = torch.zeros(B, 24)
    for b, op in enumerate(ops):
        if op >= 0:
            params[b, :cpu_ref.OP_NPARAM[op]] = synth.op_params(op, 1, 500 + b, 'mid')[0]
    x = img.to(dev).requires_grad_(True)
    pp = params.to(dev).requires_grad_(True)
    out, par = executor.execute_per_sample(x, torch.tensor(ops, device=dev), None, specified_param=pp)
    out.backward(gout.to(dev))
    for b, op in enumerate(ops):
        if op < 0:
            assert torch.equal(out[b].cpu(), img[b]) and torch.equal(x.grad[b].cpu(), gout[b])
            continue
        n = cpu_ref.OP_NPARAM[op]
        o_ref = cpu_ref.operator_apply(op, img[b:b + 1], params[b:b + 1, :n], None, OPT)
        _, gi64, gp64 = _oracle64(op, img[b:b + 1], params[b:b + 1, :n], None, gout[b:b + 1])
        np.testing.assert_allclose(out[b].detach().cpu().numpy(), o_ref[0].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(x.grad[b].cpu().numpy(), gi64[0].numpy(), rtol=1e-5, atol=5e-6)
        np.testing.assert_allclose(pp.grad[b, :n].cpu().numpy(), gp64[0].numpy(), rtol=2e-4,
                                   atol=1e-4 * max(1.0, float(gp64.abs().max())))
        assert float(pp.grad[b, n:].abs().sum()) == 0.0


def test_per_sample_learned_heads_match_grouped_execute(executor, dev):
    """execute_per_sample(features=...) == the reference's per-group execute + regroup."""
    B, H, W = 8, 32, 32
    img = synth.images(B, H, W, 73).to(dev)
    feats = synth.uniform((B, 512), 74, -1.0, 1.0).to(dev)
    ops = torch.tensor([0, 1, 2, 3, 5, 6, -1, 3], device=dev)
    out, par = executor.execute_per_sample(img, ops, None, features=feats)
    for b in range(B):
        o, p = executor.execute(img[b:b + 1], int(ops[b]), None, features=feats[b:b + 1])
        # (the batched heads sum their 512-term dot products in another order than the library GEMM of the
        # single-operator path: parameters agree to ~1e-6, images to that times the operator's gain)
        assert torch.allclose(par[b:b + 1, :p.shape[1]], p, atol=5e-6)
        assert torch.allclose(out[b:b + 1], o, atol=2e-5)


def test_golden_cfg1_and_chain6(executor, gold, dev):
    # BASELINE config 1: single 256x256 image, brightness -> contrast -> saturation
    x = synth.images(1, 256, 256, 21).to(dev)
    cur = x
    for k, op in enumerate([0, 1, 2]):
        cur, _ = executor.execute(cur, op, None, specified_param=synth.op_params(op, 1, 200 + k, 'mid').to(dev))
    np.testing.assert_allclose(cur[:, :, 100:132, 60:92].cpu().numpy(), gold['cfg1_out_crop'], rtol=0, atol=1e-5)
    assert abs(cur.double().sum().item() - float(gold['cfg1_out_sum'])) < 1e-2
    # BASELINE config 2 at a small size, both product paths
    import t2onet_amd.functional as T
    B, H, W = 3, 32, 40
    ops = [0, 1, 2, 3, 5, 6]
    tgt = synth.images(B, H, W, 32).to(dev)
    for path in ('execute', 'sequence'):
        x = synth.images(B, H, W, 31).to(dev).requires_grad_(True)
        ps = [synth.op_params(op, B, 300 + k, 'mid').to(dev).requires_grad_(True) for k, op in enumerate(ops)]
        if path == 'execute':
            cur = x
            for op, p in zip(ops, ps):
                cur, _ = executor.execute(cur, op, None, specified_param=p)
            loss = T.l1_loss(cur, tgt)
        else:
            loss, acts = executor.run_sequence(x, ops, ps, tgt)
            cur = acts[-1]
        loss.backward()
        np.testing.assert_allclose(cur.detach().cpu().numpy(), gold['chain6_out'], rtol=0, atol=1e-5, err_msg=path)
        assert abs(loss.item() - float(gold['chain6_loss'])) < 1e-6                     # L1 deviation <= 1e-5
        g = gold['chain6_gimg']
        np.testing.assert_allclose(x.grad.cpu().numpy(), g, rtol=2e-3, atol=2e-3 * np.abs(g).max(), err_msg=path)
        for k, p in enumerate(ps):
            gk = gold['chain6_gparam%d' % k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), gk, rtol=2e-3, atol=2e-3 * max(np.abs(gk).max(), 1e-6),
                                       err_msg='%s gparam %d' % (path, k))


def _fp64_oracle_sequence(img, tgt, ops, params):
    """Loss and gradients of the sequence by fp64 autograd of the oracle (the arithmetic the fp32 reference approximates)."""
    x = img.double().requires_grad_(True)
    ps = [p.double().requires_grad_(True) for p in params]
    out, _ = cpu_ref.run_sequence(x, ops, ps, OPT)
    loss = cpu_ref.l1_loss(out, tgt.double())
    loss.backward()
    return out.detach(), loss.item(), x.grad, [p.grad for p in ps]


def _close_except_branch_flips(got, ref, rtol, atol, max_frac=2e-5, what=''):
    """Element-wise comparison with an fp64 evaluation: a pixel sitting within rounding of a clamp bound, a curve
    knot or a hue-sector edge takes the other branch in fp32 -- a handful of elements per million may differ by
    O(gradient); everything else must agree."""
    bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
    assert bad.mean() <= max_frac, '%s: %d of %d elements differ (max %.3e)' % (what, bad.sum(), bad.size, np.abs(got - ref).max())


@pytest.mark.parametrize('path', ['materialised', 'fused'])
def test_chain6_gradients_vs_fp64_oracle(executor, dev, path):
    """The 6-operator chain's gradients against fp64 autograd of the oracle, 100x tighter than the comparison with
    the reference's own fp32 autograd (test_golden_cfg1_and_chain6: 2e-3): that looseness is the reference's
    fp32 noise through five chained derivatives, not the kernels'."""
    B, H, W = 3, 32, 40
    ops = [0, 1, 2, 3, 5, 6]
    img, tgt = synth.images(B, H, W, 31), synth.images(B, H, W, 32)
    params = [synth.op_params(op, B, 300 + k, 'mid') for k, op in enumerate(ops)]
    ref_out, ref_loss, ref_gx, ref_gp = _fp64_oracle_sequence(img, tgt, ops, params)
    out32, _ = cpu_ref.run_sequence(img, ops, params, OPT)
    x = img.to(dev).requires_grad_(True)
    ps = [p.to(dev).requires_grad_(True) for p in params]
    fn = executor.run_sequence if path == 'materialised' else executor.run_sequence_fused
    loss, out = fn(x, ops, ps, tgt.to(dev))
    loss.backward()
    out = out if out.dim() == 4 else out[-1]
    np.testing.assert_allclose(out.detach().cpu().numpy(), out32.numpy(), rtol=0, atol=2e-6)      # fp32 arithmetic both sides
    assert abs(loss.item() - ref_loss) < 1e-6
    gx = ref_gx.numpy()
    _close_except_branch_flips(x.grad.cpu().numpy(), gx, 2e-4, 2e-5 * np.abs(gx).max(), what='gimg')
    for k, p in enumerate(ps):
        gk = ref_gp[k].numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), gk, rtol=2e-4, atol=2e-5 * max(np.abs(gk).max(), 1e-6),
                                   err_msg='gparam %d (op %d)' % (k, ops[k]))


def test_config5_sample_vs_oracle(executor, dev):
    """One sample of BASELINE config 5 (512x512, curve-heavy 8-operator chain with repeats) against the oracle:
    forward vs the fp32 restatement (1e-5), loss and every gradient vs its fp64 autograd."""
    H = W = 512
    ops = [5, 3, 5, 3, 0, 1, 2, 6]
    g = torch.Generator().manual_seed(12)
    img = torch.rand(16, 3, H, W, generator=g)[:1]
    tgt = torch.rand(16, 3, H, W, generator=g)[:1]
    rng = {5: (8, .5, 1.5), 3: (24, .5, 1.5), 0: (1, -.3, .3), 1: (1, -.3, .3), 2: (1, -.3, .3), 6: (1, 0., 1.)}
    params = [torch.rand(1, rng[op][0], generator=g) * (rng[op][2] - rng[op][1]) + rng[op][1] for op in ops]
    out32, _ = cpu_ref.run_sequence(img, ops, params, OPT)
    ref_out, ref_loss, ref_gx, ref_gp = _fp64_oracle_sequence(img, tgt, ops, params)
    for fn in (executor.run_sequence, executor.run_sequence_fused):
        x = img.to(dev).requires_grad_(True)
        ps = [p.to(dev).requires_grad_(True) for p in params]
        loss, out = fn(x, ops, ps, tgt.to(dev))
        loss.backward()
        out = out if out.dim() == 4 else out[-1]
        np.testing.assert_allclose(out.detach().cpu().numpy(), out32.numpy(), rtol=0, atol=1e-5)
        assert abs(loss.item() - ref_loss) < 1e-6
        gx = ref_gx.numpy()
        _close_except_branch_flips(x.grad.cpu().numpy(), gx, 5e-4, 5e-5 * np.abs(gx).max(), what='gimg')
        for k, p in enumerate(ps):
            gk = ref_gp[k].numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), gk, rtol=5e-4, atol=5e-5 * max(np.abs(gk).max(), 1e-6),
                                       err_msg='gparam %d (op %d)' % (k, ops[k]))


def test_full_size_properties(executor, dev):
    """BASELINE config 2 shape (bs=64, 256x256): properties that need no oracle."""
    import t2onet_amd.functional as T
    B, H, W = 64, 256, 256
    g = torch.Generator().manual_seed(10)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
    zero1 = torch.zeros(B, 1, device=dev)
    ones8, ones24 = torch.ones(B, 8, device=dev), torch.ones(B, 24, device=dev)
    # neutral parameters are the identity (up to fp32 rounding of each formula)
    for op, p, tol in [(0, zero1, 2e-6), (1, zero1, 0.0), (2, zero1, 2e-6), (3, ones24, 2e-7), (5, ones8, 2e-7), (6, zero1, 0.0)]:
        out, _ = executor.execute(img, op, None, specified_param=p)
        assert (out - img).abs().max().item() <= tol, op
    # a zero mask returns the input, a one mask the unmasked result; run-to-run bitwise reproducible
    p = torch.full((B, 1), 0.4, device=dev)
    a, _ = executor.execute(img, 6, None, specified_param=p)
    b, _ = executor.execute(img, 6, torch.ones(B, 1, H, W, device=dev), specified_param=p)
    c, _ = executor.execute(img, 6, torch.zeros(B, 1, H, W, device=dev), specified_param=p)
    assert torch.equal(a, b) and torch.equal(c, img)
    a2, _ = executor.execute(img, 6, None, specified_param=p)
    assert torch.equal(a, a2)
    # outputs stay in [0,1]; L1 agrees with torch; sequence loss == l1(acts[-1], target)
    ops = [0, 1, 2, 3, 5, 6]
    gen = torch.Generator().manual_seed(11)
    ps = [(torch.rand(B, n, generator=gen) * (hi - lo) + lo).to(dev) for n, lo, hi in
          [(1, -.3, .3), (1, -.3, .3), (1, -.3, .3), (24, .5, 1.5), (8, .5, 1.5), (1, 0., 1.)]]
    x = img.clone().requires_grad_(True)
    pl = [q.clone().requires_grad_(True) for q in ps]
    loss, acts = executor.run_sequence(x, ops, pl, tgt)
    assert acts.min().item() >= 0.0 and acts.max().item() <= 1.0
    ref = (acts[-1] - tgt).abs().mean().item()
    assert abs(loss.item() - ref) < 1e-6 and abs(T.l1_loss(acts[-1], tgt).item() - ref) < 1e-6
    loss.backward()
    g1 = [q.grad.clone() for q in pl] + [x.grad.clone()]
    # the per-operator API path gives bit-identical images and gradients
    x2 = img.clone().requires_grad_(True)
    pl2 = [q.clone().requires_grad_(True) for q in ps]
    cur = x2
    for k, (op, q) in enumerate(zip(ops, pl2)):
        cur, _ = executor.execute(cur, op, None, specified_param=q)
        assert torch.equal(cur, acts[k])
    T.l1_loss(cur, tgt).backward()
    g2 = [q.grad for q in pl2] + [x2.grad]
    for u, v in zip(g1, g2):
        assert torch.allclose(u, v, rtol=1e-5, atol=1e-9)
    # gradient of the loss w.r.t. the image has |g| <= amplification / N and is not identically zero
    assert x.grad.abs().max().item() > 0


@pytest.mark.parametrize('ops', [[0, 1, 2, 3, 5, 6], [5, 3, 5, 3, 0, 1, 2, 6], [6, 0, -1, 6, 3], [1], [6], [-1, -1],
                                 [0, 1, 2, 3, 5, 0, 1, 2, 3, 5, 7, 1]])
@pytest.mark.parametrize('shape', [(3, 32, 40), (2, 23, 19), (2, 128, 128)])
def test_fused_sequence_equals_materialised_sequence(executor, dev, ops, shape):
    """run_sequence_fused (pointwise runs in registers, LDS curve lookup) against run_sequence
    (one kernel per operator): same final image bit for bit, same loss, same gradients."""
    B, H, W = shape
    img = synth.images(B, H, W, 31).to(dev)
    tgt = synth.images(B, H, W, 32).to(dev)
    params = torch.zeros(len(ops), B, 24)
    for k, op in enumerate(ops):
        if op >= 0:
            params[k, :, :cpu_ref.OP_NPARAM[op]] = synth.op_params(op, B, 300 + k, 'mid')
    res = []
    for fn in (executor.run_sequence, executor.run_sequence_fused):
        x = img.clone().requires_grad_(True)
        p = params.to(dev).requires_grad_(True)
        loss, out = fn(x, ops, p, tgt)
        (loss * 2.0).backward()
        res.append((loss.item(), out if out.dim() == 4 else out[-1], x.grad, p.grad))
    (l0, o0, gx0, gp0), (l1, o1, gx1, gp1) = res
    assert torch.equal(o0, o1)
    assert abs(l0 - l1) < 1e-7
    assert torch.allclose(gx0, gx1, rtol=1e-5, atol=1e-9)
    assert torch.allclose(gp0, gp1, rtol=2e-4, atol=2e-5 * max(1.0, gp0.abs().max().item()))


@pytest.mark.parametrize('ops', [[0, 1, 2, 3, 5, 6], [5, 3, 5, 3, 0, 1, 2, 6], [6, 0, -1, 6, 3], [1], [6], [0, 1, 2, 3, 5],
                                 [0, 1, 2, 3, 5, 0, 1, 2, 3, 5, 7, 1], [6, 6]])
@pytest.mark.parametrize('shape', [(3, 32, 40), (2, 23, 19), (2, 128, 128), (1, 256, 256)])
@pytest.mark.parametrize('jit', [True, False])
def test_value_and_grad_equals_forward_plus_backward(executor, dev, ops, shape, jit, monkeypatch):
    """Executor.value_and_grad (t2o_fused_sequence_l1_value_grad: no forward launch for the last segment -- its backward
    kernels also emit the loss and the image) against run_sequence_fused + backward: image, image gradient and parameter
    gradients BIT-identical (same kernels, same geometry), the loss equal up to summation order.  (2,23,19): sharpness on
    the LDS-tile kernels, where the call falls back to the two launches internally."""
    import t2onet_amd.functional as T
    monkeypatch.setattr(T, '_CHAIN_JIT', jit)
    B, H, W = shape
    img = synth.images(B, H, W, 41).to(dev)
    tgt = synth.images(B, H, W, 42).to(dev)
    params = torch.zeros(len(ops), B, 24)
    for k, op in enumerate(ops):
        if op >= 0 and cpu_ref.OP_NPARAM[op]:
            params[k, :, :cpu_ref.OP_NPARAM[op]] = synth.op_params(op, B, 330 + k, 'mid')
    x = img.clone().requires_grad_(True)
    p = params.to(dev).requires_grad_(True)
    loss, out = executor.run_sequence_fused(x, ops, p, tgt)
    (loss * 1.75).backward()
    gl = torch.tensor(1.75, device=dev)
    for want_image in (True, False):
        l2, gx, gp, o2 = executor.value_and_grad(img, ops, params.to(dev), tgt, gloss=gl, want_image=want_image)
        assert abs(l2.item() - loss.item()) <= 2e-7 * max(loss.item(), 1e-3)
        assert torch.equal(gx, x.grad)
        assert torch.equal(gp, p.grad)
        assert (o2 is None) if not want_image else torch.equal(o2, out)
    l3, gx3, gp3, _ = executor.value_and_grad(img, ops, params.to(dev), tgt, gloss=gl, want_image_grad=False)
    assert gx3 is None and torch.equal(gp3, p.grad) and l3.item() == l2.item()


def test_value_and_grad_at_256_against_fp64(executor, dev):
    """BASELINE configs[1]'s list at 256 x 256 (bs = 16 of its 64: the fp64 oracle runs on the host) through the one-call
    path: loss, image, image gradient and the gradient of every operator's parameters against fp64 autograd of the oracle."""
    ops = [0, 1, 2, 3, 5, 6]
    B, H, W = 16, 256, 256
    img, tgt = synth.images(B, H, W, 51), synth.images(B, H, W, 52)
    params = [synth.op_params(op, B, 340 + k, 'mid') for k, op in enumerate(ops)]
    ref_out, ref_loss, ref_gx, ref_gp = _fp64_oracle_sequence(img, tgt, ops, params)
    loss, gimg, gparams, out = executor.value_and_grad(img.to(dev), ops, [q.to(dev) for q in params], tgt.to(dev), want_image=True)
    assert abs(loss.item() - ref_loss) < 1e-6
    _close_except_branch_flips(out.cpu().numpy(), ref_out.numpy(), 0, 2e-5, what='out')
    gx = ref_gx.numpy()
    _close_except_branch_flips(gimg.cpu().numpy(), gx, 2e-4, 2e-5 * np.abs(gx).max(), what='gimg')
    for k in range(len(ops)):
        gk = ref_gp[k].numpy()
        # (a parameter gradient is a sum over 196 608 pixels of terms of both signs, in fp32: 1e-3 relative on the small ones)
        np.testing.assert_allclose(gparams[k, :, :gk.shape[1]].cpu().numpy(), gk, rtol=2e-3,
                                   atol=2e-4 * max(np.abs(gk).max(), 1e-6), err_msg='gparam %d (op %d)' % (k, ops[k]))


def test_fused_sequence_golden_chain6(executor, gold, dev):
    B, H, W = 3, 32, 40
    ops = [0, 1, 2, 3, 5, 6]
    x = synth.images(B, H, W, 31).to(dev).requires_grad_(True)
    tgt = synth.images(B, H, W, 32).to(dev)
    ps = [synth.op_params(op, B, 300 + k, 'mid').to(dev).requires_grad_(True) for k, op in enumerate(ops)]
    loss, out = executor.run_sequence_fused(x, ops, ps, tgt)
    loss.backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), gold['chain6_out'], rtol=0, atol=1e-5)
    assert abs(loss.item() - float(gold['chain6_loss'])) < 1e-6
    g = gold['chain6_gimg']
    np.testing.assert_allclose(x.grad.cpu().numpy(), g, rtol=2e-3, atol=2e-3 * np.abs(g).max())
    for k, p in enumerate(ps):
        gk = gold['chain6_gparam%d' % k]
        np.testing.assert_allclose(p.grad.cpu().numpy(), gk, rtol=2e-3, atol=2e-3 * max(np.abs(gk).max(), 1e-6))


def test_attention_core(dev):
    import t2onet_amd.functional as T
    for (B, L, D) in [(4, 14, 512), (64, 17, 512), (3, 1, 64), (5, 64, 1024), (2, 9, 256), (3, 5, 768), (2, 33, 128)]:
        q = synth.uniform((B, D), 91, -1, 1)
        ctx = synth.uniform((B, L, D), 92, -0.2, 0.2)
        ctx[:, L // 2:] *= 0.0 if L > 2 else 1.0          # zero-padded encoder rows take part in the softmax
        gm = synth.uniform((B, D), 93, -1, 1)
        ga = synth.uniform((B, L), 94, -1, 1)
        q64, c64 = q.double().requires_grad_(True), ctx.double().requires_grad_(True)
        s = torch.bmm(q64.unsqueeze(1), c64.transpose(1, 2))
        a = torch.softmax(s.view(-1, L), 1).view(B, 1, L)
        mix = torch.bmm(a, c64).squeeze(1)
        ((mix * gm.double()).sum() + (a.squeeze(1) * ga.double()).sum()).backward()
        qd, cd = q.to(dev).requires_grad_(True), ctx.to(dev).requires_grad_(True)
        m2, a2 = T.attention_core(qd, cd)
        ((m2 * gm.to(dev)).sum() + (a2 * ga.to(dev)).sum()).backward()
        np.testing.assert_allclose(a2.detach().cpu().numpy(), a.squeeze(1).detach().numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(m2.detach().cpu().numpy(), mix.detach().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(qd.grad.cpu().numpy(), q64.grad.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(cd.grad.cpu().numpy(), c64.grad.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('shape', [(1, 3, 48, 40), (2, 3, 33, 70), (1, 1, 5, 7), (1, 3, 397, 600)])
def test_ssim_kernel(dev, golden_dir, shape):
    import t2onet_amd.functional as T
    a = synth.uniform(shape, 51)
    b = (a + synth.uniform(shape, 52, -0.1, 0.1)).clamp(0, 1)
    ref = cpu_ref.ssim(a, b, size_average=False)
    out = T.ssim(a.to(dev), b.to(dev), size_average=False)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    assert abs(T.ssim(a.to(dev), b.to(dev)).item() - ref.mean().item()) < 1e-6
    if shape == (1, 3, 48, 40):
        g = np.load(os.path.join(golden_dir, 'ssim.npz'))
        assert abs(out[0].item() - float(g['ssim'])) < 1e-5


@pytest.mark.parametrize('shape', [(1, 3, 48, 40), (2, 3, 33, 70), (1, 1, 5, 7), (1, 3, 397, 600), (4, 3, 256, 256)])
def test_ssim_backward_kernel(dev, shape):
    """t2o_ssim_bwd (closed-form gradient of utils/ssim/__init__.py:20-40; the reference leaves it to autograd) against fp64
    autograd of the oracle's SSIM: both image gradients, per-sample output gradients, through torch.autograd on the device
    (`1 - ssim` as a loss).  1e-4 of the largest gradient entry: the statistics are differences of fp32 window sums."""
    import t2onet_amd.functional as T
    B = shape[0]
    a = synth.uniform(shape, 51)
    b = (a + synth.uniform(shape, 52, -0.1, 0.1)).clamp(0, 1)
    gout = synth.uniform((B,), 53, 0.5, 1.5)
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    (cpu_ref.ssim(a64, b64, size_average=False) * gout.double()).sum().backward()
    x, y = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    (T.ssim(x, y, size_average=False) * gout.to(dev)).sum().backward()
    for got, ref in ((x.grad, a64.grad), (y.grad, b64.grad)):
        scale = float(ref.abs().max())
        assert torch.isfinite(got).all()
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-4 * scale)
    # one-sided: only the prediction needs a gradient (the loss form); bitwise the same values, and run-to-run reproducible
    x2 = a.to(dev).requires_grad_(True)
    (1.0 - T.ssim(x2, b.to(dev))).backward()
    x3 = a.to(dev).requires_grad_(True)
    (1.0 - T.ssim(x3, b.to(dev))).backward()
    assert torch.equal(x2.grad, x3.grad)
    a64b = a.double().requires_grad_(True)
    (1.0 - cpu_ref.ssim(a64b, b.double())).backward()
    np.testing.assert_allclose(x2.grad.cpu().numpy(), a64b.grad.numpy(), rtol=0, atol=1e-4 * float(a64b.grad.abs().max()))


def test_ssim_backward_refuses_bad_arguments(dev):
    from t2onet_amd import _lib
    lib = _lib.load()
    a = torch.rand(1, 3, 8, 8, device=dev)
    g = torch.ones(1, device=dev)
    assert lib.t2o_ssim_bwd(a.data_ptr(), a.data_ptr(), g.data_ptr(), None, None, 1, 3, 8, 8, None) != 0
    assert lib.t2o_ssim_bwd(a.data_ptr(), a.data_ptr(), g.data_ptr(), a.data_ptr(), None, 1, 0, 8, 8, None) != 0


def test_config5_shape_fused_equals_materialised(executor, dev):
    """BASELINE config 5 per-GPU shape: 16 images 512x512, 8 operators (curve-heavy, repeats allowed)."""
    B, H, W = 16, 512, 512
    ops = [5, 3, 5, 3, 0, 1, 2, 6]
    g = torch.Generator().manual_seed(12)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
    params = torch.zeros(len(ops), B, 24)
    rng = {5: (8, .5, 1.5), 3: (24, .5, 1.5), 0: (1, -.3, .3), 1: (1, -.3, .3), 2: (1, -.3, .3), 6: (1, 0., 1.)}
    for k, op in enumerate(ops):
        n, lo, hi = rng[op]
        params[k, :, :n] = torch.rand(B, n, generator=g) * (hi - lo) + lo
    res = []
    for fn in (executor.run_sequence, executor.run_sequence_fused):
        x = img.clone().requires_grad_(True)
        p = params.to(dev).requires_grad_(True)
        loss, out = fn(x, ops, p, tgt)
        loss.backward()
        res.append((loss.item(), out if out.dim() == 4 else out[-1], x.grad, p.grad))
    (l0, o0, gx0, gp0), (l1, o1, gx1, gp1) = res
    assert torch.equal(o0, o1) and abs(l0 - l1) < 1e-7
    assert torch.allclose(gx0, gx1, rtol=1e-5, atol=1e-10)
    assert torch.allclose(gp0, gp1, rtol=5e-4, atol=5e-5 * max(1.0, gp0.abs().max().item()))
    assert o0.min().item() >= 0.0 and o0.max().item() <= 1.0


def test_sequence_step_is_graph_capturable(executor, dev):
    """The C ABI only enqueues work (no allocation, no sync): a whole fused forward+backward step
    captured in a HIP graph replays with bit-identical results."""
    import ctypes
    from t2onet_amd import _lib
    lib = _lib.load()
    B, H, W = 4, 64, 96
    ops = [0, 1, 2, 3, 5, 6]
    K = len(ops)
    c_ops = (ctypes.c_int * K)(*ops)
    img, tgt = synth.images(B, H, W, 1).to(dev), synth.images(B, H, W, 2).to(dev)
    params = torch.zeros(K, B, 24, device=dev)
    for k, op in enumerate(ops):
        params[k, :, :cpu_ref.OP_NPARAM[op]] = synth.op_params(op, B, 10 + k, 'mid').to(dev)
    nbuf = lib.t2o_fused_sequence_buffers(c_ops, K)
    seg = torch.empty(max(nbuf, 1), B, 3, H, W, device=dev)
    gbuf = torch.empty(2, B, 3, H, W, device=dev)
    out, gimg = torch.empty_like(img), torch.empty_like(img)
    gparams = torch.empty(K, B, 24, device=dev)
    loss, gloss = torch.zeros((), device=dev), torch.ones((), device=dev)
    ws = torch.empty(lib.t2o_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    def step():
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.t2o_fused_sequence_fwd(c_ops, K, P(img), P(params), P(tgt), P(out), P(loss), P(seg), P(ws), ws.numel(), B, H, W, st) == 0
        assert lib.t2o_fused_sequence_bwd(c_ops, K, P(img), P(params), P(tgt), P(gloss), None, P(gimg), P(gparams), P(seg), P(gbuf),
                                          P(ws), ws.numel(), B, H, W, st) == 0

    step()
    torch.cuda.synchronize()
    ref = (out.clone(), gimg.clone(), gparams.clone(), loss.clone())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()                                   # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for t in (out, gimg, gparams, loss):
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip((out, gimg, gparams, loss), ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize('op', OPS)
def test_quantized_images_ties_everywhere(executor, dev, op):
    """Coarsely quantised image (most pixels have equal channels / sit on knots / are 0 or 1):
    forward and gradients follow the reference's fp32 autograd, branch for branch."""
    from tests.test_block_programs_cpu import quantized_image, oracle_fwd_bwd
    img = quantized_image()
    B, _, H, W = img.shape
    gout = synth.uniform((B, 3, H, W), 96, -1.0, 1.0)
    p = synth.op_params(op, B, 800 + op, 'mid')
    o_ref, gi_ref, gp_ref = oracle_fwd_bwd(op, img, p, None, gout)
    x = img.to(dev).requires_grad_(True)
    pp = p.to(dev).requires_grad_(True)
    out, _ = executor.execute(x, op, None, specified_param=pp)
    out.backward(gout.to(dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), o_ref.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), gi_ref.numpy(), rtol=1e-4, atol=5e-5 if op in (0, 2) else 2e-6)
    np.testing.assert_allclose(pp.grad.cpu().numpy(), gp_ref.numpy(), rtol=1e-4, atol=1e-4 * max(1.0, float(gp_ref.abs().max())))


def test_forward_bit_identical_to_reference_arithmetic(executor, dev):
    """On the GPU too, the forward reproduces the reference's eager fp32 CPU path bit for bit for
    every operator except contrast (cos differs by an ulp between the two libms): same operation
    order, -ffp-contract=off, correctly rounded quotients."""
    img = synth.images(4, 64, 64, 5)
    for op in [0, 2, 3, 5, 6]:
        p = synth.op_params(op, 4, 50 + op, 'mid')
        ref = cpu_ref.operator_apply(op, img, p, None, OPT)
        out, _ = executor.execute(img.to(dev), op, None, specified_param=p.to(dev))
        assert torch.equal(out.cpu(), ref), 'operator %d: %d floats differ, max %.3g' % (
            op, int((out.cpu() != ref).sum()), float((out.cpu() - ref).abs().max()))
    p = synth.op_params(1, 4, 51, 'mid')
    ref = cpu_ref.operator_apply(1, img, p, None, OPT)
    out, _ = executor.execute(img.to(dev), 1, None, specified_param=p.to(dev))
    assert (out.cpu() - ref).abs().max().item() <= 2.4e-7


@pytest.mark.parametrize('B', [37, 150])             # 150: the weight-gradient kernels find an operator's samples 64 at a time
def test_fused_param_heads_match_library_gemms(dev, B):
    """t2o_param_heads_fwd/_bwd (each sample evaluates only its own operator's head) against the same heads through
    library GEMMs + gather (Executor.predict_params_gemm): parameters, feature gradient and all 28 head gradients,
    with identity / inpaint rows and heads that no sample selected."""
    import copy
    import t2onet_amd
    torch.manual_seed(5)
    ex_a = t2onet_amd.Executor(t2onet_amd.default_options()).to(dev)
    ex_b = copy.deepcopy(ex_a)
    feats = synth.uniform((B, 512), 801, -1.0, 1.0).to(dev)
    op_ids = torch.tensor([0, 1, 2, 3, 5, 6, -1, 4, 3, 5, 0, 2] * 13)[:B].to(dev).to(torch.int32)    # no sample uses op 7
    gout = synth.uniform((B, 24), 802, -1.0, 1.0).to(dev)
    fa, fb = feats.clone().requires_grad_(True), feats.clone().requires_grad_(True)
    pa = ex_a.predict_params(op_ids, fa)
    pb = ex_b.predict_params_gemm(op_ids, fb)
    np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=1e-5, atol=2e-6)
    assert float(pa[6].detach().abs().max()) == 0.0 and float(pa[7].detach().abs().max()) == 0.0          # identity, inpaint
    (pa * gout).sum().backward()
    (pb * gout).sum().backward()
    g = fb.grad.cpu().numpy()
    np.testing.assert_allclose(fa.grad.cpu().numpy(), g, rtol=1e-4, atol=1e-5 * np.abs(g).max())
    for (n, qa), (_, qb) in zip(ex_a.named_parameters(), ex_b.named_parameters()):
        if 'inpaint' in n:
            continue
        ga = torch.zeros_like(qa) if qa.grad is None else qa.grad
        gb = torch.zeros_like(qb) if qb.grad is None else qb.grad
        np.testing.assert_allclose(ga.cpu().numpy(), gb.cpu().numpy(), rtol=1e-4, atol=1e-5 * max(float(gb.abs().max()), 1e-6), err_msg=n)
    assert float(ex_a.white_op.fc1.weight.grad.abs().max()) == 0.0                      # unused head: exact zeros


def test_param_heads_add_their_gradients_into_existing_grad_tensors(dev):
    """Executor.heads_grad_in_place (set by the Trainer): the heads' backward kernel adds into the parameters' .grad
    tensors itself and autograd gets no gradient for them -- two backwards leave exactly twice the gradients of the
    ordinary path, the feature gradient is unchanged."""
    import copy
    import t2onet_amd
    torch.manual_seed(6)
    ex_a = t2onet_amd.Executor(t2onet_amd.default_options()).to(dev)
    ex_b = copy.deepcopy(ex_a)
    B = 70
    feats = synth.uniform((B, 512), 811, -1.0, 1.0).to(dev)
    op_ids = torch.tensor([0, 1, 2, 3, 5, 6, 7, -1, 4, 3] * 7)[:B].to(dev).to(torch.int32)
    gout = synth.uniform((B, 24), 812, -1.0, 1.0).to(dev)
    for p in ex_a.parameters():
        p.grad = torch.zeros_like(p)
    ex_a.__dict__['heads_grad_in_place'] = True
    fa, fb = feats.clone().requires_grad_(True), feats.clone().requires_grad_(True)
    for _ in range(2):
        (ex_a.predict_params(op_ids, fa) * gout).sum().backward()
    (ex_b.predict_params(op_ids, fb) * gout).sum().backward()
    assert torch.equal(fa.grad, 2 * fb.grad)
    for (n, qa), (_, qb) in zip(ex_a.named_parameters(), ex_b.named_parameters()):
        if 'inpaint' in n:
            continue
        want = 2 * (torch.zeros_like(qb) if qb.grad is None else qb.grad)
        np.testing.assert_allclose(qa.grad.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-6 * max(float(want.abs().max()), 1e-6), err_msg=n)


@pytest.mark.gpu
def test_runtime_specialised_chain_matches_runtime_loop_and_fp64_oracle(monkeypatch):
    """An operator order with no ahead-of-time chain kernel ([2,0,3,1,5] + sharpness; Executor.execute permits any order,
    executors/executor.py:33-55): first through the run-time-loop kernels (no specialisation), then after
    t2o_fused_sequence_prepare has compiled the list with hipRTC.  Images bit-identical, loss identical, parameter
    gradients equal up to summation order; and the specialised path against fp64 autograd of the oracle."""
    import t2onet_amd
    import t2onet_amd.functional as T
    from t2onet_amd import _lib
    from oracle import cpu_ref
    dev = torch.device('cuda:0')
    ops = [2, 0, 3, 1, 5, 6]
    B, H, W = 3, 40, 64
    img, tgt = synth.images(B, H, W, 301), synth.images(B, H, W, 302)
    ps = [synth.op_params(op, B, 310 + k, 'mid') for k, op in enumerate(ops)]
    params = torch.zeros(len(ops), B, 24)
    for k, p in enumerate(ps):
        params[k, :, :p.shape[1]] = p
    lib = _lib.load()

    def run():
        x = img.to(dev).requires_grad_(True)
        pr = params.to(dev).requires_grad_(True)
        loss, out = T.fused_sequence_l1(x, ops, pr, tgt.to(dev))
        loss.backward()
        return loss.item(), out.detach().cpu(), x.grad.cpu(), pr.grad.cpu()
    monkeypatch.setattr(T, '_CHAIN_JIT', False)
    T._prepared_chains.pop(tuple(ops), None)
    n0 = lib.t2o_jit_specialisations()
    base = run()
    assert lib.t2o_jit_specialisations() == n0                       # nothing was compiled: the run-time-loop kernels ran
    monkeypatch.setattr(T, '_CHAIN_JIT', True)
    if not T.prepare_fused_sequence(ops):
        pytest.skip('no libhiprtc on this machine: run-time specialisation unavailable')
    assert lib.t2o_jit_specialisations() == n0 + 1
    spec = run()
    assert lib.t2o_jit_specialisations() == n0 + 1                   # (cached: not compiled again)
    assert spec[0] == base[0] and torch.equal(spec[1], base[1])      # same arithmetic per pixel
    # (the two compilations order a few operations differently: a pixel on a hue-sector edge may take the other branch)
    _close_except_branch_flips(spec[2].numpy(), base[2].numpy(), 1e-5, 1e-9, max_frac=4e-4, what='gimg specialised vs run-time loop')
    np.testing.assert_allclose(spec[3].numpy(), base[3].numpy(), rtol=2e-5, atol=1e-7)
    # fp64 autograd of the oracle
    x64 = img.double().requires_grad_(True)
    p64 = [p.double().requires_grad_(True) for p in ps]
    ref, _ = cpu_ref.run_sequence(x64, ops, p64, cpu_ref.default_opt())
    l64 = (ref - tgt.double()).abs().mean()
    l64.backward()
    assert abs(spec[0] - l64.item()) < 1e-6
    gscale = float(x64.grad.abs().max())
    _close_except_branch_flips(spec[2].numpy(), x64.grad.float().numpy(), 2e-4, 2e-5 * gscale, max_frac=4e-4, what='gimg vs fp64')
    for k, p in enumerate(p64):
        got = spec[3][k, :, :p.shape[1]].numpy()
        # (a 40 x 64 image: one pixel on the other side of a hue-sector edge moves a one-parameter gradient by a few 1e-3)
        np.testing.assert_allclose(got, p.grad.float().numpy(), rtol=1e-2, atol=1e-3 * float(p.grad.abs().max()))
