"""Pin the oracle (oracle/cpu_ref.py) against outputs of the reference itself.

tests/golden/*.npz were produced by tools/gen_golden.py, which imports
/root/reference in the build container (the reference has no tests or golden
vectors of its own: SURVEY.md section 4).  Inputs/weights are regenerated from
oracle/synth.py formulas; only reference OUTPUTS are stored.
Tolerances: the oracle runs the same eager CPU ops in the same order, so the
forward comparisons are (near) bit-exact; stated per assert.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref, synth

OPT = cpu_ref.default_opt()


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'operators.npz'))


@pytest.fixture(scope='module')
def executor_sd():
    # same key order as the reference Executor.state_dict()
    order = ['brightness_op', 'sharpness_op', 'color_op', 'contrast_op', 'inpaint_op',
             'white_op', 'saturation_op', 'tone_op']            # executor.py:22-29 registration order
    nout = dict(zip(cpu_ref.OP_ATTRS, cpu_ref.OP_NPARAM))
    sd = {}
    for name in order:
        sd['%s.fc1.weight' % name] = torch.zeros(512, 512)
        sd['%s.fc1.bias' % name] = torch.zeros(512)
        sd['%s.fc2.weight' % name] = torch.zeros(nout[name], 512)
        sd['%s.fc2.bias' % name] = torch.zeros(nout[name])
    return {'executor.' + k: v for k, v in synth.fill_state_dict(sd, seed=3).items()}


def test_executor_metadata(gold):
    assert list(gold['name_list']) == cpu_ref.OP_NAMES
    assert list(gold['param_num']) == cpu_ref.OP_NPARAM
    for i in range(8):
        np.testing.assert_allclose(gold['param_bnd'][i], cpu_ref.param_range(i, OPT), rtol=0, atol=1e-12)


@pytest.mark.parametrize('op', [0, 1, 2, 3, 5, 6, 7])
def test_operator_forward_backward(gold, op):
    B, H, W = 2, 24, 20
    img = synth.images(B, H, W, 11)
    gout = synth.uniform((B, 3, H, W), 12, -1.0, 1.0)
    masks = {'none': None, 'm1': synth.masks(B, 1, H, W, 14), 'm3': synth.masks(B, 3, H, W, 15, soft=False)}
    for si, setting in enumerate(['mid', 'strong', 'neg']):
        for mname, mask in masks.items():
            key = 'op%d_%s_%s' % (op, setting, mname)
            if key + '_out' not in gold:
                continue
            x = img.clone().requires_grad_(True)
            p = synth.op_params(op, B, 100 + 10 * op + si, setting).requires_grad_(True)
            out, par = cpu_ref.executor_execute(None, x, op, mask, OPT, specified_param=p)
            out.backward(gout)
            # same ATen ops in the same order -> bit-exact forward, 1e-6 on grads
            np.testing.assert_array_equal(out.detach().numpy(), gold[key + '_out'], err_msg=key)
            np.testing.assert_allclose(x.grad.numpy(), gold[key + '_gimg'], rtol=1e-6, atol=1e-6, err_msg=key)
            gp = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
            np.testing.assert_allclose(gp, gold[key + '_gparam'], rtol=1e-5, atol=1e-5, err_msg=key)


@pytest.mark.parametrize('op', [0, 1, 2, 3, 5, 6, 7])
def test_param_heads(gold, executor_sd, op):
    B, H, W = 2, 24, 20
    img = synth.images(B, H, W, 11)
    f = synth.uniform((B, 512), 13, -1.0, 1.0).requires_grad_(True)
    out, par = cpu_ref.executor_execute(executor_sd, img, op, None, OPT, features=f)
    np.testing.assert_allclose(par.detach().numpy(), gold['op%d_feat_param' % op], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out.detach().numpy(), gold['op%d_feat_out' % op], rtol=1e-6, atol=1e-6)
    if out.requires_grad:
        out.backward(synth.uniform((B, 3, H, W), 12, -1.0, 1.0))
        np.testing.assert_allclose(f.grad.numpy(), gold['op%d_feat_gfeat' % op], rtol=1e-4, atol=1e-5)


def test_identity(gold):
    img = synth.images(2, 24, 20, 11)
    out, par = cpu_ref.executor_execute(None, img, -1, None, OPT, features=img)
    assert out is img and bool(gold['identity_same_object'])
    np.testing.assert_array_equal(par.numpy(), gold['identity_param'])


def test_cfg1_three_op_chain(gold):
    """BASELINE config 1: one 256x256 image, brightness -> contrast -> saturation."""
    x = synth.images(1, 256, 256, 21)
    ps = [synth.op_params(op, 1, 200 + k, 'mid') for k, op in enumerate([0, 1, 2])]
    out, _ = cpu_ref.run_sequence(x, [0, 1, 2], ps, OPT)
    np.testing.assert_array_equal(out[:, :, 100:132, 60:92].numpy(), gold['cfg1_out_crop'])
    assert abs(out.double().sum().item() - float(gold['cfg1_out_sum'])) < 1e-9


def test_chain6_l1_backward(gold):
    B, H, W = 3, 32, 40
    ops = [0, 1, 2, 3, 5, 6]
    x = synth.images(B, H, W, 31).requires_grad_(True)
    tgt = synth.images(B, H, W, 32)
    ps = [synth.op_params(op, B, 300 + k, 'mid').requires_grad_(True) for k, op in enumerate(ops)]
    out, _ = cpu_ref.run_sequence(x, ops, ps, OPT)
    loss = cpu_ref.l1_loss(out, tgt)
    loss.backward()
    np.testing.assert_array_equal(out.detach().numpy(), gold['chain6_out'])
    assert abs(loss.item() - float(gold['chain6_loss'])) < 1e-7
    np.testing.assert_allclose(x.grad.numpy(), gold['chain6_gimg'], rtol=1e-5, atol=1e-9)
    for k, p in enumerate(ps):
        np.testing.assert_allclose(p.grad.numpy(), gold['chain6_gparam%d' % k], rtol=1e-4, atol=1e-7)


def test_ssim(golden_dir):
    g = np.load(os.path.join(golden_dir, 'ssim.npz'))
    a = synth.images(1, 48, 40, 51)
    b = (a + synth.uniform((1, 3, 48, 40), 52, -0.1, 0.1)).clamp(0, 1)
    assert abs(cpu_ref.ssim(a, b).item() - float(g['ssim'])) < 1e-6


def test_hsv_epsilon_variant_fixture(golden_dir, monkeypatch):
    """hsv_eps.npz: the reference's brightness / saturation with the HSV shim at eps = 1e-8 (current kornia) instead of the
    1e-6 this build's spec fixes (oracle/hsv_spec.py; SURVEY 8(c): kornia is unpinned).  The oracle with the same epsilon
    reproduces the fixture; the distance between the two epsilons -- what a maintainer on a current kornia should expect
    against this implementation -- stays below the figures DESIGN.md section 2 quotes: 3e-6 on outputs in [0, 1],
    1e-5 of the largest entry on image gradients of ordinary images, 5e-4 on a dark image (values <= 0.02)."""
    from oracle import hsv_spec
    g = np.load(os.path.join(golden_dir, 'hsv_eps.npz'))
    B, H, W = 2, 24, 20
    imgs = {'std': synth.images(B, H, W, 11), 'dark': synth.images(B, H, W, 11) * 0.02}
    gout = synth.uniform((B, 3, H, W), 12, -1.0, 1.0)

    def run(eps):
        monkeypatch.setattr(hsv_spec, 'HSV_EPS', eps)
        res = {}
        for iname, img in imgs.items():
            for op in (0, 2):
                for si, setting in enumerate(['mid', 'strong', 'neg']):
                    x = img.clone().requires_grad_(True)
                    p = synth.op_params(op, B, 100 + 10 * op + si, setting).requires_grad_(True)
                    out = cpu_ref.operator_apply(op, x, p, None, OPT)
                    out.backward(gout)
                    res['%s_op%d_%s' % (iname, op, setting)] = (out.detach().numpy(), x.grad.numpy(), p.grad.numpy())
        return res
    r8, r6 = run(1e-8), run(1e-6)
    for k, (out, gimg, gpar) in r8.items():
        np.testing.assert_array_equal(out, g[k + '_eps8_out'])
        np.testing.assert_allclose(gimg, g[k + '_eps8_gimg'], rtol=1e-6, atol=1e-6 * np.abs(g[k + '_eps8_gimg']).max())
        np.testing.assert_allclose(gpar, g[k + '_eps8_gparam'], rtol=1e-5, atol=1e-6 * max(np.abs(g[k + '_eps8_gparam']).max(), 1e-6))
        assert np.abs(out - r6[k][0]).max() <= 3e-6
        rel = np.abs(gimg - r6[k][1]).max() / np.abs(r6[k][1]).max()
        assert rel <= (5e-4 if k.startswith('dark') else 1e-5), (k, rel)
    for name in ('std_op0', 'std_op2', 'dark_op0', 'dark_op2'):
        worst = max(np.abs(r8[k][0] - r6[k][0]).max() for k in r8 if k.startswith(name))
        assert abs(worst - float(g['max_out_delta_' + name])) < 1e-9
