"""Planner row: candidate sweep kernel vs the oracle, parameter recovery,
and a small beam search, all on the GPU."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref, synth

pytestmark = pytest.mark.gpu
OPT = cpu_ref.default_opt()


@pytest.fixture(scope='module')
def executor():
    import t2onet_amd
    return t2onet_amd.Executor(t2onet_amd.default_options()).to('cuda:0')


@pytest.mark.parametrize('op', [0, 1, 2, 3, 5, 7])
@pytest.mark.parametrize('shape', [(64, 64), (37, 53), (300, 450)])
def test_candidate_sweep_matches_oracle(op, shape):
    import t2onet_amd.functional as T
    H, W = shape
    img = synth.images(1, H, W, 11)
    tgt = synth.images(1, H, W, 12)
    C = 19
    params = synth.op_params(op, C, 13, 'mid')
    ref = torch.stack([cpu_ref.l1_loss(cpu_ref.operator_apply(op, img, params[c:c + 1], None, OPT), tgt) for c in range(C)])
    got = T.candidates_l1(op, img.cuda(), tgt.cuda(), params.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('op,true', [(0, 0.37), (1, -0.42), (2, 0.55)])
def test_sweep_recovers_parameter(executor, op, true):
    from t2onet_amd import planner
    img = synth.images(1, 96, 128, 21).cuda()
    p_true = torch.tensor([[true]], device='cuda')
    tgt, _ = executor.execute(img, op, None, specified_param=p_true)
    p, ok = planner.get_param(img, tgt, None, op, executor, None, 'L1', 'sweep')
    assert ok and abs(p.item() - true) < 2e-3
    p_nm, _ = planner.get_param(img, tgt, None, op, executor, None, 'L1', 'Nelder-Mead')       # the reference's procedure
    assert abs(p_nm.item() - true) < 2e-3


def test_curve_fit_and_beam_search(executor):
    from t2onet_amd import planner
    img = synth.images(1, 64, 64, 31).cuda()
    k_true = torch.tensor([[0.6, 0.8, 1.0, 1.2, 1.4, 1.2, 1.0, 0.8]], device='cuda')
    tgt, _ = executor.execute(img, 5, None, specified_param=k_true)
    p, _ = planner.get_param(img, tgt, None, 5, executor, None, 'L1', 'sweep')
    out, _ = executor.execute(img, 5, None, specified_param=p)
    assert planner.get_dist(out, tgt).item() < 3e-3                     # curve shape recovered (scale-invariant)
    # two-operator target: brightness then contrast
    mid, _ = executor.execute(img, 0, None, specified_param=torch.tensor([[0.25]], device='cuda'))
    tgt2, _ = executor.execute(mid, 1, None, specified_param=torch.tensor([[0.3]], device='cuda'))
    names = ['brightness', 'contrast', 'saturation', 'color', 'inpaint', 'tone', 'sharpness', 'white']
    actions, Is = planner.beam_search(img, tgt2, None, executor, None, 2, [0, 1, 2], names, 3, 1e-3, 'L1', 'sweep')
    best = actions[0]
    assert best[-1][2] < 5e-3                                            # final distance of the best sequence
    assert {a[0] for a in best} <= {'brightness', 'contrast', 'saturation'} and len(Is[0]) == len(best)
