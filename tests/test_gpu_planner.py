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


def test_multi_job_sweep_equals_single_job_launches(executor):
    """t2o_op_candidates_multi_l1: a whole beam step (several images x several operators) in one launch gives
    exactly the per-job kernel's numbers; the batched three-round fit returns the single-job fit's parameters."""
    import t2onet_amd.functional as T
    from t2onet_amd import planner
    H, W = 37, 53
    imgs = [synth.images(1, H, W, 40 + k).cuda() for k in range(3)]
    tgt = synth.images(1, H, W, 50).cuda()
    jobs = [(0, 0), (0, 1), (0, 2), (1, 0), (1, 2), (2, 1), (2, 5), (1, 3)]
    C = 19
    params = torch.stack([torch.cat([synth.op_params(op, C, 60 + k, 'mid'),
                                     torch.zeros(C, 24 - cpu_ref.OP_NPARAM[op])], 1) for k, (_, op) in enumerate(jobs)]).cuda()
    got = T.candidates_multi_l1([op for _, op in jobs], [i for i, _ in jobs], torch.cat(imgs), tgt, params)
    for k, (i, op) in enumerate(jobs):
        one = T.candidates_l1(op, imgs[i], tgt, params[k])
        assert torch.equal(got[k], one), (k, i, op)
    # batched fit == per-job fit (1-parameter operators)
    jobs1 = [(0, 0), (1, 1), (2, 2), (1, 0)]
    p, d = planner.fit_sweep_batch(imgs, jobs1, tgt, executor)
    for k, (i, op) in enumerate(jobs1):
        p1, _ = planner._fit_sweep_1d(imgs[i], tgt, op, executor)
        assert abs(float(p[k]) - float(p1)) < 1e-6
        out, _ = executor.execute(imgs[i], op, None, specified_param=p[k:k + 1])
        assert abs(planner.get_dist(out, tgt).item() - float(d[k])) < 1e-6
    with pytest.raises(RuntimeError):                                 # sharpness is a stencil: not a candidate-sweep operator
        T.candidates_multi_l1([6], [0], torch.cat(imgs), tgt, params[:1])
