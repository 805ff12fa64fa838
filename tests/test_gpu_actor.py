"""The actor on the GPU -- every layer on this library's own fp32 kernels (encoder trunk, request-encoder LSTM, decoder step,
attention, parameter heads, operators) -- against outputs of the reference itself (tests/golden/actor.npz from
tools/gen_golden.py: the reference's CPU run).

Operator indices must be identical (argmax mode).  Floating-point tolerances follow the distances measured with
tools/measure_parity.py (round 5, profiles/r05_parity_distances.txt), per mode -- ~3x in evaluation mode, where the figures
are the same on every box; 5-10x in training mode (until round 6 this file ran the encoder in NCHW = framework convolutions,
whose per-machine algorithm choice made them box dependent: image crops 1.3e-5, 2.6e-5 and 4.1e-5 on three boxes; with the
channels-last default every layer is this library's and the figures repeat bit for bit):

                      evaluation mode            training mode (batch statistics over B = 4)
  pred_params         3.4e-8 .. 2.4e-7           5.7e-6 / 4.8e-7        (episode / teacher-forced)
  image crops         4.1e-6 / 5.0e-6            6.5e-5 / 1.3e-5
  log-probabilities   1.2e-6                     1.5e-5
  losses              <= 2.4e-7                  <= 1.4e-6
  gradient norms      --                         9.2e-5 .. 1.3e-2 (three boxes) / 1.7e-5 relative

Training mode is looser than evaluation mode for a structural reason, not a library one: BatchNorm over 4 samples (and
BatchNorm1d over 4 VALUES per feature) divides by a standard deviation formed from a handful of numbers, which amplifies the
last-bit differences of the convolutions in front of it.  The hand-written kernels themselves are pinned at 1e-5 .. 1e-6 in
test_gpu_operators.py / test_gpu_conv.py / test_gpu_decoder.py."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref, synth

pytestmark = pytest.mark.gpu
B, H, W, L = 4, 64, 64, 17


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'actor.npz'))


def make_model(dev):
    import t2onet_amd
    from t2onet_amd.actor import Actor
    torch.backends.cudnn.deterministic = True
    opt = t2onet_amd.default_options(input_dropout_p=0.0, dropout_p=0.0)
    m = Actor(opt)
    m.load_state_dict(synth.fill_state_dict(m.state_dict(), seed=7))
    return m.to(dev), opt


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_episode_l1_step_matches_reference(gold, mode):
    from t2onet_amd.train import select_end_images
    import t2onet_amd.functional as T
    dev = torch.device('cuda:0')
    model, opt = make_model(dev)
    model.train(mode == 'train')
    p = 'ep_%s_' % mode
    x = synth.requests(B, L, 41).to(dev)
    img = synth.images(B, H, W, 42).to(dev)
    tgt = synth.images(B, H, W, 43).to(dev)
    state, pred_imgs, pred_ops, pred_params = model.episode_forward(x, img, None, reinforce_sample=0)
    np.testing.assert_array_equal(pred_ops.cpu().numpy(), gold[p + 'pred_ops'])             # bit-exact indices
    tol = ({'params': 1e-6, 'crop': 1.5e-5, 'mean': 1.5e-6, 'loss': 1e-6} if mode == 'eval'
           else {'params': 5e-5, 'crop': 5e-4, 'mean': 2e-5, 'loss': 5e-6})
    np.testing.assert_allclose(torch.stack(pred_params, 0).detach().cpu().numpy(), gold[p + 'pred_params'], rtol=1e-5, atol=tol['params'])
    np.testing.assert_allclose(pred_imgs[:, :, :, 8:24, 8:24].detach().cpu().numpy(), gold[p + 'imgs_crop'], rtol=0, atol=tol['crop'])
    np.testing.assert_allclose(pred_imgs.detach().double().mean((2, 3, 4)).cpu().numpy(), gold[p + 'imgs_mean'], rtol=0, atol=tol['mean'])
    assert state['imgs'].shape == pred_imgs.shape and len(state['hidden']) == 6 and state['masks'] is None
    loss = T.l1_loss(select_end_images(pred_imgs, pred_ops, opt.end_id), tgt)
    assert abs(loss.item() - float(gold[p + 'loss'])) < tol['loss']                             # L1 deviation (north star: <= 1e-5)
    if mode == 'eval':
        return
    loss.backward()
    names = list(gold['param_names'])
    params = dict(model.named_parameters())
    gn = np.array([0.0 if params[n].grad is None else params[n].grad.double().norm().item() for n in names])
    ref = gold[p + 'grad_norm']
    big = ref > 1e-3 * ref.max()
    # Rounds 4-5 saw 9.2e-5, 5.7e-3 and 1.3e-2 here on three boxes and blamed the request encoder's library GEMMs.  Round 6 took
    # every library GEMM AND every framework convolution out of this path (t2o_gemm; the encoder channels-last by default -- this
    # test had been running MIOpen convolutions, whose algorithm choice is per machine): the figure is now the SAME on every box,
    # 5.618e-3 with identical gradient bits on four leases (profiles/r06_parity_distances.txt).  It is the fixture that carries
    # it, not an implementation: against the oracle in fp64 the golden (the reference's fp32 run) is 6e-5 away, this library
    # 5.6e-3 -- and the oracle ITSELF run in fp32 5.5e-5 on the GPU box's host but 6.5e-3 in the build container.  Four images,
    # batch statistics over 4 values in front of ReLUs: last-bit differences of any fp32 execution come out as ~0.5 % of every
    # gradient norm.  The as-trained fixtures (extra2.npz: test_as_trained_*) hold gradients ELEMENTWISE against fp64.
    np.testing.assert_allclose(gn[big], ref[big], rtol=1e-2)
    # heads of unused operators get zeros here (gather over all heads) where the reference has None
    none_ref = gold[p + 'grad_none']
    assert np.all(gn[none_ref] == 0.0)


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_supervised_step_matches_reference(gold, mode):
    dev = torch.device('cuda:0')
    model, opt = make_model(dev)
    model.train(mode == 'train')
    p = 'sup_%s_' % mode
    x = synth.requests(B, L, 41).to(dev)
    img = synth.images(B, H, W, 42).to(dev)
    y = synth.op_targets(B, 45)
    img_y = synth.uniform((B, 6, 3, H, W), 46).to(dev)
    gt_params = synth.uniform((B, 5, 24), 47, -1, 1)
    nparam = {3: 1, 4: 1, 5: 1, 6: 24, 8: 8, 9: 1}
    for b in range(B):
        for k in range(5):
            gt_params[b, k, nparam[int(y[b, k + 1])]:] = 0
    y, gt_params = y.to(dev), gt_params.to(dev)
    pred_imgs, pred_params, logp = model.supervised_forward(x, y, img, img_y, gt_params, None)
    tol = ({'params': 1e-6, 'logp': 5e-6, 'crop': 1.5e-5, 'loss': 1e-6} if mode == 'eval'
           else {'params': 1e-5, 'logp': 1e-4, 'crop': 2e-4, 'loss': 1e-5})
    np.testing.assert_allclose(pred_params.detach().cpu().numpy(), gold[p + 'pred_params'], rtol=1e-5, atol=tol['params'])
    np.testing.assert_allclose(logp.detach().cpu().numpy(), gold[p + 'logprobs'], rtol=1e-5, atol=tol['logp'])
    np.testing.assert_allclose(pred_imgs[:, :, :, 8:24, 8:24].detach().cpu().numpy(), gold[p + 'imgs_crop'], rtol=0, atol=tol['crop'])
    op_loss, param_loss = cpu_ref.supervised_loss(pred_params, logp, y, gt_params, opt)
    assert abs(op_loss.item() - float(gold[p + 'op_loss'])) < tol['loss']
    assert abs(param_loss.item() - float(gold[p + 'param_loss'])) < tol['loss']
    if mode == 'eval':
        return
    (op_loss + param_loss).backward()
    names = list(gold['param_names'])
    params = dict(model.named_parameters())
    gn = np.array([0.0 if params[n].grad is None else params[n].grad.double().norm().item() for n in names])
    ref = gold[p + 'grad_norm']
    big = ref > 1e-3 * ref.max()
    np.testing.assert_allclose(gn[big], ref[big], rtol=5e-4)                                      # (measured 1.2e-5 .. 1.7e-5)


def test_trainer_alternates_and_learns():
    """Two reference-style iterations (supervised then episode) run end to end and change the weights."""
    from t2onet_amd.train import Trainer
    dev = torch.device('cuda:0')
    model, opt = make_model(dev)
    model.train()
    tr = Trainer(model, opt)
    x = synth.requests(B, L, 41).to(dev)
    img = synth.images(B, H, W, 42).to(dev)
    y = synth.op_targets(B, 45).to(dev)
    img_y = synth.uniform((B, 6, 3, H, W), 46).to(dev)
    gt = synth.uniform((B, 5, 24), 47, -1, 1).to(dev)
    w0 = model.decoder.out_linear.weight.detach().clone()
    op_loss, param_loss = tr.step((img, img_y, x, y, gt))
    l1 = tr.step((img, img_y, x, y, gt))
    assert torch.isfinite(op_loss) and torch.isfinite(param_loss) and torch.isfinite(l1)
    assert not torch.equal(w0, model.decoder.out_linear.weight.detach())
    assert sum(p.numel() for p in tr.grads.params) == 22165917 and tr.grads.flat.numel() >= 22165917      # (+ segment padding)


def test_backward_outside_the_trainer_after_a_trainer_step():
    """the Trainer's persistent decoder tape and weight-gradient arena are installed for the duration of a train
    step only.  After a step, a grad-enabled forward + loss.backward() OUTSIDE the Trainer (a custom loss, a gradient check)
    must deliver every gradient -- with the tape / arena left installed the decoder, fc, bn1 and convolution weight
    gradients were deferred into storage nobody flushes.  Checked against a second model without any Trainer."""
    from t2onet_amd.train import Trainer
    dev = torch.device('cuda:0')
    model, opt = make_model(dev)
    model.train()
    tr = Trainer(model, opt)
    x = synth.requests(B, L, 41).to(dev)
    img = synth.images(B, H, W, 42).to(dev)
    tgt = synth.images(B, H, W, 43).to(dev)
    y = synth.op_targets(B, 45).to(dev)
    img_y = synth.uniform((B, 6, 3, H, W), 46).to(dev)
    gt = synth.uniform((B, 5, 24), 47, -1, 1).to(dev)
    tr.episode_step(x, img, tgt, reinforce_sample=0)
    tr.supervised_step(x, y, img, img_y, gt)
    assert '_tape' not in model.__dict__ and '_tape' not in model.decoder.__dict__
    assert model.vis_encoder.trunk_plan().__dict__.get('arena') is None
    assert len(tr._arenas) <= 1                              # at most one arena per batch shape, whatever the number of passes (none is
                                                             # remembered for a shape the trunk does not defer: ADVICE r5)
    other, _ = make_model(dev)
    other.load_state_dict(model.state_dict())
    other.train()

    def custom_loss(m):
        _, pred_params, logp = m.supervised_forward(x, y, img, img_y, gt, None, None)
        return (logp * logp).sum() + pred_params.abs().sum()
    tr.grads.zero()
    custom_loss(model).backward()
    custom_loss(other).backward()
    ref = dict(other.named_parameters())
    checked = 0
    for n, p in model.named_parameters():
        g_ref = ref[n].grad
        if g_ref is None or float(g_ref.abs().max()) == 0.0:
            continue
        scale = float(g_ref.abs().max())
        assert float((p.grad - g_ref).abs().max()) <= 3e-3 * scale, n
        checked += 1
    assert checked >= 110                                    # decoder, encoder trunk, request encoder, the used heads (123 at this seed)
    # and the Trainer still works afterwards
    assert torch.isfinite(tr.episode_step(x, img, tgt))


def test_evaluation_loop_full_resolution():
    """test() of test_seq2seqL1.py at inference shapes (bs=1, non-square, short side 600 -> here a
    smaller 150x225 so the test stays quick): runs end to end and reports L1 / SSIM."""
    from t2onet_amd.evaluate import test as evaluate
    dev = torch.device('cuda:0')
    model, opt = make_model(dev)
    opt.print_every = 100
    batches = []
    for i in range(3):
        x = synth.requests(1, L, 60 + i)
        img_x = synth.images(1, 150, 225, 70 + i)
        img_y = (img_x + synth.uniform((1, 3, 150, 225), 80 + i, -0.05, 0.05)).clamp(0, 1)
        batches.append((img_x, img_y, x, ['req']))
    init_d, d = evaluate(model, batches, opt, is_test=True, verbose=False)
    ref_init = float(np.mean([(a - b).abs().mean().item() for a, b, _, _ in batches]))
    assert abs(init_d - ref_init) < 1e-6 and 0.0 <= d <= 1.0


def test_train_cli_runs_and_checkpoints(tmp_path):
    """Four iterations of the reference-shaped loop on synthetic FiveK batches; the checkpoint has the
    reference's 199-tensor layout and reloads."""
    from t2onet_amd import train_cli, default_options
    from t2onet_amd.actor import Actor
    avg = train_cli.main(['--synthetic', '--batch_size', '4', '--img_size', '64', '--num_iters', '4', '--print_every', '2',
                          '--checkpoint_every', '4', '--run_dir', str(tmp_path), '--num_workers', '0'])
    assert avg['fs_t'] > 0 and avg['l1_t'] > 0
    path = tmp_path / 'seq2seqL1_model' / 'checkpoint_iter00000004' / 'model.pth'
    sd = torch.load(str(path))
    assert len(sd) == 199
    Actor(default_options()).load_state_dict(sd)
    # evaluation ran at the checkpoint and the best model was kept (train_seq2seqL1.py:105-131)
    st = avg['stats']
    assert st['train_iter'] == [4] and len(st['val_dist']) == 1 and st['best_iter'] == 4 and 0 < st['best_val_dist'] < 1
    assert (tmp_path / 'seq2seqL1_model' / 'checkpoint_best' / 'model.pth').exists()
    # real data without the GloVe table is refused (the word rows would stay frozen at random values)
    with pytest.raises(SystemExit):
        train_cli.main(['--batch_size', '4', '--num_iters', '1', '--run_dir', str(tmp_path)])


def test_train_cli_on_a_fivek_layout_tree(tmp_path):
    """The reference-shaped loop on REAL-data plumbing (no --synthetic): a FiveK-layout tree written here (JPEGs, planner
    records, annotation JSONs, a GloVe table), FiveKAct for training, the FiveK validation split at full resolution
    (short side 600, one image per batch: train_seq2seqL1.py:155-156), checkpoint + checkpoint_best."""
    from t2onet_amd import train_cli
    from tests import fivek_tree
    img_dir, anno_dir, act_dir, glove = fivek_tree.write_tree(str(tmp_path / 'data'), n_train=8, n_val=2)
    avg = train_cli.main(['--img_dir', img_dir, '--anno_dir', anno_dir, '--act_dir', act_dir, '--word2vec', glove,
                          '--batch_size', '4', '--img_size', '64', '--num_iters', '4', '--print_every', '2',
                          '--checkpoint_every', '4', '--run_dir', str(tmp_path / 'run'), '--num_workers', '0'])
    st = avg['stats']
    assert st['train_iter'] == [4] and len(st['val_dist']) == 1 and 0 < st['best_val_dist'] < 1
    sd = torch.load(str(tmp_path / 'run' / 'seq2seqL1_model' / 'checkpoint_best' / 'model.pth'))
    assert len(sd) == 199
    # the GloVe rows are frozen (fix_input_embedding=1, lang_encoder.py:22-31): still the table's values after 4 steps
    import numpy as np
    table = torch.from_numpy(np.load(glove))
    assert torch.equal(sd['lang_encoder.embedding.weight'][4:].cpu(), table)


def test_one_rank_nccl_group_gives_the_ungrouped_run(tmp_path):
    """The N > 1 code path that can run on one GPU: a 1-rank `nccl` (RCCL) process group with the encoder hipGraphs ON.
    Three episode steps (arg-max operators, no dropout) inside the group must give bit-identical losses to the same
    steps without a group: the flat all-reduce is skipped at world size 1, and neither the communicator's watchdog
    thread nor its stream may disturb the captured graphs."""
    import copy
    import torch.distributed as dist
    import t2onet_amd
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Trainer
    dev = torch.device('cuda:0')
    opt = t2onet_amd.default_options(input_dropout_p=0.0, dropout_p=0.0)
    torch.manual_seed(51)
    base = Actor(opt).to(dev).train()
    base.use_channels_last()
    Bn = 8
    img = synth.images(Bn, 256, 256, 111).to(dev)
    tgt = synth.images(Bn, 256, 256, 112).to(dev)
    x = synth.requests(Bn, 17, 113).to(dev)
    lengths = (x != 0).sum(1).cpu()

    def run():
        tr = Trainer(copy.deepcopy(base), opt, graph_encoder=True)
        return [float(tr.episode_step(x, img, tgt, reinforce_sample=0, lengths=lengths)) for _ in range(3)]
    plain = run()
    dist.init_process_group('nccl', init_method='file://%s' % (tmp_path / 'rdv'), rank=0, world_size=1, device_id=dev)
    try:
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                                   # the communicator exists and works
        assert float(t.sum()) == 4.0
        grouped = run()
    finally:
        dist.destroy_process_group()
    assert grouped == plain, (grouped, plain)


def test_c_abi_allreduce_on_an_own_one_rank_communicator():
    """SURVEY 8(b)'s t2o_allreduce(float*, size_t, ncclComm_t, hipStream_t): a communicator built through the C ABI
    (t2o_comm_unique_id / t2o_comm_init_rank, RCCL resolved at run time), one rank -- all a 1-GPU box can run.  Sum and
    mean of one rank are the identity, bit for bit, stream-ordered with neighbouring kernels; the Trainer's flat gradient
    buffer goes through it (`use_own_communicator`) and the step equals the step without any communicator."""
    import copy
    import t2onet_amd
    from t2onet_amd import _lib
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Communicator, Trainer
    dev = torch.device('cuda:0')
    lib = _lib.load()
    assert lib.t2o_comm_available() == 1
    comm = Communicator(dev)
    assert comm.nranks == 1 and comm.rank == 0 and comm.handle
    t = synth.uniform((22165917,), 5, -1.0, 1.0).to(dev)
    ref = t.clone()
    u = t * 2.0                                              # (a producer on the same stream right before ...)
    comm.all_reduce_(u)
    comm.all_reduce_(u, mean=True)
    v = u * 0.5                                              # (... and a consumer right after: stream order must hold)
    assert torch.equal(v, ref)
    assert lib.t2o_allreduce(None, 4, comm.handle, None) != 0 and lib.t2o_allreduce(t.data_ptr(), 4, None, None) != 0
    opt = t2onet_amd.default_options(input_dropout_p=0.0, dropout_p=0.0)
    torch.manual_seed(51)
    base = Actor(opt).to(dev).train()
    base.use_channels_last()
    Bn = 4
    img, tgt = synth.images(Bn, 64, 64, 111).to(dev), synth.images(Bn, 64, 64, 112).to(dev)
    x = synth.requests(Bn, 17, 113).to(dev)
    lengths = (x != 0).sum(1).cpu()

    def run(own):
        tr = Trainer(copy.deepcopy(base), opt)
        if own:
            tr.grads._comm = comm
        return [float(tr.episode_step(x, img, tgt, reinforce_sample=0, lengths=lengths)) for _ in range(3)]
    assert run(True) == run(False)
    comm.close()


def test_episode_with_local_edit_masks():
    """mask_dict path (actor.py:78-98, :238-239): samples with a mask for the chosen operator are
    edited only inside it; others globally.  Checked against the unmasked run and the blend identity."""
    dev = torch.device('cuda:0')
    model, opt = make_model(dev)
    model.eval()
    x = synth.requests(B, L, 41).to(dev)
    img = synth.images(B, H, W, 42).to(dev)
    with torch.no_grad():
        _, imgs0, ops0, _ = model.episode_forward(x, img, None, reinforce_sample=0)
        first = ops0[:, 0].tolist()
        box = np.zeros((1, H, W), np.float32)
        box[:, 16:48, 16:48] = 1.0
        # sample 0 and 2 carry a mask for their first chosen operator; 1 and 3 do not
        mask_dict = [{str(first[0]): [box]}, {}, {str(first[2]): [box]}, {}]
        state, imgs1, ops1, _ = model.episode_forward(x, img, mask_dict, reinforce_sample=0)
    assert state['masks'].shape == (B, 5, 3, H, W)
    assert torch.equal(ops1[:, 0], ops0[:, 0])
    m = torch.from_numpy(box).to(dev)
    for b in (0, 2):                      # masked: inside = the global edit, outside = the input (clamped blend)
        want = (imgs0[b, 0] * m + img[b] * (1 - m)).clamp(0, 1)
        assert torch.allclose(imgs1[b, 0], want, atol=1e-6)
    for b in (1, 3):
        assert torch.equal(imgs1[b, 0], imgs0[b, 0])


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(4, 8, 6, 6), (2, 64, 16, 16), (3, 5, 5, 5), (64, 64, 32, 32), (5, 512, 3, 7), (7, 128, 9, 5)])
@pytest.mark.parametrize('with_res', [False, True])
@pytest.mark.parametrize('nhwc', [False, True])
def test_fused_batchnorm_relu_matches_torch(shape, with_res, nhwc):
    """t2o_bn_relu_fwd / _bwd (training-mode BatchNorm2d + residual add + ReLU of the image encoder,
    models/actor_resnet.py:38-44) against PyTorch's own batch norm in fp64 on the CPU: output,
    running statistics, num_batches_tracked and every gradient."""
    import torch.nn as nn
    import torch.nn.functional as F
    import t2onet_amd.functional as T
    from oracle import synth
    dev = torch.device('cuda:0')
    N, C, H, W = shape
    x = synth.uniform(shape, 301, -2.0, 3.0)
    res = synth.uniform(shape, 302, -1.0, 1.0) if with_res else None
    gout = synth.uniform(shape, 303, -1.0, 1.0)
    bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(synth.uniform((C,), 304, 0.5, 1.5))
        bn.bias.copy_(synth.uniform((C,), 305, -0.5, 0.5))
        bn.running_mean.copy_(synth.uniform((C,), 306, -0.1, 0.1))
        bn.running_var.copy_(synth.uniform((C,), 307, 0.5, 1.5))
    import copy
    ref = copy.deepcopy(bn).double().train()
    x64 = x.double().requires_grad_(True)
    r64 = None if res is None else res.double().requires_grad_(True)
    pre = ref(x64)
    y_ref = F.relu(pre if r64 is None else pre + r64)
    y_ref.backward(gout.double())

    g = copy.deepcopy(bn).to(dev).train()
    fmt = torch.channels_last if nhwc else torch.contiguous_format     # channels-last: the t2o_bn_relu_nhwc_* kernels
    xg = x.to(dev).contiguous(memory_format=fmt).requires_grad_(True)
    rg = None if res is None else res.to(dev).contiguous(memory_format=fmt).requires_grad_(True)
    y = T.batch_norm_relu(xg, g, rg)
    if nhwc and C >= 4 and (C & (C - 1)) == 0:
        assert T._is_nhwc(xg) and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gout.to(dev).contiguous(memory_format=fmt))
    # elements whose pre-activation is within rounding of 0 may take the other ReLU branch
    safe = (pre if r64 is None else pre + r64).detach().abs() > 1e-5
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().float().numpy(), rtol=1e-5, atol=2e-6)
    assert bool(safe.float().mean() > 0.999)
    np.testing.assert_allclose(g.running_mean.cpu().numpy(), ref.running_mean.float().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g.running_var.cpu().numpy(), ref.running_var.float().numpy(), rtol=1e-5, atol=1e-6)
    assert int(g.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    scale = float(x64.grad.abs().max())
    np.testing.assert_allclose(xg.grad.cpu().numpy(), x64.grad.float().numpy(), rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(g.weight.grad.cpu().numpy(), ref.weight.grad.float().numpy(), rtol=1e-4,
                               atol=1e-5 * float(ref.weight.grad.abs().max()))
    np.testing.assert_allclose(g.bias.grad.cpu().numpy(), ref.bias.grad.float().numpy(), rtol=1e-4,
                               atol=1e-5 * float(ref.bias.grad.abs().max()))
    if with_res:
        np.testing.assert_allclose(rg.grad.cpu().numpy(), r64.grad.float().numpy(), rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_channels_last_encoder_matches_torch_batchnorm():
    """The whole image encoder in channels-last mode (fused batch norm with and without the activation -- the
    shortcut branch --, one multi-tensor launch for the 21 num_batches_tracked counters) against the same module
    with PyTorch's own batch norm: output, input gradient, running statistics, counters."""
    import copy
    import t2onet_amd.actor_resnet as R
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    net = R.ResNet().to(dev).train()
    x = synth.images(4, 64, 64, 91).to(dev)
    gy = synth.uniform((4, 512), 92, -1.0, 1.0).to(dev)
    ref = copy.deepcopy(net)
    fast = copy.deepcopy(net).to(memory_format=torch.channels_last)
    saved = R._FUSED
    try:
        R._FUSED = False
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        yr.backward(gy)
        R._FUSED = True
        xf = x.clone().requires_grad_(True)
        yf = fast(xf)
        yf.backward(gy)
    finally:
        R._FUSED = saved
    np.testing.assert_allclose(yf.detach().cpu().numpy(), yr.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    g = xr.grad.cpu().numpy()
    np.testing.assert_allclose(xf.grad.cpu().numpy(), g, rtol=1e-3, atol=1e-3 * np.abs(g).max())
    for (n, a), (_, b) in zip(fast.named_buffers(), ref.named_buffers()):
        if n.endswith('num_batches_tracked'):
            assert int(a) == int(b) == 1, n
        else:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=n)
    for (n, a), (_, b) in zip(fast.named_parameters(), ref.named_parameters()):
        gb = b.grad.cpu().numpy()
        np.testing.assert_allclose(a.grad.cpu().numpy(), gb, rtol=2e-3, atol=2e-3 * max(np.abs(gb).max(), 1e-8), err_msg=n)


@pytest.mark.gpu
def test_trainer_with_graphed_encoder_matches_eager():
    """Trainer(graph_encoder=True): the image encoder replayed from hipGraphs gives the same losses and the
    same flat gradient as eager execution, for the teacher-forced and the episode step, and capture leaves the
    batch-norm running statistics untouched."""
    import copy
    import t2onet_amd
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Trainer
    dev = torch.device('cuda:0')
    opt = t2onet_amd.default_options()
    opt.input_dropout_p = opt.dropout_p = 0.0      # the request encoder's dropout (library RNG state) is not what is compared here
    torch.manual_seed(11)
    base = Actor(opt).to(dev).train()
    B, H, W = 4, 64, 64
    img_x = synth.images(B, H, W, 81).to(dev)
    img_y = torch.stack([synth.images(B, H, W, 82 + k) for k in range(6)], 1).to(dev)
    x = synth.requests(B, 17, 83).to(dev)
    y = synth.op_targets(B, 84).to(dev)
    gt = synth.uniform((B, 5, 24), 85, -1.0, 1.0).to(dev)
    lengths = (x != 0).sum(1).cpu()
    def logical_grads(tr):
        # the flat buffer stores each gradient with its parameter's strides (channels-last weights in NHWC mode):
        # compare in logical (N,C,H,W) order
        return torch.cat([p.grad.contiguous().reshape(-1) for p in tr.grads.params])

    results = []
    for graph, nhwc in ((False, False), (True, False), (False, True), (True, True)):
        row = []
        for kind in ('supervised', 'episode'):      # each step from the SAME initial weights (a fresh copy): what differs
            model = copy.deepcopy(base)             # between the variants is then only rounding, not an Adam update
            if nhwc:                                # channels-last encoder: NHWC convolutions + t2o_bn_relu_nhwc_*
                model.use_channels_last()
            tr = Trainer(model, opt, graph_encoder=graph)
            stats0 = [b.clone() for b in model.vis_encoder.buffers()]
            if graph:
                tr._maybe_graph(img_x)
                for b0, b1 in zip(stats0, model.vis_encoder.buffers()):
                    assert torch.equal(b0, b1)
                assert '_graphed_encoders' in model.__dict__ and tr.graph_encoder
                assert float(tr.grads.flat.abs().max()) == 0.0          # capture left nothing in the gradient buffer
            if kind == 'supervised':
                torch.manual_seed(12)
                sup = tr.supervised_step(x, y, img_x, img_y, gt, lengths=lengths)
                row += [float(sup[0]), float(sup[1]), logical_grads(tr)]
            else:
                torch.manual_seed(13)
                epi = tr.episode_step(x, img_x, img_y[:, -1], lengths=lengths)
                row += [float(epi), logical_grads(tr)]
            if nhwc:                                # gradients live in the flat buffer with the parameter's own strides
                w = model.vis_encoder.layer1[0].conv1.weight
                assert w.grad.stride() == w.stride() and w.is_contiguous(memory_format=torch.channels_last)
        results.append(row)
    o0, p0, gs0, e0, ge0 = results[0]
    names = [n for n, p in base.named_parameters() if p.requires_grad]
    sizes = [p.numel() for p in base.parameters() if p.requires_grad]

    def worst(a, b):
        out, off = [], 0
        for n, k in zip(names, sizes):
            d = float((a[off:off + k] - b[off:off + k]).abs().max())
            out.append((d, n))
            off += k
        return sorted(out, reverse=True)[:3]

    for (o1, p1, gs1, e1, ge1), variant in zip(results[1:], ('graph', 'nhwc', 'graph+nhwc')):
        print(variant, 'supervised worst', worst(gs1, gs0), 'episode worst', worst(ge1, ge0))
        assert abs(o0 - o1) < 1e-5 and abs(p0 - p1) < 1e-4 * max(1.0, abs(p0))
        # element-wise, in units of the largest gradient.  MIOpen picks different convolution kernels for NCHW and NHWC
        # (and split-K weight gradients add atomically): stem-weight gradients, sums of 16 k cancelling terms, move
        # by a few 1e-4 of the maximum between two eager runs on different boxes already
        np.testing.assert_allclose(gs1.cpu().numpy(), gs0.cpu().numpy(), rtol=0, atol=5e-4 * float(gs0.abs().max()))
        # the episode samples operators: same seed, same draws (encoder graphs consume no random numbers)
        assert abs(e0 - e1) < 1e-5
        # (3e-3: with encoder graphs the feature head runs as fc + BatchNorm1d modules, eagerly as the fused kernel -- two fp32
        # roundings of the same values, amplified through five chained encoder passes: measured 1.5e-3 on 5 of 22 M entries.
        # This end-to-end bound is not the feature head's only guard: tests/test_gpu_decoder.py::test_feature_head_matches_fp64
        # holds the fused kernel to 3e-6 (values) / 3e-5 (gradients) of the fc + bn1 + relu MODULES in fp64 -- ADVICE r4)
        np.testing.assert_allclose(ge1.cpu().numpy(), ge0.cpu().numpy(), rtol=0, atol=3e-3 * float(ge0.abs().max()))


@pytest.mark.gpu
def test_flat_adam_matches_torch_adam():
    """FlatAdam (one t2o_adam_step launch over the flat parameter / gradient / moment buffers) against
    torch.optim.Adam with the reference's settings (train_seq2seqL1.py:169), several steps, a channels-last weight and
    an odd total size (tail elements)."""
    import copy
    from t2onet_amd.train import FlatGradients, FlatAdam
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    net = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 3, 3, padding=1)).to(dev)
    net[0].to(memory_format=torch.channels_last)
    extra = torch.nn.Parameter(torch.randn(7, device=dev))              # 4*8*9 + 8 + 8*3*9 + 3 + 7 = 522 = 4 * 130 + 2
    params = list(net.parameters()) + [extra]
    ref_params = [p.detach().clone().requires_grad_(True) for p in params]
    ref_opt = torch.optim.Adam(ref_params, lr=1e-3)
    grads = FlatGradients(params)
    opt = FlatAdam(grads, lr=1e-3)
    assert params[0].is_contiguous(memory_format=torch.channels_last) and params[0].data_ptr() == opt.flat_param.data_ptr()
    for it in range(5):
        g = [synth.uniform(tuple(p.shape), 900 + 10 * it + k, -1.0, 1.0).to(dev) * (10.0 ** (k - 2)) for k, p in enumerate(params)]
        grads.zero()
        for p, q, gi in zip(params, ref_params, g):
            p.grad.copy_(gi)
            q.grad = gi.clone()
        opt.step()
        ref_opt.step()
        for p, q in zip(params, ref_params):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    x = torch.randn(2, 4, 5, 5, device=dev)
    assert torch.isfinite(net(x)).all()                                  # the module still runs on its re-homed parameters
    # optimiser checkpoints: the state names its segment layout; a state of the earlier PACKED layout (no offsets) converts;
    # a state of another model is refused with a message, not a bare size mismatch
    sd = opt.state_dict()
    assert sd['layout'] == 2 and sd['offsets'] == grads.offsets
    packed = {'step': sd['step'], 'lr': sd['lr'], 'betas': sd['betas'], 'eps': sd['eps'],
              'exp_avg': torch.cat([sd['exp_avg'][o:o + p.numel()] for o, p in zip(grads.offsets, params)]),
              'exp_avg_sq': torch.cat([sd['exp_avg_sq'][o:o + p.numel()] for o, p in zip(grads.offsets, params)])}
    keep = (sd['exp_avg'].clone(), sd['exp_avg_sq'].clone())
    opt.exp_avg.fill_(7.0)
    opt.exp_avg_sq.fill_(7.0)
    opt.load_state_dict(packed)
    for o, p in zip(grads.offsets, params):
        assert torch.equal(opt.exp_avg[o:o + p.numel()], keep[0][o:o + p.numel()])
        assert torch.equal(opt.exp_avg_sq[o:o + p.numel()], keep[1][o:o + p.numel()])
    opt.load_state_dict({**sd, 'exp_avg': keep[0], 'exp_avg_sq': keep[1]})
    assert torch.equal(opt.exp_avg, keep[0])
    with pytest.raises(ValueError, match='other parameters'):
        opt.load_state_dict({**sd, 'numels': sd['numels'][:-1], 'offsets': sd['offsets'][:-1]})


@pytest.mark.gpu
def test_graphed_training_stays_on_the_eager_trajectory():
    """Several episode train steps at the encoder's real stage sizes (256x256 images: every own convolution kernel
    is in the captured graphs), hipGraph replays against eager execution from the same weights and seeds.  One
    replay matching is not enough: a memset node that ran out of order inside the replayed graphs (fixed, see
    t2o_conv.hip k_conv_zero) left the first step intact and produced garbage gradients from the second on."""
    import copy
    import t2onet_amd
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Trainer
    dev = torch.device('cuda:0')
    opt = t2onet_amd.default_options()
    opt.input_dropout_p = opt.dropout_p = 0.0
    torch.manual_seed(21)
    base = Actor(opt).to(dev).train()
    base.use_channels_last()
    B, H, W = 8, 256, 256
    img = synth.images(B, H, W, 91).to(dev)
    tgt = synth.images(B, H, W, 92).to(dev)
    x = synth.requests(B, 17, 93).to(dev)
    lengths = (x != 0).sum(1).cpu()
    runs = {}
    for graph in (False, True):
        model = copy.deepcopy(base)
        tr = Trainer(model, opt, graph_encoder=graph)
        torch.manual_seed(22)
        losses = []
        for step in range(5):
            losses.append(float(tr.episode_step(x, img, tgt, lengths=lengths)))
            bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
            assert not bad, (graph, step, losses, bad[:6], len(bad))
        assert all(bool(torch.isfinite(p).all()) for p in model.parameters()), (graph, losses)
        if graph:
            assert '_graphed_encoders' in model.__dict__ and tr.graph_encoder
        runs[graph] = losses
    # the library's remaining atomic kernels make two EAGER runs differ by ~4e-4 at step 2 and (sampled operators flip)
    # by ~1e-2 from step 3 on: two steps are held tightly, the rest to the trajectory's own spread
    np.testing.assert_allclose(runs[True][:2], runs[False][:2], rtol=0, atol=2e-3)
    np.testing.assert_allclose(runs[True], runs[False], rtol=0, atol=4e-2)
    assert runs[False][-1] < runs[False][0] and runs[True][-1] < runs[True][0]      # (and it trains)


@pytest.mark.gpu
def test_graph_memset_nodes_become_kernel_nodes():
    """t2o_graph_memsets_to_kernels on a captured graph: kernel -> hipMemsetAsync (part of the buffer) -> kernel.  The
    rewritten graph has no memset node left, keeps the order (the fill runs between the two kernels) and the values
    (1-, 2- and 4-byte patterns)."""
    import ctypes
    from t2onet_amd import graphs
    dev = torch.device('cuda:0')
    hip = ctypes.CDLL('libamdhip64.so')
    buf = torch.zeros(4096, dtype=torch.int32, device=dev)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        buf.fill_(7)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        assert hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr() + 4 * 1024), 0xAB, ctypes.c_size_t(4 * 1024), st) == 0
        assert hip.hipMemsetD32Async(ctypes.c_void_p(buf.data_ptr() + 4 * 3072), 5, ctypes.c_size_t(512), st) == 0
        buf.add_(1)
    assert graphs._harden(g) == 2
    assert graphs._harden.__doc__
    for _ in range(3):
        buf.fill_(-1)
        junk = torch.full((1 << 20,), 3, device=dev)
        del junk
        g.replay()
        torch.cuda.synchronize()
        expect = torch.full((4096,), 8, dtype=torch.int32)
        expect[1024:2048] = int(np.array([0xABABABAB], dtype=np.uint32).view(np.int32)[0]) + 1
        expect[3072:3584] = 6
        assert torch.equal(buf.cpu(), expect)


@pytest.mark.gpu
def test_graphed_steps_stay_on_the_eager_trajectory_without_sampling():
    """reinforce_sample=0 (arg-max operators: no random draw anywhere, request-encoder dropout off): five train steps
    through (a) eager execution, (b) the per-call encoder hipGraphs, (c) ONE hipGraph for the whole step behind the
    request encoder (graphs.GraphedEpisodeStep) must give the same losses step by step -- a graph that only perturbed
    a gradient would drift within a step or two at this tolerance."""
    import copy
    import t2onet_amd
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Trainer
    dev = torch.device('cuda:0')
    opt = t2onet_amd.default_options()
    opt.input_dropout_p = opt.dropout_p = 0.0
    torch.manual_seed(31)
    base = Actor(opt).to(dev).train()
    base.use_channels_last()
    B, H, W = 8, 256, 256
    img = synth.images(B, H, W, 95).to(dev)
    tgt = synth.images(B, H, W, 96).to(dev)
    x = synth.requests(B, 17, 97).to(dev)
    lengths = (x != 0).sum(1).cpu()
    runs = {}
    for mode in ('eager', 'encoder_graphs', 'step_graph'):
        model = copy.deepcopy(base)
        tr = Trainer(model, opt, graph_encoder=mode != 'eager', graph_step=mode == 'step_graph')
        losses = []
        for step in range(5):
            losses.append(float(tr.episode_step(x, img, tgt, reinforce_sample=0, lengths=lengths)))
        assert all(bool(torch.isfinite(p).all()) for p in model.parameters()), (mode, losses)
        if mode == 'step_graph':
            assert tr.graph_step and len(tr._step_graphs) == 1, 'the whole-step graph was not used'
            assert '_graphed_encoders' not in model.__dict__
        if mode == 'encoder_graphs':
            assert '_graphed_encoders' in model.__dict__
        runs[mode] = (losses, [p.detach().clone() for p in model.parameters()])
    for mode in ('encoder_graphs', 'step_graph'):
        np.testing.assert_allclose(runs[mode][0], runs['eager'][0], rtol=0, atol=2e-3, err_msg=mode)
    assert runs['eager'][0][-1] < runs['eager'][0][0]


@pytest.mark.gpu
def test_step_graph_draws_fresh_samples_and_trains():
    """The whole-step hipGraph with sampled operators: every replay draws new random numbers (the losses of repeated
    steps from FROZEN weights differ), parameters stay finite, and with a learning rate the loss goes down."""
    import copy
    import t2onet_amd
    from t2onet_amd.actor import Actor
    from t2onet_amd.train import Trainer
    dev = torch.device('cuda:0')
    opt = t2onet_amd.default_options()
    torch.manual_seed(41)
    base = Actor(opt).to(dev).train()
    base.use_channels_last()
    B, H, W = 8, 256, 256
    img = synth.images(B, H, W, 101).to(dev)
    tgt = synth.images(B, H, W, 102).to(dev)
    x = synth.requests(B, 17, 103).to(dev)
    lengths = (x != 0).sum(1).cpu()
    frozen = Trainer(copy.deepcopy(base), opt, lr=0.0, graph_step=True)
    seen = {round(float(frozen.episode_step(x, img, tgt, lengths=lengths)), 7) for _ in range(6)}
    assert len(frozen._step_graphs) == 1 and len(seen) > 1, seen
    # a second request length: its own graph
    x2 = x.clone()
    x2[:, 9:] = 0
    x2[:, 8] = 2
    frozen.episode_step(x2, img, tgt, lengths=(x2 != 0).sum(1).cpu())
    assert len(frozen._step_graphs) == 2
    tr = Trainer(copy.deepcopy(base), opt, graph_step=True)
    losses = [float(tr.episode_step(x, img, tgt, lengths=lengths)) for _ in range(12)]
    assert all(bool(torch.isfinite(p).all()) for p in tr.model.parameters())
    assert min(losses[-4:]) < losses[0], losses


def test_end_select_l1_equals_stack_gather_l1():
    """functional.end_select_l1 (one kernel on the list of step images) == torch.stack + select_end_images + l1_loss
    (train_seq2seqL1.py:78-85): loss and every step image's gradient bit for bit, END at any step or nowhere."""
    import t2onet_amd.functional as T
    from t2onet_amd.train import select_end_images, first_end_step, end_l1_loss
    dev = torch.device('cuda:0')
    B, Tn, H, W = 6, 5, 20, 12
    imgs = [synth.uniform((B, 3, H, W), 2100 + t, 0.0, 1.0).to(dev).requires_grad_(True) for t in range(Tn)]
    tgt = synth.uniform((B, 3, H, W), 2110, 0.0, 1.0).to(dev)
    ops = torch.tensor([[5, 2, 4, 4, 4], [2, 3, 3, 3, 3], [3, 4, 5, 6, 8], [3, 4, 5, 6, 2], [3, 2, 2, 2, 2], [4, 4, 2, 5, 2]], device=dev)
    assert first_end_step(ops, 2).tolist() == [1, 0, 4, 4, 1, 2]
    loss0 = T.l1_loss(select_end_images(torch.stack(imgs, 1), ops, 2), tgt)
    g0 = torch.autograd.grad(loss0 * 3.0, imgs)
    loss1 = end_l1_loss(imgs, ops, 2, tgt)
    g1 = torch.autograd.grad(loss1 * 3.0, imgs)
    assert torch.equal(loss0, loss1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
