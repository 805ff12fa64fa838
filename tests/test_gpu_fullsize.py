"""BASELINE.json configs[2] at FULL size inside the GPU suite: bs = 64, 256 x 256, one arg-max episode train step
(experiments/t2onet/train_seq2seqL1.py:74-88; request-encoder dropout 0 so that two runs see the same network) through
the Trainer -- the path bench.py times -- compared across the implementation choices the bench's default makes:

  (a) Winograd F(2x2,3x3) on the 256- / 512-channel stages vs the direct kernels everywhere;
  (b) the image encoder's trunk on one bs = 64 batch against the fp64 oracle ResNet (oracle/cpu_ref.py resnet18, run in
      fp64 on the device: same arithmetic as on the host, minutes faster).

Operators must be IDENTICAL (arg-max), the loss equal to 1e-5, every gradient tensor close in relative L2 (the measured
distances are printed with -s).  What "close" can mean here was measured, not assumed: two fp32 executions of this step
that differed ONLY in the rounding of the request encoder's then-library GEMMs (rounds 3-5; own kernels since round 6) ended 2.1e-3 apart on single batch-norm bias
gradients, Winograd vs direct (a) 2.6e-3 -- the step chains five encoder passes and five operator applications through
each other's gradients, and 1e-7 perturbations come out amplified by ~1e4 whatever the kernels; the forward values (loss,
operators, encoder output at 6e-6 of fp64) are where the implementations are held tight.  (b) holds the trunk to the
accuracy class of the framework's own fp32 kernels on the same batch, measured in the same test."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
B, S = 64, 256


def _batch():
    img, tgt = synth.images(B, S, S, 901).to(DEV), synth.images(B, S, S, 902).to(DEV)
    x = synth.requests(B, 17, 903)
    return x.to(DEV), (x != 0).sum(1), img, tgt


def _model(seed=10):
    import t2onet_amd
    from t2onet_amd.actor import Actor
    opt = t2onet_amd.default_options(input_dropout_p=0.0, dropout_p=0.0)
    torch.manual_seed(seed)
    model = Actor(opt).to(DEV).train()
    model.use_channels_last()
    return model, opt


def _one_step(tr, batch):
    """(loss, operators (B, T), flat gradient copy) of one arg-max episode step at lr = 0."""
    model = tr.model
    seen = []
    orig = model.episode_forward

    def spy(*a, **k):
        out = orig(*a, **k)
        seen.append(out[2].detach().clone())
        return out
    model.episode_forward = spy
    try:
        x, lengths, img, tgt = batch
        loss = tr.episode_step(x, img, tgt, reinforce_sample=0, lengths=lengths)
    finally:
        del model.episode_forward
    return float(loss), seen[0], tr.grads.flat.clone()


def _compare(tr, a, b, what, loss_tol, l2_tol):
    (la, oa, ga), (lb, ob, gb) = a, b
    assert torch.equal(oa, ob), '%s: the two runs chose different operators' % what
    assert abs(la - lb) <= loss_tol * max(abs(la), 1e-3), (what, la, lb)
    worst = (0.0, None)
    names = {id(p): n for n, p in tr.model.named_parameters()}
    norms = [float(gb[off:off + p.numel()].double().norm() / p.numel() ** 0.5) for p, off in zip(tr.grads.params, tr.grads.offsets)]
    floor = 1e-4 * float(np.median([n for n in norms if n > 0]))     # r.m.s. entry below this: rounding noise of a gradient that is
    for p, off, rms in zip(tr.grads.params, tr.grads.offsets, norms):   # zero in exact arithmetic (a bias in front of a batch norm)
        u, v = ga[off:off + p.numel()].double(), gb[off:off + p.numel()].double()
        den = float(v.norm())
        if rms <= floor:
            assert float(u.norm() / p.numel() ** 0.5) <= 10 * floor, names[id(p)]
            continue
        d = float((u - v).norm()) / den
        if d > worst[0]:
            worst = (d, names[id(p)])
    print('%s: loss %.8f vs %.8f, worst gradient tensor %s: relative L2 %.3e' % (what, la, lb, worst[1], worst[0]))
    assert worst[0] <= l2_tol, (what, worst)


def test_winograd_and_direct_kernels_give_the_same_full_size_step(monkeypatch):
    import t2onet_amd.encoder as E
    from t2onet_amd.train import Trainer
    model, opt = _model()
    tr = Trainer(model, opt, lr=0.0)
    batch = _batch()
    assert E._WINOGRAD
    wino = _one_step(tr, batch)
    monkeypatch.setattr(E, '_WINOGRAD', False)
    tr._trunk.weights_changed()
    direct = _one_step(tr, batch)
    assert not torch.equal(wino[2], direct[2])                      # (two algorithms did run)
    _compare(tr, wino, direct, 'winograd vs direct', 1e-5, 6e-3)
    monkeypatch.setattr(E, '_WINOGRAD', True)
    tr._trunk.weights_changed()
    again = _one_step(tr, batch)
    # a repeat: the loss bit for bit; the gradients to 1e-6 (this library's kernels add in fixed orders, but the step also
    # contains framework scatter-adds -- the two embedding gradients -- whose atomic order is not fixed)
    assert again[0] == wino[0]
    _compare(tr, again, wino, 'repeat', 0.0, 1e-6)


@pytest.mark.parametrize('size', [256, 128])
def test_trunk_at_batch_64_against_the_fp64_oracle(size):
    """relu(bn1(fc(trunk(img)))) -- Actor.image_features, models/actor.py:142-143 -- for one 64 x 3 x 256 x 256 batch (the bench
    size) and one 64 x 3 x 128 x 128 batch (the size the reference trains at, datasets/FiveKdataset.py:25,68: 32 x 32 ... 4 x 4
    maps -- other kernel choices: TrunkPlan.flop_table) against oracle/cpu_ref.image_features in fp64 (training mode: batch
    statistics everywhere), forward, image gradient and the gradients of every encoder parameter."""
    S = size
    model, opt = _model(seed=12)
    img = synth.images(B, S, S, 911)
    gout = synth.uniform((B, 512), 912, -1.0, 1.0)
    sd64 = {k: v.detach().to(DEV).double().contiguous() for k, v in model.state_dict().items()
            if k.startswith(('vis_encoder.', 'bn1.'))}
    named = dict(model.named_parameters())
    leaves = {k: v.requires_grad_(True) for k, v in sd64.items() if k in named}
    sd64.update(leaves)
    x64 = img.to(DEV).double().requires_grad_(True)
    ref = cpu_ref.image_features(sd64, x64, training=True)
    (ref * gout.to(DEV).double()).sum().backward()
    x = img.to(DEV).requires_grad_(True)
    got = model.image_features(x)
    (got * gout.to(DEV)).sum().backward()

    def rel(u, v):
        return float((u.double() - v.double()).norm() / v.double().norm())

    # the yardstick: the SAME batch through the framework's own fp32 kernels (the oracle's functional ResNet on fp32 device
    # tensors = library convolutions and batch norms), against the same fp64 values
    sd32 = {k: v.detach().float().requires_grad_(k in leaves) for k, v in sd64.items()}
    x32 = img.to(DEV).requires_grad_(True)
    lib = cpu_ref.image_features(sd32, x32, training=True)
    (lib * gout.to(DEV)).sum().backward()
    scale = float(ref.abs().max())
    ferr, ferr_lib = float((got.double() - ref).abs().max()) / scale, float((lib.double() - ref).abs().max()) / scale
    gerr, gerr_lib = rel(x.grad, x64.grad), rel(x32.grad, x64.grad)
    worst, worst_lib = (0.0, None), (0.0, None)
    for k, v in leaves.items():
        if k == 'vis_encoder.fc.bias':                              # zero in exact arithmetic (bn1 removes any shift): rounding noise
            assert float(named[k].grad.abs().max()) < 1e-4 * float(named['bn1.bias'].grad.abs().max())
            continue
        d, dl = rel(named[k].grad, v.grad), rel(sd32[k].grad, v.grad)
        if d > worst[0]:
            worst = (d, k)
        if dl > worst_lib[0]:
            worst_lib = (dl, k)
    print('trunk bs=64 %dx%d vs fp64: forward max err / scale %.3e (library fp32 %.3e), image gradient rel L2 %.3e (%.3e), worst parameter '
          'gradient %s %.3e (library: %s %.3e)' % (S, S, ferr, ferr_lib, gerr, gerr_lib, worst[1], worst[0], worst_lib[1], worst_lib[0]))
    assert ferr < 5e-5
    assert gerr < max(2e-3, 3 * gerr_lib)
    assert worst[0] < max(2e-3, 3 * worst_lib[0]), (worst, worst_lib)
