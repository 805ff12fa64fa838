"""The fused decoder step and feature head (t2o_decoder.hip, decoder_step.py) against the SAME modules run by PyTorch
in fp64 (models/action_decoder.py:38-64, models/attention.py:17-44, models/actor.py:50): every output, every data
gradient and every parameter gradient; the Trainer's deferred weight gradients (one product per weight over the steps
of a train step) against per-step autograd; evaluation mode; the argument validation of the C entry points."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _close(got, ref, tol, what=''):
    ref = ref.detach().double().cpu()
    scale = float(ref.abs().max()) or 1.0
    np.testing.assert_allclose(got.detach().double().cpu().numpy(), ref.numpy(), rtol=tol, atol=tol * scale, err_msg=what)


def _decoder(seed, dev=DEV):
    from t2onet_amd.action_decoder import Decoder
    torch.manual_seed(seed)
    dec = Decoder(11, 5, 300, 256, 2, bidirectional=True, use_attention=True)
    with torch.no_grad():
        for p in dec.parameters():
            p.mul_(2.0)                                          # (livelier activations than the default initialisation)
    return dec.to(dev)


def _reference_step(dec64, prev_op, hidden, enc, feat):
    """The reference's forward_step, literally (action_decoder.py:38-64 with attention.py:17-44), on an fp64 copy."""
    B = prev_op.shape[0]
    token = dec64.embedding(prev_op)
    vis = F.relu(dec64.vis_linear(feat)).unsqueeze(1)
    out, hidden = dec64.rnn(torch.cat((token, vis), 2), hidden)
    attn = F.softmax(torch.bmm(out, enc.transpose(1, 2)).view(-1, enc.shape[1]), dim=1).view(B, -1, enc.shape[1])
    mix = torch.bmm(attn, enc)
    comb = torch.cat((mix, out), dim=2)
    ctx = torch.tanh(dec64.attention.linear_out(comb.view(-1, 2 * dec64.hidden_size))).view(B, -1, dec64.hidden_size)
    logp = F.log_softmax(dec64.out_linear(ctx.contiguous().view(-1, dec64.hidden_size)), dim=1).view(B, 1, -1)
    return logp, hidden, attn, ctx.squeeze(1)


def _inputs(B, L, seed):
    D = 512
    feat = synth.uniform((B, D), seed + 1, 0.0, 1.5)
    h = synth.uniform((2, B, D), seed + 2, -1.0, 1.0)
    c = synth.uniform((2, B, D), seed + 3, -1.0, 1.0)
    enc = synth.uniform((B, L, D), seed + 4, -0.3, 0.3)
    op = (synth.uniform((B, 1), seed + 5, 0.0, 1.0) * 11).long().clamp(0, 10)
    return feat, h, c, enc, op


@pytest.mark.parametrize('B,L', [(64, 17), (8, 5), (4, 12), (37, 9), (1, 3), (70, 4)])
def test_fused_step_matches_fp64_modules(B, L):
    dec = _decoder(5)
    dec64 = copy.deepcopy(dec).double().cpu()
    feat, h, c, enc, op = _inputs(B, L, 100 + B)
    g = [synth.uniform(s, 200 + i, -1.0, 1.0) for i, s in enumerate([(B, 1, 11), (2, B, 512), (2, B, 512), (B, 512)])]
    # fp64 reference
    r_in = [t.double().requires_grad_(True) for t in (feat, h, c, enc)]
    logp, (hn, cn), attn, ctx = _reference_step(dec64, op, (r_in[1], r_in[2]), r_in[3], r_in[0])
    ((logp * g[0].double()).sum() + (hn * g[1].double()).sum() + (cn * g[2].double()).sum() + (ctx * g[3].double()).sum()).backward()
    # fused step
    t_in = [t.to(DEV).requires_grad_(True) for t in (feat, h, c, enc)]
    logp2, (hn2, cn2), attn2, ctx2 = dec.forward_step(op.to(DEV), (t_in[1], t_in[2]), t_in[3], t_in[0])
    assert logp2.shape == (B, 1, 11) and hn2.shape == (2, B, 512) and attn2.shape == (B, 1, L) and ctx2.shape == (B, 512)
    for got, ref, name in ((logp2, logp, 'logp'), (hn2, hn, 'h'), (cn2, cn, 'c'), (attn2, attn, 'attn'), (ctx2, ctx, 'ctx')):
        _close(got, ref, 3e-6, name)
    gd = [t.to(DEV) for t in g]
    ((logp2 * gd[0]).sum() + (hn2 * gd[1]).sum() + (cn2 * gd[2]).sum() + (ctx2 * gd[3]).sum()).backward()
    for a, b, name in zip(t_in, r_in, ('feat', 'h', 'c', 'enc')):
        _close(a.grad, b.grad, 2e-5, 'd ' + name)
    for (name, p), (_, q) in zip(dec.named_parameters(), dec64.named_parameters()):
        assert p.grad is not None, name
        _close(p.grad, q.grad, 2e-5, 'd ' + name)


def test_fused_step_equals_the_per_layer_path(monkeypatch):
    """Same module, fused step on and off (library GEMMs + framework gate kernels): outputs and gradients agree in fp32."""
    import t2onet_amd.action_decoder as AD
    dec = _decoder(7)
    B, L = 16, 9
    feat, h, c, enc, op = _inputs(B, L, 300)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(AD, '_FUSED_STEP', fused)
        dec.zero_grad(set_to_none=True)
        t_in = [t.to(DEV).requires_grad_(True) for t in (feat, h, c, enc)]
        logp, (hn, cn), _, ctx = dec.forward_step(op.to(DEV), (t_in[1], t_in[2]), t_in[3], t_in[0])
        (logp.sum() * 0.3 + hn.square().sum() + cn.sum() + ctx.square().sum()).backward()
        outs.append([logp, hn, cn, ctx] + [t.grad for t in t_in] + [p.grad.clone() for p in dec.parameters()])
    for a, b in zip(*outs):
        _close(a, b, 2e-5)


def test_context_only_and_state_only_gradients():
    """An episode's last step receives a gradient through its context only; a step whose operator was END may receive
    one through the states only."""
    dec = _decoder(9)
    dec64 = copy.deepcopy(dec).double().cpu()
    B, L = 8, 6
    feat, h, c, enc, op = _inputs(B, L, 400)
    for which in ('ctx', 'state'):
        dec.zero_grad(set_to_none=True)
        dec64.zero_grad(set_to_none=True)
        r_in = [t.double().requires_grad_(True) for t in (feat, h, c, enc)]
        _, (hn, cn), _, ctx = _reference_step(dec64, op, (r_in[1], r_in[2]), r_in[3], r_in[0])
        (ctx.square().sum() if which == 'ctx' else (hn[0] * cn[1]).sum()).backward()
        t_in = [t.to(DEV).requires_grad_(True) for t in (feat, h, c, enc)]
        _, (hn2, cn2), _, ctx2 = dec.forward_step(op.to(DEV), (t_in[1], t_in[2]), t_in[3], t_in[0])
        (ctx2.square().sum() if which == 'ctx' else (hn2[0] * cn2[1]).sum()).backward()
        for a, b in zip(t_in, r_in):
            if b.grad is None:
                assert a.grad is None or float(a.grad.abs().max()) == 0.0
            else:
                _close(a.grad, b.grad, 2e-5, which)
        for (name, p), (_, q) in zip(dec.named_parameters(), dec64.named_parameters()):
            if q.grad is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            else:
                _close(p.grad, q.grad, 2e-5, which + ' d ' + name)


@pytest.mark.parametrize('B,training', [(64, True), (8, True), (2, True), (5, False), (1, False)])
def test_feature_head_matches_fp64(B, training):
    from t2onet_amd import decoder_step as DS
    torch.manual_seed(2)
    fc, bn = nn.Linear(512, 512).to(DEV), nn.BatchNorm1d(512).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.1, 0.1)
        bn.running_var.uniform_(0.5, 1.5)
    fc64, bn64 = copy.deepcopy(fc).double().cpu(), copy.deepcopy(bn).double().cpu()
    for m in (bn, bn64):
        m.train(training)
    pooled = synth.uniform((B, 512), 31, 0.0, 1.0)
    gout = synth.uniform((B, 512), 32, -1.0, 1.0)
    x64 = pooled.double().requires_grad_(True)
    ref = F.relu(bn64(fc64(x64)))
    (ref * gout.double()).sum().backward()
    x = pooled.to(DEV).requires_grad_(True)
    assert DS.feature_supported(x, fc, bn)
    got = DS.image_feature(x, fc, bn)
    # (two rows in training mode: the normalisation divides by half the difference of two nearly equal numbers wherever a
    # column's values are close -- 1 / std up to 1e3 here -- and fp32 rounding of fc's output is amplified accordingly)
    tol = 2e-3 if (training and B == 2) else 3e-5
    _close(got, ref, 3e-6 if tol < 1e-3 else 1e-4)
    (got * gout.to(DEV)).sum().backward()
    _close(x.grad, x64.grad, tol)
    for p, q in ((fc.weight, fc64.weight), (bn.weight, bn64.weight), (bn.bias, bn64.bias)):
        _close(p.grad, q.grad, tol)
    if training:            # batch statistics remove any shift of the Linear's output: this gradient is zero up to rounding
        assert float(fc.bias.grad.abs().max()) < (1e-4 if B > 2 else 5e-2) * float(bn.bias.grad.abs().max()) and float(fc64.bias.grad.abs().max()) < 1e-9
    else:
        _close(fc.bias.grad, fc64.bias.grad, 3e-5)
    # running statistics exactly as torch updates them
    _close(bn.running_mean, bn64.running_mean, 1e-6)
    _close(bn.running_var, bn64.running_var, 1e-6)
    assert int(bn.num_batches_tracked) == int(bn64.num_batches_tracked)


def test_deferred_weight_gradients_equal_per_step_autograd():
    """A persistent tape (what the Trainer installs): three chained steps + feature heads, weight gradients formed once by
    tape.flush(), against the same chain with private tapes (gradients returned to autograd step by step)."""
    from t2onet_amd import decoder_step as DS
    dec = _decoder(11)
    torch.manual_seed(4)
    fc, bn = nn.Linear(512, 512).to(DEV), nn.BatchNorm1d(512).to(DEV)
    B, L, S = 16, 7, 3
    _, h, c, enc, _ = _inputs(B, L, 500)
    pooled = [synth.uniform((B, 512), 510 + s, 0.0, 1.0).to(DEV) for s in range(S)]
    ops = [(synth.uniform((B, 1), 520 + s, 0.0, 1.0) * 11).long().clamp(0, 10).to(DEV) for s in range(S)]

    class Holder:                                                  # what tape.flush() reads: decoder, vis_encoder.fc, bn1
        pass
    model = Holder()
    model.decoder, model.bn1, model.vis_encoder = dec, bn, Holder()
    model.vis_encoder.fc = fc
    params = list(dec.parameters()) + list(fc.parameters()) + list(bn.parameters())

    def run(tape):
        for p in params:
            p.grad = None
        bn.reset_running_stats()
        hid = (list(h.to(DEV).unbind(0)), list(c.to(DEV).unbind(0)))
        e = enc.to(DEV).requires_grad_(True)
        loss = 0.0
        for s in range(S):
            feat = DS.image_feature(pooled[s], fc, bn, tape)
            logp, hid, _, ctx = DS.decoder_step(dec, ops[s], hid, e, feat, tape)
            loss = loss + ctx.square().sum() + (logp.sum() * 0.1 if s != 1 else 0.0)     # (step 1: no gradient through its scores)
        loss.backward()
        if tape is not None:
            assert all(p.grad is None for p in params)              # nothing was handed to autograd
            tape.flush(model)
        return [p.grad.clone() for p in params] + [e.grad.clone()]

    ref = run(None)
    tape = DS.DecoderTape(B, 512, 300, 11, 512, S + 1, torch.device(DEV), persistent=True)
    got = run(tape)
    names = [n for n, _ in dec.named_parameters()] + ['fc.w', 'fc.b', 'bn.w', 'bn.b', 'enc']
    for a, b, name in zip(got, ref, names):
        if name != 'fc.b':                                          # (zero in exact arithmetic -- bn removes any shift: rounding noise)
            _close(a, b, 1e-5, name)
    got2 = run(tape)                                                # the tape rewinds: a second train step gives the same
    for a, b, name in zip(got2, ref, names):
        if name != 'fc.b':
            _close(a, b, 1e-5, name)


def test_zero_grad_between_forward_and_backward_gives_autograd_gradients():
    """The reference's loop (train_seq2seqL1.py:63,86): forward, optimizer.zero_grad() (set_to_none under torch >= 2),
    backward, optimizer.step() with a STOCK optimiser -- every parameter must end up with an ordinary .grad."""
    import t2onet_amd
    from t2onet_amd.actor import Actor
    opt = t2onet_amd.default_options()
    model = Actor(opt).to(DEV)
    model.use_channels_last()
    model.train()
    optim = torch.optim.Adam(model.parameters(), lr=1e-3)
    B = 4
    x = torch.zeros(B, 17, dtype=torch.long)
    x[:, 0], x[:, 1:4], x[:, 4] = 1, 7, 2
    img = synth.images(B, 64, 64, 3).to(DEV)
    for _ in range(2):                                              # (the second pass meets the .grad tensors of the first)
        _, imgs, _, _ = model.episode_forward(x.to(DEV), img, None, reinforce_sample=0)
        loss = (imgs[:, -1] - 0.5).abs().mean()
        optim.zero_grad()
        loss.backward()
        have = [n for n, p in model.named_parameters() if p.grad is not None]
        assert any(n.startswith('vis_encoder.conv1') for n in have) and any(n.startswith('decoder.rnn') for n in have)
        assert any(n.startswith('vis_encoder.fc') for n in have) and any(n.startswith('bn1') for n in have)
        optim.step()
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_argument_validation():
    import ctypes
    from t2onet_amd import _lib
    lib = _lib.load()
    a = _lib.DecoderStepArgs()
    assert lib.t2o_decoder_step_fwd(None, None) == 1
    a.B, a.L, a.D, a.E, a.V = 4, 5, 500, 300, 11                    # D % 64 != 0
    assert lib.t2o_decoder_step_fwd(ctypes.byref(a), None) == 1 and b'D % 64' in lib.t2o_last_error()
    a.D = 512
    assert lib.t2o_decoder_step_fwd(ctypes.byref(a), None) == 1 and b'null' in lib.t2o_last_error()
    f = _lib.ImageFeatureArgs()
    f.B, f.K, f.D = 65, 512, 512
    buf = torch.zeros(16, device=DEV)
    for n in ('fc_w', 'pooled', 'fc_out', 'stats', 'feat'):
        setattr(f, n, buf.data_ptr())
    assert lib.t2o_image_feature_fwd(ctypes.byref(f), None) == 1 and b'B <= 64' in lib.t2o_last_error()
    f.B, f.training = 1, 1
    assert lib.t2o_image_feature_fwd(ctypes.byref(f), None) == 1 and b'more than one row' in lib.t2o_last_error()
