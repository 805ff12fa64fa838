"""The request encoder's LSTM step kernels (t2o_rnn.hip, functional.lstm_layer) against torch.nn.LSTM on packed
sequences in fp64 (models/lang_encoder.py:70-113: pack_padded_sequence -> LSTM -> pad_packed_sequence), forward and
every gradient; and the whole RNNEncoder against the library path."""
import numpy as np
import pytest
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _close(got, ref, tol):
    ref = ref.detach().float().cpu()
    scale = float(ref.abs().max()) or 1.0
    np.testing.assert_allclose(got.detach().float().cpu().numpy(), ref.numpy(), rtol=tol, atol=tol * scale)


# (B, L, E, H, D, lengths): ragged batch (not a multiple of the 8-sample tile), full / single-token rows, one and two
# directions, the encoder's own shape
CASES = [(5, 7, 30, 64, 2, [7, 3, 5, 1, 7]), (3, 4, 16, 64, 1, [4, 2, 4]), (9, 1, 12, 128, 2, [1] * 9),
         (64, 17, 300, 256, 2, None), (16, 6, 512, 256, 2, [6, 6, 5, 4, 6, 1, 2, 3, 6, 6, 6, 2, 1, 5, 4, 3])]


@pytest.mark.parametrize('case', CASES)
def test_lstm_layer_matches_packed_nn_lstm_fp64(case):
    import t2onet_amd.functional as T
    B, L, E, H, D, lengths = case
    if lengths is None:
        lengths = [int(v) for v in (synth.uniform((B,), 5, 0.0, 1.0) * L).long().clamp(1, L)]
        lengths[0] = L
    torch.manual_seed(3)
    ref = nn.LSTM(E, H, 1, batch_first=True, bidirectional=D == 2).double()
    x = synth.uniform((B, L, E), 11, -1.0, 1.0)
    for b in range(B):
        x[b, lengths[b]:] = 0
    dout = synth.uniform((B, L, D * H), 12, -1.0, 1.0)
    dh = synth.uniform((D, B, H), 13, -1.0, 1.0)
    dc = synth.uniform((D, B, H), 14, -1.0, 1.0)
    x64 = x.double().requires_grad_(True)
    po, (h, c) = ref(pack_padded_sequence(x64, torch.tensor(lengths), batch_first=True, enforce_sorted=False))
    out = pad_packed_sequence(po, batch_first=True, total_length=L)[0]
    ((out * dout.double()).sum() + (h * dh.double()).sum() + (c * dc.double()).sum()).backward()
    xg = x.to(DEV).requires_grad_(True)
    ws = []
    for d in range(D):
        sfx = '_l0' + ('_reverse' if d else '')
        ws.append(tuple(getattr(ref, n + sfx).detach().float().to(DEV).requires_grad_(True)
                        for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')))
    o, hn, cn = T.lstm_layer(xg, torch.tensor(lengths), ws)
    _close(o, out.detach(), 2e-6)
    _close(hn, h.detach(), 2e-6)
    _close(cn, c.detach(), 2e-6)
    assert float(o[0, lengths[0]:].abs().sum()) == 0.0 if lengths[0] < L else True
    ((o * dout.to(DEV)).sum() + (hn * dh.to(DEV)).sum() + (cn * dc.to(DEV)).sum()).backward()
    _close(xg.grad, x64.grad, 2e-5)
    for d in range(D):
        sfx = '_l0' + ('_reverse' if d else '')
        for k, n in enumerate(('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')):
            _close(ws[d][k].grad, getattr(ref, n + sfx).grad, 2e-5)


def test_request_encoder_own_kernels_match_the_library_path(monkeypatch):
    """RNNEncoder on the GPU: own LSTM kernels vs the packed library call (dropout off), outputs and all gradients."""
    import t2onet_amd
    import t2onet_amd.lang_encoder as LE
    from t2onet_amd.actor import Actor
    opt = t2onet_amd.default_options(input_dropout_p=0.0, dropout_p=0.0)
    torch.manual_seed(5)
    model = Actor(opt).to(DEV).train()
    x = synth.requests(16, 17, 21).to(DEV)
    lengths = (x != 0).sum(1).cpu()
    res = {}
    for own in (True, False):
        monkeypatch.setattr(LE, '_OWN_LSTM', own)
        model.zero_grad(set_to_none=True)
        out, (h, c), emb = model.lang_encoder(x, lengths)
        (out.square().sum() + h.sum() + 2 * c.sum()).backward()
        res[own] = (out.detach().clone(), h.detach().clone(), c.detach().clone(),
                    {n: p.grad.detach().clone() for n, p in model.lang_encoder.named_parameters() if p.grad is not None})
    for a, b in zip(res[True][:3], res[False][:3]):
        _close(a, b, 1e-5)
    assert res[True][3].keys() == res[False][3].keys()
    for n in res[True][3]:
        _close(res[True][3][n], res[False][3][n], 1e-4)
    # without host lengths (one device->host copy for the truncation length) and with `longest` given (none at all)
    monkeypatch.setattr(LE, '_OWN_LSTM', True)
    with torch.no_grad():
        o2 = model.lang_encoder(x)[0]
        o3 = model.lang_encoder(x, (x != 0).sum(1), int(lengths.max()))[0]
    assert torch.equal(o2, res[True][0]) and torch.equal(o3, res[True][0])
