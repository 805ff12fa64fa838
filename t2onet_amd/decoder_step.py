"""One decoding step and the image feature head on the fused small-M kernels of t2o_decoder.hip.

Two autograd nodes over the C ABI:

  * image_feature(pooled, fc, bn1)              relu(bn1(fc(pooled)))                 models/actor.py:50,142-143,215-216
  * decoder_step(decoder, prev_op, state, enc, feat)   Decoder.forward_step           models/action_decoder.py:38-64

Forward and backward are ONE library call each (6 / 9 launches for the step, 1 / 2 for the head).  The backward
kernels produce data gradients only; every layer's (X, dY) pair stays in a `DecoderTape` and the weight gradients are
one product per weight over all recorded steps:

  * a Trainer (train.py) owns a persistent tape with one slot per decoder step of a train step and calls
    `tape.flush()` after the backward pass: 12 products with K = steps x batch instead of 12 per step, added into the
    parameters' persistent .grad buffers.  Opt-in (`Trainer` sets `model._tape`): nothing infers it from `.grad`.
  * without a tape every call records into a private one-slot tape and its backward returns the parameter gradients
    to autograd like any other node (plain `loss.backward()` / `torch.autograd.grad` / stock optimisers).
"""
import ctypes

import torch

from . import _lib
from . import functional as T
from .functional import _need_gpu, _stream

_STEP_KINDS = None


def _step_kinds(D, E, V, Lmax):
    """name -> floats per row, in tape order (forward values first, then the kept pre-activation gradients)."""
    return (('step_in', E + D), ('featc', D), ('hp0', D), ('hp1', D), ('gates0', 4 * D), ('gates1', 4 * D), ('h0n', D), ('c0n', D),
            ('h1n', D), ('c1n', D), ('mix', D), ('ctx', D), ('logp', V), ('attn', Lmax),
            ('d_logits', V), ('d_lin', D), ('d_gates1', 4 * D), ('d_gates0', 4 * D), ('d_step_in', E + D), ('d_vis', D))


_SCRATCH = ('d_ctx', 'd_mix', 'd_qa', 'd_q', 'd_x1')
_FEAT_KINDS = lambda K, D: (('pooledc', K), ('fc_out', D), ('feat', D), ('d_fc', D))       # noqa: E731


def _al(n):
    return -(-n // 64) * 64          # floats: every array starts on a 256-byte boundary


class DecoderTape:
    """Storage for `steps` decoder steps (and as many feature heads) of a batch of B rows: arrays (steps, B, width) per
    kind carved out of one allocation, so that a weight gradient over the first n steps is one product on a
    (n * B, width) view.  `persistent` tapes are reused step after step (Trainer): begin() rewinds them."""

    L_MAX = 64

    def __init__(self, B, D, E, V, K, steps, device, persistent=False):
        self.B, self.D, self.E, self.V, self.K, self.steps, self.persistent = B, D, E, V, K, steps, persistent
        sk, fk = _step_kinds(D, E, V, self.L_MAX), _FEAT_KINDS(K, D)
        total, layout = 0, []
        for name, w in sk + fk:
            layout.append((name, total, (steps, B, w)))
            total += _al(steps * B * w)
        for name in _SCRATCH:
            layout.append((name, total, (B, D)))
            total += _al(B * D)
        for name in ('stats', 'd_bn'):
            layout.append((name, total, (steps, 2, D)))
            total += _al(steps * 2 * D)
        self.flat = torch.empty(total, dtype=torch.float32, device=device)
        self.a = {name: self.flat[off:off + shape[0] * shape[1] * shape[2]].view(shape) if len(shape) == 3
                  else self.flat[off:off + shape[0] * shape[1]].view(shape) for name, off, shape in layout}
        self.prev_ops = torch.zeros((steps, B), dtype=torch.long, device=device)
        self.begin()

    def begin(self):
        """Start of a train step: all slots free."""
        self.n_steps = self.n_feats = 0
        self.step_done, self.feat_done, self.step_logp = set(), set(), set()

    def take_step(self):
        if self.n_steps >= self.steps:
            return None
        self.n_steps += 1
        return self.n_steps - 1

    def take_feat(self):
        if self.n_feats >= self.steps:
            return None
        self.n_feats += 1
        return self.n_feats - 1

    def rows(self, name, n):
        a = self.a[name]
        return a[:n].reshape(n * a.shape[1], a.shape[2])

    # ------------------------------------------------------------------ weight gradients
    def _clear_missing(self, n, done, names):
        for s in range(n):
            if s not in done:
                for name in names:
                    self.a[name][s].zero_()

    def step_weight_grads(self, dec, sink):
        """sink(param, fn): fn(out, beta) must ADD (beta = 1) or WRITE (beta = 0) the gradient of `param` into `out`."""
        n = self.n_steps
        if n == 0:
            return
        D, E = self.D, self.E
        self._clear_missing(n, self.step_done, ('d_lin', 'd_gates1', 'd_gates0', 'd_step_in', 'd_vis'))
        R = self.rows
        rnn = dec.rnn

        def prod(p, dy, x, cols=None):
            if p is None or not p.requires_grad:
                return
            sink(p, lambda out, beta: T.gemm(dy, x, out=out if cols is None else out[:, cols[0]:cols[1]], a_kmajor=True, b_kmajor=True,
                                             accumulate=bool(beta)), cols)

        def colsum(p, dy):
            if p is None or not p.requires_grad:
                return
            sink(p, lambda out, beta: T.colsum(dy, out=out, accumulate=bool(beta)), None)

        dvis, dg0, dg1, dlin = R('d_vis', n), R('d_gates0', n), R('d_gates1', n), R('d_lin', n)
        prod(dec.vis_linear.weight, dvis, R('featc', n))
        colsum(dec.vis_linear.bias, dvis)
        prod(rnn.weight_ih_l0, dg0, R('step_in', n))
        prod(rnn.weight_hh_l0, dg0, R('hp0', n))
        prod(rnn.weight_ih_l1, dg1, R('h0n', n))
        prod(rnn.weight_hh_l1, dg1, R('hp1', n))
        if rnn.bias:
            for b in (rnn.bias_ih_l0, rnn.bias_hh_l0):
                colsum(b, dg0)
            for b in (rnn.bias_ih_l1, rnn.bias_hh_l1):
                colsum(b, dg1)
        lo = dec.attention.linear_out
        prod(lo.weight, dlin, R('mix', n), (0, D))
        prod(lo.weight, dlin, R('h1n', n), (D, 2 * D))
        colsum(lo.bias, dlin)
        if self.step_logp:
            self._clear_missing(n, self.step_logp, ('d_logits',))
            dl = R('d_logits', n)
            prod(dec.out_linear.weight, dl, R('ctx', n))
            colsum(dec.out_linear.bias, dl)
        emb = dec.embedding.weight
        if emb.requires_grad:
            demb = R('d_step_in', n)[:, :E]
            idx = self.prev_ops[:n].reshape(-1).clamp(0, emb.shape[0] - 1)
            sink(emb, lambda out, beta: (out.zero_() if beta == 0 else out).index_add_(0, idx, demb), None)

    def feat_weight_grads(self, fc, bn, sink):
        n = self.n_feats
        if n == 0:
            return
        self._clear_missing(n, self.feat_done, ('d_fc', 'd_bn'))
        dfc = self.rows('d_fc', n)
        if fc.weight.requires_grad:
            sink(fc.weight, lambda out, beta: T.gemm(dfc, self.rows('pooledc', n), out=out, a_kmajor=True, b_kmajor=True, accumulate=bool(beta)), None)
        if fc.bias is not None and fc.bias.requires_grad:
            sink(fc.bias, lambda out, beta: T.colsum(dfc, out=out, accumulate=bool(beta)), None)
        dbn = self.a['d_bn'][:n]
        if bn.weight is not None and bn.weight.requires_grad:
            sink(bn.weight, lambda out, beta: out.copy_(dbn[:, 0].sum(0)) if beta == 0 else out.add_(dbn[:, 0].sum(0)), None)
        if bn.bias is not None and bn.bias.requires_grad:
            sink(bn.bias, lambda out, beta: out.copy_(dbn[:, 1].sum(0)) if beta == 0 else out.add_(dbn[:, 1].sum(0)), None)

    def flush(self, model):
        """Add the weight gradients of every recorded step into the parameters' .grad (persistent tapes: the Trainer
        calls this once after the backward pass) and rewind."""
        def sink(p, fn, cols):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            fn(p.grad, 1)
        with torch.no_grad():
            self.step_weight_grads(model.decoder, sink)
            self.feat_weight_grads(model.vis_encoder.fc, model.bn1, sink)
        self.begin()


def _collect(build):
    """Run a *_weight_grads with a sink that writes fresh tensors; returns {id(param): grad}."""
    out = {}

    def sink(p, fn, cols):
        g = out.get(id(p))
        if g is None:
            g = out[id(p)] = torch.empty_like(p)
            fn(g, 0)
        else:
            fn(g, 0 if cols is not None else 1)       # a further column block is written, anything else added
    with torch.no_grad():
        build(sink)
    return out


def _p(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------------------- feature head
def feature_supported(pooled, fc, bn):
    return (pooled.is_cuda and pooled.dtype == torch.float32 and pooled.dim() == 2 and 1 <= pooled.shape[0] <= 64
            and pooled.shape[1] % 4 == 0 and fc.in_features == pooled.shape[1] and fc.out_features % 4 == 0
            and isinstance(bn, torch.nn.BatchNorm1d) and bn.num_features == fc.out_features
            and (bn.momentum is not None or not bn.training)
            and ((bn.training and pooled.shape[0] > 1) or (not bn.training and bn.running_mean is not None)))


class _ImageFeatureFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pooled, tape, slot, fc, bn, *params):
        B, K = pooled.shape
        D = fc.out_features
        pooled = pooled.contiguous()
        _need_gpu(pooled)
        a = _lib.ImageFeatureArgs()
        a.fc_w, a.fc_b = _p(fc.weight), _p(fc.bias)
        a.bn_w, a.bn_b = _p(bn.weight), _p(bn.bias)
        training = bn.training or bn.running_mean is None
        if bn.track_running_stats and bn.running_mean is not None:
            a.running_mean, a.running_var = _p(bn.running_mean), _p(bn.running_var)
            a.num_batches_tracked = _p(bn.num_batches_tracked) if training else None
        a.pooled = pooled.data_ptr()
        A = tape.a
        a.pooled_copy, a.fc_out, a.stats, a.feat = (A['pooledc'][slot].data_ptr(), A['fc_out'][slot].data_ptr(),
                                                    A['stats'][slot].data_ptr(), A['feat'][slot].data_ptr())
        a.momentum, a.eps = float(bn.momentum if bn.momentum is not None else 0.0), float(bn.eps)
        a.training, a.B, a.K, a.D = int(training), B, K, D
        _lib.check(_lib.load().t2o_image_feature_fwd(ctypes.byref(a), _stream(pooled.device)), 't2o_image_feature_fwd')
        ctx.args, ctx.tape, ctx.slot, ctx.mods, ctx.nparams = a, tape, slot, (fc, bn), len(params)
        ctx.set_materialize_grads(False)
        return A['feat'][slot]

    @staticmethod
    def backward(ctx, g_feat):
        a, tape, slot = ctx.args, ctx.tape, ctx.slot
        fc, bn = ctx.mods
        none = (None,) * (4 + ctx.nparams)
        if g_feat is None:
            return (None,) + none
        g_feat = g_feat.contiguous()
        d_pooled = torch.empty((a.B, a.K), dtype=torch.float32, device=g_feat.device) if ctx.needs_input_grad[0] else None
        a.g_feat, a.d_fc, a.d_bn, a.d_pooled = g_feat.data_ptr(), tape.a['d_fc'][slot].data_ptr(), tape.a['d_bn'][slot].data_ptr(), _p(d_pooled)
        _lib.check(_lib.load().t2o_image_feature_bwd(ctypes.byref(a), _stream(g_feat.device)), 't2o_image_feature_bwd')
        tape.feat_done.add(slot)
        if tape.persistent:
            return (d_pooled,) + none
        got = _collect(lambda sink: tape.feat_weight_grads(fc, bn, sink))
        params = (fc.weight, fc.bias, bn.weight, bn.bias)
        return (d_pooled, None, None, None, None) + tuple(got.get(id(p)) if p is not None else None for p in params)


def image_feature(pooled, fc, bn, tape=None):
    """relu(bn(fc(pooled))) for pooled (B <= 64, K); `tape`: the Trainer's persistent DecoderTape or None."""
    slot = tape.take_feat() if (tape is not None and tape.B == pooled.shape[0] and torch.is_grad_enabled()) else None
    if slot is None:
        tape = DecoderTape(pooled.shape[0], fc.out_features, 4, 1, pooled.shape[1], 1, pooled.device)
        slot = tape.take_feat()
    # (the parameters are inputs in both modes: the node must exist even when `pooled` carries no gradient -- a frozen
    # encoder; with a persistent tape its backward hands autograd None for them and tape.flush() adds the real gradients)
    return _ImageFeatureFn.apply(pooled, tape, slot, fc, bn, fc.weight, fc.bias, bn.weight, bn.bias)


# ---------------------------------------------------------------------------------------------------------- decoder step
def _step_params(dec):
    rnn, lo = dec.rnn, dec.attention.linear_out
    b = (rnn.bias_ih_l0, rnn.bias_hh_l0, rnn.bias_ih_l1, rnn.bias_hh_l1) if rnn.bias else (None,) * 4
    return (dec.embedding.weight, dec.vis_linear.weight, dec.vis_linear.bias, rnn.weight_ih_l0, rnn.weight_hh_l0, b[0], b[1],
            rnn.weight_ih_l1, rnn.weight_hh_l1, b[2], b[3], lo.weight, lo.bias, dec.out_linear.weight, dec.out_linear.bias)


_PARAM_FIELDS = ('emb', 'vis_w', 'vis_b', 'w_ih0', 'w_hh0', 'b_ih0', 'b_hh0', 'w_ih1', 'w_hh1', 'b_ih1', 'b_hh1', 'lo_w', 'lo_b',
                 'out_w', 'out_b')


def step_supported(dec, input_var, hidden, enc, feat):
    """The fused step takes: fp32 GPU tensors, the actor's decoder layout (2-layer unidirectional LSTM, dot-product
    attention without a weight, no active dropout), a (h, c) state."""
    import torch.nn as nn
    rnn = dec.rnn
    if not (feat.is_cuda and feat.dtype == torch.float32 and enc.is_cuda and enc.dtype == torch.float32 and enc.dim() == 3):
        return False
    if not (isinstance(rnn, nn.LSTM) and rnn.num_layers == 2 and not rnn.bidirectional and rnn.proj_size == 0 and rnn.batch_first
            and (rnn.dropout == 0 or not dec.training) and (dec.input_dropout.p == 0 or not dec.training)):
        return False
    if not (dec.use_attention and not dec.attention.use_weight and isinstance(hidden, (tuple, list)) and len(hidden) == 2):
        return False
    D, E, V = dec.hidden_size, dec.word_vec_dim, dec.output_size
    return (input_var.dim() == 2 and input_var.shape[1] == 1 and D % 64 == 0 and D <= 1024 and E % 4 == 0 and 1 <= V <= 16
            and 1 <= enc.shape[1] <= 64 and enc.shape[2] == D and rnn.input_size == E + D and rnn.hidden_size == D
            and feat.shape[1] == D and dec.vis_linear.in_features == D)


class _DecoderStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, h0, c0, h1, c1, enc, prev_op, tape, slot, dec, *params):
        B, L, D = enc.shape
        feat, h0, c0, h1, c1, enc = (t.contiguous() for t in (feat, h0, c0, h1, c1, enc))
        _need_gpu(feat, h0, c0, h1, c1, enc)
        prev_op = prev_op.reshape(-1).contiguous()
        A = tape.a
        a = _lib.DecoderStepArgs()
        for name, p in zip(_PARAM_FIELDS, _step_params(dec)):
            setattr(a, name, _p(p))
        a.prev_op, a.feat, a.h0, a.c0, a.h1, a.c1, a.enc = (prev_op.data_ptr(), feat.data_ptr(), h0.data_ptr(), c0.data_ptr(),
                                                            h1.data_ptr(), c1.data_ptr(), enc.data_ptr())
        attn = A['attn'][slot].view(-1)[:B * L].view(B, L)
        for name in ('step_in', 'hp0', 'hp1', 'gates0', 'gates1', 'h0n', 'c0n', 'h1n', 'c1n', 'mix', 'ctx', 'logp'):
            setattr(a, name, A[name][slot].data_ptr())
        a.attn = attn.data_ptr()
        a.feat_copy = A['featc'][slot].data_ptr()
        a.prev_op_copy = tape.prev_ops[slot].data_ptr()
        a.B, a.L, a.D, a.E, a.V = B, L, D, dec.word_vec_dim, dec.output_size
        _lib.check(_lib.load().t2o_decoder_step_fwd(ctypes.byref(a), _stream(feat.device)), 't2o_decoder_step_fwd')
        ctx.args, ctx.tape, ctx.slot, ctx.dec, ctx.nparams = a, tape, slot, dec, len(params)
        ctx.keep = (prev_op, feat, h0, c0, h1, c1, enc)             # (the argument block holds raw pointers)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(attn)
        return A['logp'][slot], A['h0n'][slot], A['c0n'][slot], A['h1n'][slot], A['c1n'][slot], A['ctx'][slot], attn

    @staticmethod
    def backward(ctx, g_logp, g_h0n, g_c0n, g_h1n, g_c1n, g_ctx, _g_attn):
        a, tape, slot, dec = ctx.args, ctx.tape, ctx.slot, ctx.dec
        B, L, D = a.B, a.L, a.D
        dev = tape.flat.device
        none = (None,) * (4 + ctx.nparams)
        gs = [None if g is None else g.contiguous() for g in (g_logp, g_ctx, g_h0n, g_c0n, g_h1n, g_c1n)]
        if all(g is None for g in gs):
            return (None,) * 6 + none
        if gs[0] is None and gs[1] is None:
            gs[1] = torch.zeros((B, D), dtype=torch.float32, device=dev)
        for name, g in zip(('g_logp', 'g_ctx', 'g_h0n', 'g_c0n', 'g_h1n', 'g_c1n'), gs):
            setattr(a, name, _p(g))
        A = tape.a
        for name in ('d_logits', 'd_lin', 'd_gates1', 'd_gates0', 'd_step_in', 'd_vis'):
            setattr(a, name, A[name][slot].data_ptr())
        for name in _SCRATCH:
            setattr(a, name, A[name].data_ptr())
        out = torch.empty((5, B, D), dtype=torch.float32, device=dev)         # fresh: autograd may keep what it is handed
        d_enc = torch.empty((B, L, D), dtype=torch.float32, device=dev)
        a.d_feat, a.d_h0, a.d_c0, a.d_h1, a.d_c1 = (out[i].data_ptr() for i in range(5))
        a.d_enc = d_enc.data_ptr()
        _lib.check(_lib.load().t2o_decoder_step_bwd(ctypes.byref(a), _stream(dev)), 't2o_decoder_step_bwd')
        tape.step_done.add(slot)
        if gs[0] is not None:
            tape.step_logp.add(slot)
        data = (out[0], out[1], out[2], out[3], out[4], d_enc)
        if tape.persistent:
            return data + none
        got = _collect(lambda sink: tape.step_weight_grads(dec, sink))
        return data + (None, None, None, None) + tuple(got.get(id(p)) if p is not None else None for p in _step_params(dec))


def decoder_step(dec, input_var, hidden, enc, feat, tape=None):
    """Decoder.forward_step on the fused kernels: returns (logp (B,1,V), (h, c) as per-layer lists, attn (B,1,L), ctx (B,D))."""
    h, c = hidden
    h0, h1 = (h[0], h[1])
    c0, c1 = (c[0], c[1])
    B = feat.shape[0]
    slot = tape.take_step() if (tape is not None and tape.B == B and torch.is_grad_enabled()) else None
    if slot is None:
        D = dec.hidden_size
        tape = DecoderTape(B, D, dec.word_vec_dim, dec.output_size, 4, 1, feat.device)
        slot = tape.take_step()
    outs = _DecoderStepFn.apply(feat, h0, c0, h1, c1, enc, input_var, tape, slot, dec, *_step_params(dec))
    logp, h0n, c0n, h1n, c1n, ctx, attn = outs
    return logp.view(B, 1, -1), ([h0n, h1n], [c0n, c1n]), attn.view(B, 1, -1), ctx
