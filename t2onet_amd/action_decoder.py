"""Operator decoder of the actor: one decoding step at a time.

Counterpart of models/action_decoder.py:9-78.  What has to stay for checkpoints and callers:
the sub-module names (`embedding`, `rnn`, `out_linear`, `vis_linear`, `attention`,
`input_dropout`) and the two entry points `forward_step` / `_init_state`.

Per step:  [operator embedding (300) | relu(vis_linear(image feature)) (512)]  ->  2-layer LSTM
(812 -> 512)  ->  dot-product attention over the request encoding (HIP kernel, attention.py)
->  out_linear  ->  log-softmax over the 11 operator tokens.  On the GPU the step is 7 launches of this library
(decoder_step.py / t2o_decoder.hip: every Linear, the LSTM cells with their gates, tanh and log-softmax fused into small-M
matrix-core products); the per-layer path below (library GEMMs plus PyTorch's fused gate kernel) remains for
configurations the fused step does not take (GRU, active dropout, attention with a weight) and for host tensors, which
only the multi-process CPU tests of the data-parallel logic use.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as T
from .attention import Attention


_FUSED_STEP = True        # module switch for the tests: False sends GPU calls through the per-layer path below


def merge_directions(state):
    """(layers*2, B, h) bidirectional encoder state -> (layers, B, 2h): forward and backward halves of
    each layer side by side (action_decoder.py:75-78)."""
    fwd, bwd = state[0::2], state[1::2]
    return torch.cat((fwd, bwd), dim=2)


class Decoder(nn.Module):
    def __init__(self, vocab_size, max_len, word_vec_dim, hidden_size, n_layers, rnn_type='lstm',
                 bidirectional=False, input_dropout_p=0, dropout_p=0, use_attention=False):
        super().__init__()
        width = 2 * hidden_size if bidirectional else hidden_size      # the encoder's output width
        self.max_length, self.output_size, self.word_vec_dim = max_len, vocab_size, word_vec_dim
        self.hidden_size, self.bidirectional_encoder, self.use_attention = width, bidirectional, use_attention
        rnn_cls = {'lstm': nn.LSTM, 'gru': nn.GRU}[rnn_type.lower()]
        self.embedding = nn.Embedding(vocab_size, word_vec_dim)
        self.rnn = rnn_cls(word_vec_dim + width, width, n_layers, batch_first=True, dropout=dropout_p)
        self.out_linear = nn.Linear(width, vocab_size)
        self.vis_linear = nn.Linear(width, width)
        if use_attention:
            self.attention = Attention(width)
        self.input_dropout = nn.Dropout(p=input_dropout_p)

    def forward_step(self, input_var, hidden, encoder_outputs, img_feat):
        """input_var (B,1) previous operator token, hidden LSTM state, encoder_outputs (B,L,d),
        img_feat (B,d).  Returns (log-probabilities (B,1,n_tokens), new hidden, attention (B,1,L) or
        None, context (B,d))."""
        n = input_var.shape[0]
        gpu = img_feat.is_cuda and img_feat.dtype == torch.float32
        if gpu and _FUSED_STEP:
            from . import decoder_step as DS
            if DS.step_supported(self, input_var, hidden, encoder_outputs, img_feat):
                # the whole step on this library's small-M kernels: 7 launches forward, 10 backward (t2o_decoder.hip)
                stacked = torch.is_tensor(hidden[0])
                state = (hidden[0].unbind(0), hidden[1].unbind(0)) if stacked else hidden
                logp, state, attn, context = DS.decoder_step(self, input_var, state, encoder_outputs, img_feat, self.__dict__.get('_tape'))
                if stacked:
                    state = (torch.stack(state[0], 0), torch.stack(state[1], 0))
                return logp, state, attn, context
        token = self.embedding(input_var)                                   # (B,1,300)
        vis = T.linear_acc(img_feat, self.vis_linear.weight, self.vis_linear.bias) if gpu else self.vis_linear(img_feat)
        visual = F.relu(vis).unsqueeze(1)                                   # (B,1,d)
        step_in = self.input_dropout(torch.cat((token, visual), dim=2))
        context, hidden = self._rnn_step(step_in, hidden)
        attn = None
        if self.use_attention:
            context, attn = self.attention(context, encoder_outputs)
        flat = context.reshape(n, self.hidden_size)
        scores = T.linear_acc(flat, self.out_linear.weight, self.out_linear.bias) if gpu else self.out_linear(flat)
        return F.log_softmax(scores, dim=-1).view(n, 1, -1), hidden, attn, context.squeeze(1)

    def _rnn_step(self, step_in, hidden):
        """One decoding step of self.rnn.  For the LSTM this is num_layers fused cell calls (two GEMMs + one
        gate kernel each, the arithmetic of nn.LSTM on a length-1 sequence) instead of the library's sequence
        entry point, whose per-call host cost (~1.8 ms forward, more backward: descriptors, workspaces) made
        the five decoder steps the largest host-side item of the train step."""
        rnn = self.rnn
        if not (isinstance(rnn, nn.LSTM) and step_in.shape[1] == 1 and isinstance(hidden, (tuple, list))
                and not rnn.bidirectional and rnn.proj_size == 0 and (rnn.dropout == 0 or not self.training)):
            return rnn(step_in, hidden)
        h0, c0 = hidden
        unstacked = isinstance(h0, (list, tuple))       # per-layer tensors in, per-layer tensors out (no stack / select pairs:
        x = step_in[:, 0]                               # their backward is a zero fill + copies + adds per decoder step)
        hs, cs = [], []
        gpu = x.is_cuda and x.dtype == torch.float32
        for layer in range(rnn.num_layers):
            w = [getattr(rnn, '%s_l%d' % (name, layer)) for name in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')] \
                if rnn.bias else [getattr(rnn, 'weight_ih_l%d' % layer), getattr(rnn, 'weight_hh_l%d' % layer), None, None]
            if gpu:
                x, c = T.lstm_cell_acc(x, h0[layer], c0[layer], w[0], w[1], w[2], w[3])
            else:
                x, c = torch._VF.lstm_cell(x, (h0[layer], c0[layer]), w[0], w[1], w[2], w[3])
            hs.append(x)
            cs.append(c)
        if unstacked:
            return x.unsqueeze(1), (hs, cs)
        return x.unsqueeze(1), (torch.stack(hs, 0), torch.stack(cs, 0))

    def _init_state(self, encoder_hidden):
        """Encoder final state -> decoder initial state."""
        if encoder_hidden is None:
            return None
        if not self.bidirectional_encoder:
            return encoder_hidden
        if isinstance(encoder_hidden, tuple):
            return tuple(merge_directions(s) for s in encoder_hidden)
        return merge_directions(encoder_hidden)
