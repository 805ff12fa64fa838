"""One-step attention LSTM decoder (models/action_decoder.py:9-78): same module names, so the
reference's state_dict loads.  The LSTM / Linear GEMMs run on MIOpen / hipBLASLt through
PyTorch-ROCm (MFMA); the attention core is the HIP kernel in attention.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .attention import Attention


class Decoder(nn.Module):
    def __init__(self, vocab_size, max_len, word_vec_dim, hidden_size, n_layers, rnn_type='lstm',
                 bidirectional=False, input_dropout_p=0, dropout_p=0, use_attention=False):
        super().__init__()
        self.max_length = max_len
        self.output_size = vocab_size
        self.hidden_size = hidden_size * 2 if bidirectional else hidden_size
        self.word_vec_dim = word_vec_dim
        self.bidirectional_encoder = bidirectional
        self.use_attention = use_attention
        self.embedding = nn.Embedding(self.output_size, self.word_vec_dim)
        self.rnn = getattr(nn, rnn_type.upper())(self.word_vec_dim + self.hidden_size, self.hidden_size, n_layers,
                                                 batch_first=True, dropout=dropout_p)
        self.out_linear = nn.Linear(self.hidden_size, self.output_size)
        self.vis_linear = nn.Linear(self.hidden_size, self.hidden_size)
        if use_attention:
            self.attention = Attention(self.hidden_size)
        self.input_dropout = nn.Dropout(p=input_dropout_p)

    def forward_step(self, input_var, hidden, encoder_outputs, img_feat):
        """input_var (B,1) previous operator id -> (log-probs (B,1,n_cls), hidden, attn (B,1,L), context (B,d))."""
        B = input_var.size(0)
        vis_feat = F.relu(self.vis_linear(img_feat))
        embedded = torch.cat((self.embedding(input_var), vis_feat.view(B, 1, -1)), 2)
        embedded = self.input_dropout(embedded)
        context, hidden = self.rnn(embedded, hidden)
        attn = None
        if self.use_attention:
            context, attn = self.attention(context, encoder_outputs)
        logits = self.out_linear(context.contiguous().view(-1, self.hidden_size))
        return F.log_softmax(logits.view(B, 1, -1), -1), hidden, attn, context.squeeze(1)

    def _init_state(self, encoder_hidden):
        if encoder_hidden is None:
            return None
        if isinstance(encoder_hidden, tuple):
            return tuple(self._cat_directions(h) for h in encoder_hidden)
        return self._cat_directions(encoder_hidden)

    def _cat_directions(self, h):
        if self.bidirectional_encoder:                       # (2L,B,h) -> (L,B,2h)
            h = torch.cat([h[0:h.size(0):2], h[1:h.size(0):2]], 2)
        return h
