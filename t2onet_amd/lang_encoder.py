"""Request encoder of the actor (counterpart of models/lang_encoder.py:7-113).

`embedding` (+ its `mask_spec` / `mask_word` buffers) and `rnn` keep the reference's names and
shapes so its checkpoints load.  Requests are zero-padded id rows `[START, w_1 .. w_n, END, 0 ..]`;
the 2-layer BiLSTM runs over the packed (length-sorted) batch and its outputs are zero at pads.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

_OWN_LSTM = True      # this library's LSTM step kernels on the GPU (module switch for the tests, which compare with nn.LSTM)


class Embedding(nn.Embedding):
    """Word table whose first `num_spec` rows (NULL/START/END/UNK) always train; with
    `fix_embedding` the remaining (GloVe) rows receive no gradient (lang_encoder.py:7-31)."""

    def __init__(self, num_embeddings, embedding_dim, num_spec, fix_embedding=False):
        super().__init__(num_embeddings, embedding_dim)
        self.fix_embedding = fix_embedding
        is_special = torch.zeros(num_embeddings, embedding_dim)
        is_special[:num_spec] = 1.0
        self.register_buffer('mask_spec', is_special)
        self.register_buffer('mask_word', 1.0 - is_special)

    def forward(self, tokens):
        if not self.fix_embedding:
            return F.embedding(tokens, self.weight)
        trainable = F.embedding(tokens, self.weight * self.mask_spec)
        frozen = F.embedding(tokens, self.weight.detach() * self.mask_word)
        return trainable + frozen


class RNNEncoder(nn.Module):
    def __init__(self, vocab_size, word_embedding_size, hidden_size, n_spec_token, bidirectional=False,
                 input_dropout_p=0, dropout_p=0, n_layers=1, pad_id=0, rnn_type='lstm', variable_lengths=True,
                 word2vec=None, fix_embedding=False):
        super().__init__()
        self.variable_lengths, self.pad_id, self.rnn_type = variable_lengths, pad_id, rnn_type
        self.num_dirs = 2 if bidirectional else 1
        self.embedding = Embedding(vocab_size, word_embedding_size, n_spec_token, fix_embedding)
        if word2vec is not None:                      # pretrained word rows below the special tokens
            if word2vec.shape[0] != vocab_size - n_spec_token:
                raise ValueError('word2vec has %d rows, vocabulary needs %d' % (word2vec.shape[0], vocab_size - n_spec_token))
            with torch.no_grad():
                self.embedding.weight[n_spec_token:].copy_(word2vec)
        self.input_dropout = nn.Dropout(input_dropout_p)
        rnn_cls = {'lstm': nn.LSTM, 'gru': nn.GRU}[rnn_type.lower()]
        self.rnn = rnn_cls(word_embedding_size, hidden_size, n_layers, batch_first=True, bidirectional=bidirectional,
                           dropout=dropout_p)

    def forward(self, input_labels, lengths=None, longest=None):
        """(B,L) ids -> (outputs (B,max_len,h*dirs), final state, embedded tokens).

        `lengths`: CPU tensor of request lengths when the caller has it on the host already (the
        data loader does) -- saves this module's one device-to-host sync.  The reference hands a
        DEVICE tensor to pack_padded_sequence (lang_encoder.py:94), which current torch rejects.
        `longest`: the batch's maximal length as a python int when the caller knows it (a captured hipGraph:
        no host value may depend on device data); `lengths` may then be a device tensor or None."""
        if not self.variable_lengths:
            embedded = self.input_dropout(self.embedding(input_labels))
            outputs, state = self.rnn(embedded)
            return outputs, state, embedded
        dev = input_labels.device
        if dev.type == 'cuda' and _OWN_LSTM and isinstance(self.rnn, nn.LSTM) and self.rnn.proj_size == 0 \
                and self.rnn.batch_first and self.rnn.hidden_size % 64 == 0 and self.rnn.hidden_size <= 256 and input_labels.shape[0] > 0:
            # GPU: this library's LSTM step kernels (functional.lstm_layer): one input GEMM + one launch per time step
            # and layer for both directions, lengths stay on the device (no sort, no packing), static shapes for a
            # given `longest` (hipGraph-capturable).  The library's sequence call is ~450 launches per train step.
            dev_lengths = lengths if (lengths is not None and lengths.device == dev) else (input_labels != self.pad_id).sum(dim=1)
            if longest is None:
                longest = int(lengths.max()) if lengths is not None else int(dev_lengths.max())
            embedded = self.input_dropout(self.embedding(input_labels[:, :longest]))
            from . import functional as T
            rnn, D = self.rnn, self.num_dirs
            layer_in, hs, cs = embedded, [], []
            for layer in range(rnn.num_layers):
                dirs = []
                for d in range(D):
                    sfx = '_l%d%s' % (layer, '_reverse' if d else '')
                    names = ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh') if rnn.bias else ('weight_ih', 'weight_hh')
                    dirs.append(tuple(getattr(rnn, n + sfx) for n in names))
                layer_in, h, c = T.lstm_layer(layer_in, dev_lengths, dirs)
                hs.append(h)
                cs.append(c)
                if layer + 1 < rnn.num_layers and rnn.dropout > 0 and self.training:
                    layer_in = F.dropout(layer_in, rnn.dropout, True)
            return layer_in, (torch.cat(hs, 0), torch.cat(cs, 0)), embedded
        if lengths is None:
            lengths = (input_labels != self.pad_id).sum(dim=1).cpu()
        lengths = lengths.cpu()
        # the same STABLE descending order computed twice -- on the host for pack_padded_sequence's lengths,
        # on the device for the gathers -- instead of copying the permutation to the device (a blocking copy
        # that drains the GPU queue at the top of every step)
        by_length = torch.argsort(lengths, descending=True, stable=True)
        dev_lengths = (input_labels != self.pad_id).sum(dim=1) if lengths.device != dev else lengths
        by_length_dev = torch.argsort(dev_lengths, descending=True, stable=True)
        undo = torch.argsort(by_length_dev)
        longest = int(lengths[by_length[0]])
        ordered = input_labels[by_length_dev, :longest]
        embedded = self.input_dropout(self.embedding(ordered))
        packed = pack_padded_sequence(embedded, lengths[by_length], batch_first=True)
        packed_out, state = self.rnn(packed)
        outputs = pad_packed_sequence(packed_out, batch_first=True)[0][undo]
        if isinstance(state, tuple):
            state = tuple(s[:, undo] for s in state)
        else:
            state = state[:, undo]
        return outputs, state, embedded
