"""Request encoder of the actor (counterpart of models/lang_encoder.py:7-113).

`embedding` (+ its `mask_spec` / `mask_word` buffers) and `rnn` keep the reference's names and
shapes so its checkpoints load.  Requests are zero-padded id rows `[START, w_1 .. w_n, END, 0 ..]`;
the 2-layer BiLSTM runs over the packed (length-sorted) batch and its outputs are zero at pads.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence


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

    def forward(self, input_labels, lengths=None):
        """(B,L) ids -> (outputs (B,max_len,h*dirs), final state, embedded tokens).

        `lengths`: CPU tensor of request lengths when the caller has it on the host already (the
        data loader does) -- saves this module's one device-to-host sync.  The reference hands a
        DEVICE tensor to pack_padded_sequence (lang_encoder.py:94), which current torch rejects."""
        if not self.variable_lengths:
            embedded = self.input_dropout(self.embedding(input_labels))
            outputs, state = self.rnn(embedded)
            return outputs, state, embedded
        dev = input_labels.device
        if lengths is None:
            lengths = (input_labels != self.pad_id).sum(dim=1).cpu()
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
