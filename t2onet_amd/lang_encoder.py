"""Request encoder (models/lang_encoder.py): masked embedding + packed 2-layer BiLSTM.
Module / buffer names match the reference's state_dict."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Embedding(nn.Embedding):
    """With fix_embedding, word rows are frozen and only the num_spec special-token rows train
    (lang_encoder.py:7-31)."""

    def __init__(self, num_embeddings, embedding_dim, num_spec, fix_embedding=False):
        super().__init__(num_embeddings, embedding_dim)
        self.fix_embedding = fix_embedding
        spec = torch.cat([torch.ones(num_spec, embedding_dim), torch.zeros(num_embeddings - num_spec, embedding_dim)])
        self.register_buffer('mask_spec', spec)
        self.register_buffer('mask_word', 1 - spec)

    def forward(self, tokens):
        if not self.fix_embedding:
            return F.embedding(tokens, self.weight)
        return F.embedding(tokens, self.weight * self.mask_spec) + \
            F.embedding(tokens, self.weight.detach() * self.mask_word)


class RNNEncoder(nn.Module):
    def __init__(self, vocab_size, word_embedding_size, hidden_size, n_spec_token, bidirectional=False,
                 input_dropout_p=0, dropout_p=0, n_layers=1, pad_id=0, rnn_type='lstm', variable_lengths=True,
                 word2vec=None, fix_embedding=False):
        super().__init__()
        self.variable_lengths = variable_lengths
        self.embedding = Embedding(vocab_size, word_embedding_size, n_spec_token, fix_embedding)
        if word2vec is not None:                               # GloVe rows under the special tokens
            assert word2vec.shape[0] == vocab_size - n_spec_token
            with torch.no_grad():
                self.embedding.weight[n_spec_token:] = word2vec
        self.input_dropout = nn.Dropout(input_dropout_p)
        self.pad_id = pad_id
        self.rnn_type = rnn_type
        self.rnn = getattr(nn, rnn_type.upper())(word_embedding_size, hidden_size, n_layers, batch_first=True,
                                                 bidirectional=bidirectional, dropout=dropout_p)
        self.num_dirs = 2 if bidirectional else 1

    def forward(self, input_labels, lengths=None):
        """input_labels (B,L) zero-padded ids -> (output (B,maxlen,h*dirs) zero at pads, hidden, embedded).
        `lengths` (CPU int tensor) may be passed when the caller already has it on the host (the
        data loader does): it avoids the one device->host sync of this module.  (The reference
        hands a device tensor to pack_padded_sequence, which current torch rejects.)"""
        if self.variable_lengths:
            if lengths is None:
                lengths = (input_labels != self.pad_id).sum(1).cpu()
            sorted_len, sort_ix = lengths.sort(descending=True)
            recover_ix = sort_ix.argsort().to(input_labels.device)
            input_labels = input_labels[:, :int(sorted_len[0])][sort_ix.to(input_labels.device)]
        embedded = self.input_dropout(self.embedding(input_labels))
        if not self.variable_lengths:
            output, hidden = self.rnn(embedded)
            return output, hidden, embedded
        packed = nn.utils.rnn.pack_padded_sequence(embedded, sorted_len, batch_first=True)
        output, hidden = self.rnn(packed)
        output, _ = nn.utils.rnn.pad_packed_sequence(output, batch_first=True)
        output = output[recover_ix]
        if self.rnn_type == 'lstm':
            hidden = (hidden[0][:, recover_ix, :], hidden[1][:, recover_ix, :])
        else:
            hidden = hidden[:, recover_ix, :]
        return output, hidden, embedded
