"""Dot-product attention layer with the reference's surface (models/attention.py:5-44).
The score / softmax / mix part (two bmm + softmax in the reference, :37-40) is one HIP kernel
(one wavefront per sample); the output projection stays a GEMM."""
import torch
import torch.nn as nn

from . import functional as T


class Attention(nn.Module):
    def __init__(self, dim, use_weight=False, hidden_size=512):
        super().__init__()
        self.use_weight = use_weight
        self.hidden_size = hidden_size
        if use_weight:
            self.attn_weight = nn.Linear(hidden_size, hidden_size, bias=False)
        self.linear_out = nn.Linear(2 * dim, dim)

    def forward(self, output, context):
        """output (B,1,d) decoder state, context (B,L,d) encoder outputs -> (out (B,1,d), attn (B,1,L)).
        The softmax runs over every encoder row, zero-padded ones included (no padding mask),
        exactly as the reference does."""
        B, T_out, d = output.shape
        if T_out != 1:
            raise NotImplementedError('the decoder advances one step at a time (action_decoder.py:38-64)')
        if self.use_weight:
            output = self.attn_weight(output.contiguous().view(-1, d)).view(B, -1, d)
        q = output.reshape(B, d)
        mix, attn = T.attention_core(q, context)
        comb = torch.cat((mix, q), dim=1)
        lin = T.linear_acc(comb, self.linear_out.weight, self.linear_out.bias) if comb.is_cuda else self.linear_out(comb)
        out = torch.tanh(lin).view(B, 1, d)
        return out, attn.view(B, 1, -1)
