"""PositionwiseFeedForward on the gfx950 kernels (drop-in for onmt/modules/position_ffn.py:6-41)."""
import torch.nn as nn

from .. import ops


class PositionwiseFeedForward(nn.Module):
    """x + D2(W2 D1(relu(W1 LN(x)))); bias+ReLU and bias+residual ride in the GEMM epilogues."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.dropout_1 = nn.Dropout(dropout)
        self.relu = nn.ReLU()
        self.dropout_2 = nn.Dropout(dropout)

    def forward(self, x):
        ln = self.layer_norm
        h = ops.layer_norm(x, ln.weight, ln.bias, ln.eps)
        inter = ops.linear(h, self.w_1.weight, self.w_1.bias, relu=True)
        inter = ops.dropout(inter, self.dropout_1.p, self.training)
        if self.training and self.dropout_2.p > 0:
            out = ops.linear(inter, self.w_2.weight, self.w_2.bias)
            return ops.dropout(out, self.dropout_2.p, True, residual=x)
        return ops.linear(inter, self.w_2.weight, self.w_2.bias, residual=x)

    def update_dropout(self, dropout):
        self.dropout_1.p = dropout
        self.dropout_2.p = dropout
