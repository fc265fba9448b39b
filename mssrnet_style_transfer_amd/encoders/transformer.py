"""Transformer encoder on the gfx950 kernels.

Drop-in for onmt/encoders/transformer.py:13-141: ``TransformerEncoderLayer`` (attributes
``self_attn, feed_forward, layer_norm, dropout``) and ``TransformerEncoder`` (attributes
``embeddings, transformer, layer_norm``; ``forward(src, lengths) -> (emb, memory_bank,
lengths)``), pre-norm layers, LayerNorm eps 1e-6.  ``StyleGenerator2`` reaches into
``embeddings``, ``transformer`` and ``layer_norm`` directly, so those stay plain attributes.
"""
import torch.nn as nn

from .. import ops
from ..encoder_engine import run_stack
from ..modules.multi_headed_attn import MultiHeadedAttention
from ..modules.position_ffn import PositionwiseFeedForward
from ..utils.misc import padding_mask
from .encoder import EncoderBase


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, heads, d_ff, dropout, attention_dropout, max_relative_positions=0):
        super().__init__()
        self.self_attn = MultiHeadedAttention(heads, d_model, dropout=attention_dropout,
                                              max_relative_positions=max_relative_positions)
        self.feed_forward = PositionwiseFeedForward(d_model, d_ff, dropout)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.dropout = nn.Dropout(dropout)

    def forward(self, inputs, mask):
        """inputs [B, L, d], mask Bool [B, 1, L] (True = padding) -> [B, L, d]."""
        ln = self.layer_norm
        input_norm = ops.layer_norm(inputs, ln.weight, ln.bias, ln.eps)
        context, _ = self.self_attn(input_norm, input_norm, input_norm, mask=mask, attn_type="self")
        out = ops.dropout(context, self.dropout.p, self.training, residual=inputs)
        return self.feed_forward(out)

    def update_dropout(self, dropout, attention_dropout):
        self.self_attn.update_dropout(attention_dropout)
        self.feed_forward.update_dropout(dropout)
        self.dropout.p = dropout


class TransformerEncoder(EncoderBase):
    def __init__(self, num_layers, d_model, heads, d_ff, dropout, attention_dropout, embeddings,
                 max_relative_positions):
        super().__init__()
        self.embeddings = embeddings
        self.transformer = nn.ModuleList(
            [TransformerEncoderLayer(d_model, heads, d_ff, dropout, attention_dropout,
                                     max_relative_positions=max_relative_positions)
             for _ in range(num_layers)])
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)

    @classmethod
    def from_opt(cls, opt, embeddings):
        return cls(opt.enc_layers, opt.enc_rnn_size, opt.heads, opt.transformer_ff,
                   opt.dropout[0] if type(opt.dropout) is list else opt.dropout,
                   opt.attention_dropout[0] if type(opt.attention_dropout) is list else opt.attention_dropout,
                   embeddings, opt.max_relative_positions)

    def forward(self, src, lengths=None, eval_rows=0):
        """src: Long [L, B, 1] or float [L, B, V]; lengths Long [B].  eval_rows (no reference counterpart): the first
        eval_rows sentences are encoded without dropout even in train mode."""
        self._check_args(src, lengths)
        if eval_rows:
            return self.forward_embedded(self.embeddings(src, eval_rows=eval_rows), lengths, eval_rows)
        return self.forward_embedded(self.embeddings(src), lengths)

    def forward_embedded(self, emb, lengths, eval_rows=0):
        """The encoder behind its embedding layer: emb [L, B, d] (``self.embeddings`` of ids or of soft words, possibly
        of several inputs concatenated along the batch) -> (emb, memory_bank, lengths).  No reference counterpart as a
        method: it is the body of onmt/encoders/transformer.py:118-129 after the embedding call."""
        pad = padding_mask(lengths, emb.size(0))                 # [B, L], True = padding
        out = run_stack(self, emb.transpose(0, 1), pad, eval_rows)         # all layers + final LN: one C call
        if out is None and eval_rows:
            raise RuntimeError("TransformerEncoder: eval_rows needs the encoder engine (not available for this graph / shape)")
        if out is None:                                                    # second-order graphs, long sequences
            out = emb.transpose(0, 1).contiguous()
            mask = pad.unsqueeze(1)
            for layer in self.transformer:
                out = layer(out, mask)
            ln = self.layer_norm
            out = ops.layer_norm(out, ln.weight, ln.bias, ln.eps)
        return emb, out.transpose(0, 1).contiguous(), lengths

    def update_dropout(self, dropout, attention_dropout):
        self.embeddings.update_dropout(dropout)
        for layer in self.transformer:
            layer.update_dropout(dropout, attention_dropout)
