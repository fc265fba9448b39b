"""Embeddings + PositionalEncoding on the gfx950 kernels.

Drop-in for onmt/modules/embeddings.py:11-54,89-292 (single word feature, which is all
MSSRNet uses): state-dict keys ``make_embedding.emb_luts.0.weight`` and the buffer
``make_embedding.pe.pe`` [5000,1,d]; ``forward(source, step=None)`` takes ids [len,B,1] or
the fork's soft inputs [len,B,V] (probabilities times the FULL table, padding row included).
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .util_class import Elementwise


class PositionalEncoding(nn.Module):
    def __init__(self, dropout, dim, max_len=5000):
        if dim % 2 != 0:
            raise ValueError("Cannot use sin/cos positional encoding with odd dim (got dim={:d})".format(dim))
        pe = torch.zeros(max_len, dim)
        position = torch.arange(0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float) * -(math.log(10000.0) / dim))
        pe[:, 0::2] = torch.sin(position.float() * div_term)
        pe[:, 1::2] = torch.cos(position.float() * div_term)
        super().__init__()
        self.register_buffer("pe", pe.unsqueeze(1))
        self.dropout = nn.Dropout(p=dropout)
        self.dim = dim

    def table(self):
        pe = self.pe
        return (pe if pe.dtype == torch.float32 else pe.float()).view(pe.size(0), self.dim)

    def forward(self, emb, step=None):
        """emb*sqrt(dim) + pe[:len] (or pe[step]), then dropout."""
        out = ops.scale_add_pe(emb, self.table(), step, math.sqrt(self.dim))
        return ops.dropout(out, self.dropout.p, self.training)


class VecEmbedding(nn.Module):
    """Embedding of vector sequences [len, B, 1, vec_size] -> [len, B, emb_dim] (a bias-free projection, then the
    positional encoding).  No MSSRNet configuration builds one (onmt/modules/embeddings.py:57-88 serves the
    ``vec`` data type); it exists because ``onmt/modules/__init__.py:9-10`` imports the name from this module path."""

    def __init__(self, vec_size, emb_dim, position_encoding=False, dropout=0):
        super().__init__()
        self.embedding_size = emb_dim
        self.word_padding_idx = 0
        self.position_encoding = position_encoding
        self.proj = nn.Linear(vec_size, emb_dim, bias=False)
        if position_encoding:
            self.pe = PositionalEncoding(dropout, emb_dim)

    def forward(self, x, step=None):
        out = ops.linear(x.squeeze(2), self.proj.weight)
        return self.pe(out, step=step) if self.position_encoding else out

    def load_pretrained_vectors(self, file):
        if file:
            raise ValueError("VecEmbedding has no pretrained vectors to load")


class Embeddings(nn.Module):
    def __init__(self, word_vec_size, word_vocab_size, word_padding_idx, position_encoding=False,
                 feat_merge="concat", feat_vec_exponent=0.7, feat_vec_size=-1, feat_padding_idx=None,
                 feat_vocab_sizes=None, dropout=0, sparse=False, fix_word_vecs=False):
        if feat_vocab_sizes:
            raise NotImplementedError("extra word features are not used by MSSRNet")
        if not position_encoding:
            raise ValueError("Not support anymore!")       # as the reference (embeddings.py:285)
        super().__init__()
        self.word_padding_idx = word_padding_idx
        self.word_vec_size = word_vec_size
        self.embedding_size = word_vec_size
        lut = nn.Embedding(word_vocab_size, word_vec_size, padding_idx=word_padding_idx, sparse=sparse)
        self.make_embedding = nn.Sequential()
        self.make_embedding.add_module("emb_luts", Elementwise(feat_merge, [lut]))
        self.position_encoding = position_encoding
        self.make_embedding.add_module("pe", PositionalEncoding(dropout, self.embedding_size))
        if fix_word_vecs:
            self.word_lut.weight.requires_grad = False

    @property
    def word_lut(self):
        return self.make_embedding[0][0]

    @property
    def emb_luts(self):
        return self.make_embedding[0]

    def load_pretrained_vectors(self, emb_file):
        if emb_file:
            pretrained = torch.load(emb_file, weights_only=True)
            n = min(self.word_vec_size, pretrained.size(1))
            self.word_lut.weight.data[:, :n] = pretrained[:, :n]

    def forward(self, source, step=None, eval_rows=0):
        """eval_rows (no reference counterpart): sentences [0, eval_rows) of the batch are embedded without dropout
        even in train mode (eval-mode rows batched with training rows, see Trainer.train_wgan_step)."""
        pe = self.make_embedding[1]
        weight = self.word_lut.weight
        s = math.sqrt(pe.dim)
        if source.dim() > 2 and source.size(-1) > 1:           # soft path (embeddings.py:278-280)
            ln, B, V = source.shape
            emb = ops.matmul(source.reshape(ln * B, V), weight, tb=False).view(ln, B, -1)
            out = ops.scale_add_pe(emb, pe.table(), step, s)
        else:                                                    # fused lookup * sqrt(d) + pe
            out = ops.embed_pe(source[:, :, 0], weight, pe.table(), step, s, self.word_padding_idx)
        if eval_rows and pe.training and pe.dropout.p > 0:
            return torch.cat([out[:, :eval_rows], ops.dropout(out[:, eval_rows:].contiguous(), pe.dropout.p, True)], 1)
        return ops.dropout(out, pe.dropout.p, pe.training)

    def update_dropout(self, dropout):
        self.make_embedding[1].dropout.p = dropout
