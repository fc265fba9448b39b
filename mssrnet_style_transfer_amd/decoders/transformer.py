"""Transformer decoder on the gfx950 kernels.

Drop-in for onmt/decoders/transformer.py:14-357: ``TransformerDecoderLayer`` (attributes
``self_attn, context_attn, feed_forward, layer_norm_1, layer_norm_2, drop``) and
``TransformerDecoder`` (``embeddings, state, transformer_layers, layer_norm``;
``init_state / map_state / detach_state / forward(tgt, memory_bank, step=None,
memory_lengths=...) -> (dec_outs [T,B,d], {"std": attn [T,B,L]})``) including the step-wise
``state["cache"]["layer_i"]`` dict protocol used by the soft decode and by greedy search.
"""
import torch
import torch.nn as nn

from .. import ops
from ..modules.multi_headed_attn import MultiHeadedAttention
from ..modules.position_ffn import PositionwiseFeedForward
from ..utils.misc import sequence_mask
from .decoder import DecoderBase


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, heads, d_ff, dropout, attention_dropout, self_attn_type="scaled-dot",
                 max_relative_positions=0, aan_useffn=False, full_context_alignment=False, alignment_heads=0):
        super().__init__()
        if self_attn_type != "scaled-dot":
            raise NotImplementedError("MSSRNet uses scaled-dot self attention only")
        self.self_attn = MultiHeadedAttention(heads, d_model, dropout=attention_dropout,
                                              max_relative_positions=max_relative_positions)
        self.context_attn = MultiHeadedAttention(heads, d_model, dropout=attention_dropout)
        self.feed_forward = PositionwiseFeedForward(d_model, d_ff, dropout)
        self.layer_norm_1 = nn.LayerNorm(d_model, eps=1e-6)
        self.layer_norm_2 = nn.LayerNorm(d_model, eps=1e-6)
        self.drop = nn.Dropout(dropout)
        self.full_context_alignment = full_context_alignment
        self.alignment_heads = alignment_heads

    def forward(self, *args, **kwargs):
        """Returns (output [B,T,d], top_attn [B,T,L] = head 0 of the context attention, attn_align)."""
        with_align = kwargs.pop("with_align", False)
        output, attns = self._forward(*args, **kwargs)
        top_attn = attns[:, 0, :, :].contiguous()
        attn_align = None
        if with_align:
            if self.full_context_alignment:
                _, attns = self._forward(*args, **kwargs, future=True)
            if self.alignment_heads > 0:
                attns = attns[:, :self.alignment_heads, :, :].contiguous()
            attn_align = attns.mean(dim=1)
        return output, top_attn, attn_align

    def _forward(self, inputs, memory_bank, src_pad_mask, tgt_pad_mask, layer_cache=None, step=None,
                 future=False):
        dec_mask = None
        if step is None:
            tgt_len = tgt_pad_mask.size(-1)
            if not future:
                future_mask = torch.ones([tgt_len, tgt_len], device=tgt_pad_mask.device,
                                         dtype=torch.bool).triu_(1).view(1, tgt_len, tgt_len)
                dec_mask = tgt_pad_mask | future_mask
            else:
                dec_mask = tgt_pad_mask
        ln1, ln2 = self.layer_norm_1, self.layer_norm_2
        input_norm = ops.layer_norm(inputs, ln1.weight, ln1.bias, ln1.eps)
        query, _ = self.self_attn(input_norm, input_norm, input_norm, mask=dec_mask, layer_cache=layer_cache,
                                  attn_type="self")
        query = ops.dropout(query, self.drop.p, self.training, residual=inputs)
        query_norm = ops.layer_norm(query, ln2.weight, ln2.bias, ln2.eps)
        mid, attns = self.context_attn(memory_bank, memory_bank, query_norm, mask=src_pad_mask,
                                       layer_cache=layer_cache, attn_type="context")
        output = self.feed_forward(ops.dropout(mid, self.drop.p, self.training, residual=query))
        return output, attns

    def update_dropout(self, dropout, attention_dropout):
        self.self_attn.update_dropout(attention_dropout)
        self.context_attn.update_dropout(attention_dropout)
        self.feed_forward.update_dropout(dropout)
        self.drop.p = dropout


class TransformerDecoder(DecoderBase):
    def __init__(self, num_layers, d_model, heads, d_ff, copy_attn, self_attn_type, dropout, attention_dropout,
                 embeddings, max_relative_positions, aan_useffn, full_context_alignment, alignment_layer,
                 alignment_heads):
        super().__init__()
        self.embeddings = embeddings
        self.state = {}
        self.transformer_layers = nn.ModuleList(
            [TransformerDecoderLayer(d_model, heads, d_ff, dropout, attention_dropout,
                                     self_attn_type=self_attn_type,
                                     max_relative_positions=max_relative_positions, aan_useffn=aan_useffn,
                                     full_context_alignment=full_context_alignment,
                                     alignment_heads=alignment_heads)
             for _ in range(num_layers)])
        self._copy = copy_attn
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self.alignment_layer = alignment_layer

    @classmethod
    def from_opt(cls, opt, embeddings):
        return cls(opt.dec_layers, opt.dec_rnn_size, opt.heads, opt.transformer_ff, opt.copy_attn,
                   opt.self_attn_type,
                   opt.dropout[0] if type(opt.dropout) is list else opt.dropout,
                   opt.attention_dropout[0] if type(opt.attention_dropout) is list else opt.attention_dropout,
                   embeddings, opt.max_relative_positions, opt.aan_useffn, opt.full_context_alignment,
                   opt.alignment_layer, alignment_heads=opt.alignment_heads)

    def init_state(self, src, memory_bank, enc_hidden):
        self.state["src"] = src
        self.state["cache"] = None

    def map_state(self, fn):
        def _recursive_map(struct, batch_dim=0):
            for k, v in struct.items():
                if v is not None:
                    if isinstance(v, dict):
                        _recursive_map(v)
                    else:
                        struct[k] = fn(v, batch_dim)

        self.state["src"] = fn(self.state["src"], 1)
        if self.state["cache"] is not None:
            _recursive_map(self.state["cache"])

    def detach_state(self):
        self.state["src"] = self.state["src"].detach()

    def forward(self, tgt, memory_bank, step=None, **kwargs):
        if step == 0:
            self._init_cache(memory_bank)
        tgt_words = tgt[:, :, 0].transpose(0, 1)
        emb = self.embeddings(tgt, step=step)
        assert emb.dim() == 3
        output = emb.transpose(0, 1).contiguous()
        src_memory_bank = memory_bank.transpose(0, 1).contiguous()
        pad_idx = self.embeddings.word_padding_idx
        src_lens = kwargs["memory_lengths"]
        src_max_len = self.state["src"].shape[0]
        src_pad_mask = ~sequence_mask(src_lens, src_max_len).unsqueeze(1)
        # float tgt (soft inputs): compares the probability of word 0 with pad_idx, as the reference does
        tgt_pad_mask = tgt_words.detach().eq(pad_idx).unsqueeze(1)
        with_align = kwargs.pop("with_align", False)
        attn_aligns = []
        attn = None
        for i, layer in enumerate(self.transformer_layers):
            layer_cache = self.state["cache"]["layer_{}".format(i)] if step is not None else None
            output, attn, attn_align = layer(output, src_memory_bank, src_pad_mask, tgt_pad_mask,
                                             layer_cache=layer_cache, step=step, with_align=with_align)
            if attn_align is not None:
                attn_aligns.append(attn_align)
        ln = self.layer_norm
        output = ops.layer_norm(output, ln.weight, ln.bias, ln.eps)
        dec_outs = output.transpose(0, 1).contiguous()
        attn = attn.transpose(0, 1).contiguous()
        attns = {"std": attn}
        if self._copy:
            attns["copy"] = attn
        if with_align:
            attns["align"] = attn_aligns[self.alignment_layer]
        return dec_outs, attns

    def _init_cache(self, memory_bank):
        self.state["cache"] = {}
        for i, layer in enumerate(self.transformer_layers):
            self.state["cache"]["layer_{}".format(i)] = {"memory_keys": None, "memory_values": None,
                                                         "self_keys": None, "self_values": None}

    def update_dropout(self, dropout, attention_dropout):
        self.embeddings.update_dropout(dropout)
        for layer in self.transformer_layers:
            layer.update_dropout(dropout, attention_dropout)
