"""Transformer decoder on the gfx950 kernels.

Drop-in for the two classes of onmt/decoders/transformer.py:14-357.  What the reference's callers
rely on, and therefore what is kept: the class names, the constructor argument order, the
attribute names that end up in checkpoints (``self_attn, context_attn, feed_forward,
layer_norm_1, layer_norm_2, drop`` per layer; ``embeddings, transformer_layers, layer_norm``
on the decoder), the ``state`` dict with its ``"src"`` entry and the per-layer
``state["cache"]["layer_i"]`` dicts that MultiHeadedAttention fills, and
``forward(tgt, memory_bank, step=None, memory_lengths=...) -> (dec_outs [T,B,d],
{"std": attn [T,B,L]})``.

What MSSRNet never turns on is NOT carried over: guided alignment (``with_align``,
``full_context_alignment``, ``alignment_heads``, ``lambda_align = 0`` in every README config) and the
copy-attention output.  The constructor still accepts and records those arguments (they are part of
the pickled attribute set) and a call that asks for them fails loudly instead of silently
returning something else.
"""
import torch
import torch.nn as nn

from .. import ops
from ..modules.multi_headed_attn import MultiHeadedAttention
from ..modules.position_ffn import PositionwiseFeedForward
from ..utils.misc import padding_mask
from ..decoder_stack import run_stack
from .decoder import DecoderBase

_CACHE_SLOTS = ("memory_keys", "memory_values", "self_keys", "self_values")


def _first(value):
    """OpenNMT options hold dropout rates as one-element lists."""
    return value[0] if isinstance(value, list) else value


def _norm(ln, x):
    return ops.layer_norm(x, ln.weight, ln.bias, ln.eps)


class TransformerDecoderLayer(nn.Module):
    """Pre-norm decoder layer: x + D(self-attn(LN1 x)) -> h + D(ctx-attn(LN2 h)) -> FFN
    (onmt/decoders/transformer.py:110-171)."""

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
        # recorded only: the alignment branches of the reference (:96-107) are not reproduced
        self.full_context_alignment = full_context_alignment
        self.alignment_heads = alignment_heads

    def _self_mask(self, tgt_pad_mask, step):
        """Causal + target-padding mask for a whole sequence; None while decoding step by step, where
        the cache holds only the past (reference :132-149)."""
        if step is not None:
            return None
        T = tgt_pad_mask.size(-1)
        causal = torch.ones(T, T, dtype=torch.bool, device=tgt_pad_mask.device).triu_(1)
        return tgt_pad_mask | causal.unsqueeze(0)

    def _forward(self, inputs, memory_bank, src_pad_mask, tgt_pad_mask, layer_cache=None, step=None):
        """(output [B,T,d], context-attention probabilities [B,H,T,L])."""
        p, training = self.drop.p, self.training
        x1 = _norm(self.layer_norm_1, inputs)
        sa, _ = self.self_attn(x1, x1, x1, mask=self._self_mask(tgt_pad_mask, step), layer_cache=layer_cache,
                               attn_type="self")
        h = ops.dropout(sa, p, training, residual=inputs)
        ca, probs = self.context_attn(memory_bank, memory_bank, _norm(self.layer_norm_2, h), mask=src_pad_mask,
                                      layer_cache=layer_cache, attn_type="context")
        return self.feed_forward(ops.dropout(ca, p, training, residual=h)), probs

    def forward(self, inputs, memory_bank, src_pad_mask, tgt_pad_mask, layer_cache=None, step=None,
                with_align=False):
        """(output, head 0 of the context attention [B,T,L], None) -- reference :75-108 with the
        alignment head always absent."""
        if with_align:
            raise NotImplementedError("guided alignment (with_align) is not part of the MSSRNet path")
        out, probs = self._forward(inputs, memory_bank, src_pad_mask, tgt_pad_mask, layer_cache=layer_cache,
                                   step=step)
        return out, probs[:, 0].contiguous(), None

    def update_dropout(self, dropout, attention_dropout):
        for attn in (self.self_attn, self.context_attn):
            attn.update_dropout(attention_dropout)
        self.feed_forward.update_dropout(dropout)
        self.drop.p = dropout


class TransformerDecoder(DecoderBase):
    def __init__(self, num_layers, d_model, heads, d_ff, copy_attn, self_attn_type, dropout, attention_dropout,
                 embeddings, max_relative_positions, aan_useffn, full_context_alignment, alignment_layer,
                 alignment_heads):
        super().__init__()
        self.embeddings = embeddings
        self.state = {}
        layers = []
        for _ in range(num_layers):
            layers.append(TransformerDecoderLayer(
                d_model, heads, d_ff, dropout, attention_dropout, self_attn_type=self_attn_type,
                max_relative_positions=max_relative_positions, aan_useffn=aan_useffn,
                full_context_alignment=full_context_alignment, alignment_heads=alignment_heads))
        self.transformer_layers = nn.ModuleList(layers)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)
        self._copy = copy_attn                  # recorded only (no copy generator on this path)
        self.alignment_layer = alignment_layer  # recorded only

    @classmethod
    def from_opt(cls, opt, embeddings):
        """Reads the same option names as the reference's from_opt (:249-266)."""
        return cls(num_layers=opt.dec_layers, d_model=opt.dec_rnn_size, heads=opt.heads, d_ff=opt.transformer_ff,
                   copy_attn=opt.copy_attn, self_attn_type=opt.self_attn_type, dropout=_first(opt.dropout),
                   attention_dropout=_first(opt.attention_dropout), embeddings=embeddings,
                   max_relative_positions=opt.max_relative_positions, aan_useffn=opt.aan_useffn,
                   full_context_alignment=opt.full_context_alignment, alignment_layer=opt.alignment_layer,
                   alignment_heads=opt.alignment_heads)

    # ---- decoding state: {"src": [Ls,B,1] ids, "cache": {"layer_i": {four tensors [B,H,len,dh]}} | None}
    def init_state(self, src, memory_bank, enc_hidden):
        self.state["src"] = src
        self.state["cache"] = None

    def _init_cache(self, memory_bank):
        self.state["cache"] = {"layer_%d" % i: dict.fromkeys(_CACHE_SLOTS)
                               for i in range(len(self.transformer_layers))}

    def map_state(self, fn):
        """Applies ``fn(tensor, batch_dim)`` to the source ids (batch on dim 1) and to every cached
        key / value tensor (batch on dim 0): how greedy search drops finished sentences
        (onmt/translate/translator_wh.py:744-746)."""
        self.state["src"] = fn(self.state["src"], 1)
        for layer_cache in (self.state["cache"] or {}).values():
            for slot, held in layer_cache.items():
                if held is not None:
                    layer_cache[slot] = fn(held, 0)

    def detach_state(self):
        self.state["src"] = self.state["src"].detach()

    def forward(self, tgt, memory_bank, step=None, memory_lengths=None, with_align=False, **unused):
        """tgt: Long [T,B,1] ids or float [T,B,V] probabilities (one row when ``step`` is given)."""
        if with_align:
            raise NotImplementedError("guided alignment (with_align) is not part of the MSSRNet path")
        if step == 0:
            self._init_cache(memory_bank)
        x = self.embeddings(tgt, step=step).transpose(0, 1).contiguous()          # [B,T,d]
        memory = memory_bank.transpose(0, 1).contiguous()                          # [B,Ls,d]
        # the source width comes from the ids given to init_state, as in the reference (:304)
        src_pad = padding_mask(memory_lengths, self.state["src"].shape[0]).unsqueeze(1)
        # column 0 of a SOFT input is the probability of word 0: comparing it with the padding index
        # reproduces the reference (:294,306), i.e. effectively no target padding mask on that path
        tgt_pad = tgt[:, :, 0].t().detach().eq(self.embeddings.word_padding_idx).unsqueeze(1)
        caches = self.state["cache"] if step is not None else None
        if step is None and len(self.transformer_layers) > 0:
            # the whole target sequence at once: all layers + final LayerNorm as one engine call per direction
            layer0 = self.transformer_layers[0]
            fused = run_stack(self, x, memory, layer0._self_mask(tgt_pad, None), src_pad[:, 0])
            if fused is not None:
                dec_outs, top = fused
                return dec_outs.transpose(0, 1).contiguous(), {"std": top.transpose(0, 1).contiguous()}
        top = None
        for i, layer in enumerate(self.transformer_layers):
            x, top, _ = layer(x, memory, src_pad, tgt_pad, step=step,
                              layer_cache=None if caches is None else caches["layer_%d" % i])
        dec_outs = _norm(self.layer_norm, x).transpose(0, 1).contiguous()
        return dec_outs, {"std": top.transpose(0, 1).contiguous()}

    def update_dropout(self, dropout, attention_dropout):
        self.embeddings.update_dropout(dropout)
        for layer in self.transformer_layers:
            layer.update_dropout(dropout, attention_dropout)
