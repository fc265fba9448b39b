"""Callers of the encoder path: classifier / critics / bond layer and the model builders.

Counterparts of onmt/style_generator_builder.py (StyleClassifier :148-193, ContextReps2
:229-256, StyleDiscriminator :259-286, BondLayer :525-534, builders :289-632),
onmt/style_generator.py:15-49 and onmt/model_builder.py:117-237, with the same attribute /
state-dict names (``encoder.*``, ``sent_fn.w_fn.{0,2}.*``, ``sent_out_layer.*``,
``sent_middle_fn.*``, ``style_cls.weight``, ``w.*``, ``layer_norm.*``, ``generator.0.*``,
``bond_layer.*``).  All arithmetic goes through ``ops`` (libmssr_hip.so) and is twice
differentiable, as the WGAN-GP penalties require.
"""
import torch
import torch.nn as nn
from torch.nn.init import xavier_uniform_

from . import ops
from .decoders.transformer import TransformerDecoder
from .encoder_engine import run_stack
from .encoders.transformer import TransformerEncoder
from .modules.embeddings import Embeddings
from .modules.util_class import Cast
from .utils.misc import padding_mask


class ContextReps2(nn.Module):
    """Masked softmax pooling over time: softmax_t(w2 relu(w1 x)) weighted sum (batch first)."""

    def __init__(self, in_size, hidden_size=None):
        super().__init__()
        if hidden_size is None:
            hidden_size = int(in_size / 2) + 1
        self.w_fn = nn.Sequential(nn.Linear(in_size, hidden_size), nn.ReLU(), nn.Linear(hidden_size, 1))

    def forward(self, x_states, src_lengths):
        w0, w2 = self.w_fn[0], self.w_fn[2]
        if not x_states.is_contiguous() and x_states.transpose(0, 1).is_contiguous():
            # an encoder's [L, B, d] memory bank viewed batch first: the GEMM and the pooling read it where it lies
            h = ops.linear(x_states.transpose(0, 1), w0.weight, w0.bias, relu=True).transpose(0, 1)
        else:
            h = ops.linear(x_states, w0.weight, w0.bias, relu=True)
        # scores w2 . h + b2, -inf past each length, softmax over time, weighted sum: one launch (csrc/pool.hip)
        return ops.ctx_pool(x_states, h, w2.weight, w2.bias, src_lengths)


class StyleClassifier(nn.Module):
    """Teacher classifier and text discriminator: encoder -> pooling -> Linear."""

    def __init__(self, encoder, num_styles, opt):
        super().__init__()
        self.encoder = encoder
        model_dim = opt.enc_rnn_size
        self.sent_fn = ContextReps2(model_dim)
        self.sent_out_layer = nn.Linear(model_dim, num_styles)

    def do_cls(self, style_representation, src_lengths):
        sent_vec = self.sent_fn(style_representation, src_lengths)
        return ops.linear(sent_vec, self.sent_out_layer.weight, self.sent_out_layer.bias).float()

    def forward(self, src, src_lengths):
        _, memory_bank, _ = self.encoder(src, src_lengths)
        return self.do_cls(memory_bank.transpose(0, 1), src_lengths)

    def forward_embedded(self, emb, src_lengths):
        """forward() on inputs that are embedded already (several texts batched behind the embedding layer)."""
        _, memory_bank, _ = self.encoder.forward_embedded(emb, src_lengths)
        return self.do_cls(memory_bank.transpose(0, 1), src_lengths)


class StyleDiscriminator(nn.Module):
    def __init__(self, num_styles, opt):
        super().__init__()
        style_dim = opt.n_heads_sgenerator * opt.dim_each_head_sgenerator
        hidden_size = opt.n_heads_sdiscr * opt.dim_each_head_sdiscr
        self.sent_fn = ContextReps2(style_dim, hidden_size=hidden_size)
        assert opt.n_layers_sdiscr >= 1
        layers = []
        for i in range(opt.n_layers_sdiscr):
            layers.append(nn.Linear(hidden_size if i != 0 else style_dim, hidden_size))
            layers.append(nn.ReLU())
        self.sent_middle_fn = nn.Sequential(*layers)
        self.sent_out_layer = nn.Linear(hidden_size, num_styles)

    def forward(self, x_states, src_lengths):
        h = self.sent_fn(x_states, src_lengths)
        for m in self.sent_middle_fn:
            if isinstance(m, nn.Linear):
                h = ops.linear(h, m.weight, m.bias, relu=True)
        return ops.linear(h, self.sent_out_layer.weight, self.sent_out_layer.bias).float()


class BondLayer(nn.Module):
    """memory_bank + LN(W style)."""

    def __init__(self, h_size, model_dim):
        super().__init__()
        self.w = nn.Linear(h_size, model_dim)
        self.layer_norm = nn.LayerNorm(model_dim, eps=1e-6)

    def forward(self, memory_bank, style_represents):
        sf = ops.linear(style_represents, self.w.weight, self.w.bias)
        ln = self.layer_norm
        return ops.add(memory_bank, ops.layer_norm(sf, ln.weight, ln.bias, ln.eps))


class StyleGenerator2(nn.Module):
    """desired style + X -> per-token style representation [L, B, d] (onmt/style_generator.py)."""

    def __init__(self, encoder, h_size, num_styles):
        super().__init__()
        assert isinstance(encoder, TransformerEncoder)
        self.style_cls = nn.Embedding(num_styles, h_size)
        self.encoder = encoder

    def forward(self, desired_style, src, src_lengths, eval_rows=0):
        """eval_rows (no reference counterpart): the first eval_rows sentences run without dropout in train mode."""
        enc = self.encoder
        enc._check_args(src, src_lengths)
        emb = enc.embeddings(src, eval_rows=eval_rows) if eval_rows else enc.embeddings(src)
        # the style token row is neither scaled by sqrt(d) nor position-encoded (quirk 10)
        cls_emb = self.style_cls(desired_style).to(emb.dtype)
        x = torch.cat((cls_emb.unsqueeze(0), emb), 0).transpose(0, 1)
        pad = padding_mask(src_lengths + 1, src.size(0) + 1)
        out = run_stack(enc, x, pad, eval_rows)
        if out is None and eval_rows:
            raise RuntimeError("StyleGenerator2: eval_rows needs the encoder engine (not available for this graph / shape)")
        if out is None:
            out = x.contiguous()
            mask = pad.unsqueeze(1)
            for layer in enc.transformer:
                out = layer(out, mask)
            ln = enc.layer_norm
            out = ops.layer_norm(out, ln.weight, ln.bias, ln.eps)
        return out.transpose(0, 1).contiguous()[1:]


class NMTModel(nn.Module):
    """encoder + decoder (+ ``generator`` and ``bond_layer`` attached by the builder), as
    onmt/models/model.py:4-56; the WGAN trainer calls the parts directly."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def forward(self, src, tgt, lengths, bptt=False, with_align=False):
        dec_in = tgt[:-1]
        enc_state, memory_bank, lengths = self.encoder(src, lengths)
        if bptt is False:
            self.decoder.init_state(src, memory_bank, enc_state)
        return self.decoder(dec_in, memory_bank, memory_lengths=lengths, with_align=with_align)

    def update_dropout(self, dropout):
        self.encoder.update_dropout(dropout, dropout)
        self.decoder.update_dropout(dropout, dropout)


def _glorot(module):
    for p in module.parameters():
        if p.dim() > 1:
            xavier_uniform_(p)


def _embeddings(cfg, dim):
    return Embeddings(dim, cfg.vocab_size, cfg.pad_idx, position_encoding=True, dropout=cfg.dropout)


def _encoder(cfg, spec):
    layers, heads, dh = spec
    d = heads * dh
    return TransformerEncoder(layers, d, heads, 4 * d, cfg.dropout, cfg.attention_dropout, _embeddings(cfg, d), 0)


def build_main_model(cfg):
    """model_builder.build_base_model (:117-237) + BondLayer (train_single_wgan_wh.py:116)."""
    d = cfg.d_model
    enc = TransformerEncoder(cfg.layers, d, cfg.heads, cfg.d_ff, cfg.dropout, cfg.attention_dropout,
                             _embeddings(cfg, d), 0)
    dec = TransformerDecoder(cfg.layers, d, cfg.heads, cfg.d_ff, False, "scaled-dot", cfg.dropout,
                             cfg.attention_dropout, _embeddings(cfg, d), 0, False, False, -3, 0)
    model = NMTModel(enc, dec)
    generator = nn.Sequential(nn.Linear(d, cfg.vocab_size), Cast(torch.float32), nn.LogSoftmax(dim=-1))
    _glorot(model)
    _glorot(generator)
    model.generator = generator
    model.bond_layer = BondLayer(cfg.d_sgen, d)
    return model


class _Opt:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def build_classifier(cfg, spec=None, num_out=None):
    """Teacher classifier (spec=cfg.cls) or text discriminator (spec=cfg.tdisc, 2*num_styles logits)."""
    spec = spec or cfg.cls
    model = StyleClassifier(_encoder(cfg, spec), num_out or cfg.num_styles, _Opt(enc_rnn_size=spec[1] * spec[2]))
    _glorot(model)
    return model


def build_text_discriminator(cfg):
    return build_classifier(cfg, cfg.tdisc, cfg.n_disc_out)


def build_style_generator(cfg):
    model = StyleGenerator2(_encoder(cfg, cfg.sgen), cfg.d_sgen, cfg.num_styles)
    _glorot(model)
    return model


def build_style_discriminator(cfg):
    opt = _Opt(n_heads_sgenerator=cfg.sgen[1], dim_each_head_sgenerator=cfg.sgen[2], n_heads_sdiscr=cfg.sdisc[1],
               dim_each_head_sdiscr=cfg.sdisc[2], n_layers_sdiscr=cfg.sdisc[0])
    return StyleDiscriminator(cfg.n_disc_out, opt)        # torch default init, as the reference


def build_all(cfg, device="cuda", seed=None):
    """The five networks of a WGAN run (the teacher is a frozen seeded-random stand-in for the
    pre-trained checkpoint the reference loads through ClsProxy)."""
    if seed is not None:
        torch.manual_seed(seed)
    nets = dict(model=build_main_model(cfg), style_generator=build_style_generator(cfg),
                style_discriminator=build_style_discriminator(cfg),
                text_discriminator=build_text_discriminator(cfg), teacher=build_classifier(cfg))
    for m in nets.values():
        m.to(device)
    nets["teacher"].eval()
    for p in nets["teacher"].parameters():
        p.requires_grad_(False)
    return nets
