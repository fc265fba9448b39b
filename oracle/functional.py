"""CPU oracle (TEST INFRASTRUCTURE, not product code): a restatement of the reference's
transformer hot path as pure functions over a state-dict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  Each function names the reference lines it follows.  Parameters are looked up
by the reference's state-dict keys, e.g.
``P[prefix + "self_attn.linear_keys.weight"]``, so the same dict of tensors can be filled
from a reference module, from a golden fixture or from the product's modules.
Everything is plain torch fp32 on the CPU; pinned against tests/golden/*.pt, which were
produced by the reference itself (tests/golden/make_golden.py).
"""
import math

import torch

LN_EPS = 1e-6          # nn.LayerNorm(d, eps=1e-6) at every site (quirk 3)
MASK_FILL = -1e18      # onmt/modules/multi_headed_attn.py:199


class Ctx:
    """Dropout rates in force for one forward (0 when the owning module is in eval())."""

    def __init__(self, dropout=0.0, attention_dropout=0.0):
        self.p = float(dropout)
        self.ap = float(attention_dropout)

    @staticmethod
    def drop(x, p):
        if p <= 0.0:
            return x
        keep = (torch.rand_like(x) >= p).to(x.dtype)
        return x * keep / (1.0 - p)


EVAL = Ctx(0.0, 0.0)


def sub(P, prefix):
    """View of the parameters under ``prefix`` (keeps the same tensor objects)."""
    return {k[len(prefix):]: v for k, v in P.items() if k.startswith(prefix)}


def n_layers(P, stem):
    idx = {int(k[len(stem):].split(".")[0]) for k in P if k.startswith(stem)}
    return max(idx) + 1 if idx else 0


def sequence_mask(lengths, max_len=None):
    """onmt/utils/misc.py:45-56 -> Bool[B, max_len], True on valid positions."""
    max_len = int(max_len if max_len is not None else lengths.max())
    return torch.arange(max_len, device=lengths.device).unsqueeze(0) < lengths.unsqueeze(1)


def layer_norm(x, gamma, beta):
    """Row-wise (x-mean)/sqrt(biased var + 1e-6) * gamma + beta."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + LN_EPS) * gamma + beta


def linear(x, P, name):
    y = x.matmul(P[name + ".weight"].t())
    b = P.get(name + ".bias")
    return y if b is None else y + b


def multi_head_attention(P, heads, key, value, query, mask=None, layer_cache=None, attn_type=None,
                         ctx=EVAL):
    """onmt/modules/multi_headed_attn.py:77-227 (max_relative_positions == 0).

    key/value/query are batch-first [B, len, d]; ``mask`` is Bool [B,1,Lk] or [B,Lq,Lk],
    True = masked.  Returns (output [B,Lq,d], attn [B,H,Lq,Lk] = PRE-dropout softmax).
    """
    B = key.size(0)
    d = P["linear_query.weight"].size(0)
    dh = d // heads

    def split(x):                      # [B, len, d] -> [B, H, len, dh]   (:121-124)
        return x.view(B, -1, heads, dh).transpose(1, 2)

    if layer_cache is not None and attn_type == "self":          # :133-148
        q = linear(query, P, "linear_query")
        k = split(linear(query, P, "linear_keys"))
        v = split(linear(query, P, "linear_values"))
        if layer_cache["self_keys"] is not None:
            k = torch.cat((layer_cache["self_keys"], k), dim=2)
        if layer_cache["self_values"] is not None:
            v = torch.cat((layer_cache["self_values"], v), dim=2)
        layer_cache["self_keys"], layer_cache["self_values"] = k, v
    elif layer_cache is not None and attn_type == "context":     # :149-160
        q = linear(query, P, "linear_query")
        if layer_cache["memory_keys"] is None:
            k = split(linear(key, P, "linear_keys"))
            v = split(linear(value, P, "linear_values"))
        else:
            k, v = layer_cache["memory_keys"], layer_cache["memory_values"]
        layer_cache["memory_keys"], layer_cache["memory_values"] = k, v
    else:                                                        # :161-166
        k = split(linear(key, P, "linear_keys"))
        v = split(linear(value, P, "linear_values"))
        q = linear(query, P, "linear_query")
    q = split(q) / math.sqrt(dh)                                 # scaled BEFORE the product (:187)
    scores = torch.matmul(q, k.transpose(2, 3)).float()          # :189,195
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1), MASK_FILL)   # finite fill (:197-199)
    attn = torch.softmax(scores, dim=-1).to(q.dtype)             # :202
    ctxv = torch.matmul(Ctx.drop(attn, ctx.ap), v)               # :203-205
    ctxv = ctxv.transpose(1, 2).contiguous().view(B, -1, d)      # :126-129
    return linear(ctxv, P, "final_linear"), attn


def feed_forward(P, x, ctx=EVAL):
    """onmt/modules/position_ffn.py:35-37: x + D2(W2 D1(relu(W1 LN(x))))."""
    h = layer_norm(x, P["layer_norm.weight"], P["layer_norm.bias"])
    h = Ctx.drop(torch.relu(linear(h, P, "w_1")), ctx.p)
    return Ctx.drop(linear(h, P, "w_2"), ctx.p) + x


def encoder_layer(P, heads, x, mask, ctx=EVAL):
    """onmt/encoders/transformer.py:48-52 (pre-norm)."""
    n = layer_norm(x, P["layer_norm.weight"], P["layer_norm.bias"])
    a, _ = multi_head_attention(sub(P, "self_attn."), heads, n, n, n, mask=mask, attn_type="self", ctx=ctx)
    return feed_forward(sub(P, "feed_forward."), Ctx.drop(a, ctx.p) + x, ctx)


def sinusoid_table(max_len, dim):
    """onmt/modules/embeddings.py:26-32 -> [max_len, dim]."""
    pe = torch.zeros(max_len, dim)
    pos = torch.arange(0, max_len).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def embeddings(P, source, pad_idx, step=None, ctx=EVAL):
    """onmt/modules/embeddings.py:260-288 + PositionalEncoding.forward (:38-54).

    ``source`` is Long [len,B,1] (lookup; the padding row receives no gradient, as
    nn.Embedding(padding_idx=) does) or float [len,B,V] (the fork's soft path: a product
    with the FULL table, padding row included -- quirk 5).  ``P`` holds
    ``make_embedding.emb_luts.0.weight`` and the buffer ``make_embedding.pe.pe``.
    """
    W = P["make_embedding.emb_luts.0.weight"]
    if source.dim() > 2 and source.size(-1) > 1:
        emb = torch.matmul(source, W)
    else:
        ids = source[:, :, 0]
        rows = torch.arange(W.size(0), device=W.device).unsqueeze(1)
        Wl = torch.where(rows == pad_idx, W.detach(), W)
        emb = Wl[ids]
    dim = W.size(1)
    pe = P["make_embedding.pe.pe"]                       # [5000, 1, dim]
    emb = emb * math.sqrt(dim)
    emb = emb + (pe[:emb.size(0)] if step is None else pe[step])
    return Ctx.drop(emb, ctx.p)


def encoder(P, heads, src, lengths, pad_idx, ctx=EVAL):
    """onmt/encoders/transformer.py:117-136 -> (emb [L,B,d], memory_bank [L,B,d], lengths)."""
    emb = embeddings(sub(P, "embeddings."), src, pad_idx, ctx=ctx)
    out = emb.transpose(0, 1).contiguous()
    mask = ~sequence_mask(lengths, src.size(0)).unsqueeze(1)
    for i in range(n_layers(P, "transformer.")):
        out = encoder_layer(sub(P, "transformer.%d." % i), heads, out, mask, ctx)
    out = layer_norm(out, P["layer_norm.weight"], P["layer_norm.bias"])
    return emb, out.transpose(0, 1).contiguous(), lengths


def decoder_layer(P, heads, x, memory_bank, src_pad_mask, tgt_pad_mask, layer_cache=None, step=None,
                  ctx=EVAL):
    """onmt/decoders/transformer.py:110-171; returns (output, head-0 context attention) (:93-94)."""
    dec_mask = None
    if step is None:
        T = tgt_pad_mask.size(-1)
        future = torch.ones(T, T, dtype=torch.bool, device=x.device).triu(1).unsqueeze(0)
        dec_mask = tgt_pad_mask | future
    n1 = layer_norm(x, P["layer_norm_1.weight"], P["layer_norm_1.bias"])
    a, _ = multi_head_attention(sub(P, "self_attn."), heads, n1, n1, n1, mask=dec_mask,
                                layer_cache=layer_cache, attn_type="self", ctx=ctx)
    query = Ctx.drop(a, ctx.p) + x
    n2 = layer_norm(query, P["layer_norm_2.weight"], P["layer_norm_2.bias"])
    mid, attns = multi_head_attention(sub(P, "context_attn."), heads, memory_bank, memory_bank, n2,
                                      mask=src_pad_mask, layer_cache=layer_cache, attn_type="context",
                                      ctx=ctx)
    out = feed_forward(sub(P, "feed_forward."), Ctx.drop(mid, ctx.p) + query, ctx)
    return out, attns[:, 0, :, :].contiguous()


class DecoderState:
    """TransformerDecoder.state (onmt/decoders/transformer.py:268-284,339-352)."""

    def __init__(self, src):
        self.src = src
        self.cache = None

    def init_cache(self, n):
        self.cache = [{"memory_keys": None, "memory_values": None, "self_keys": None, "self_values": None}
                      for _ in range(n)]

    def map(self, fn):
        self.src = fn(self.src, 1)
        if self.cache is not None:
            for c in self.cache:
                for k, v in c.items():
                    if v is not None:
                        c[k] = fn(v, 0)


def decoder(P, heads, state, tgt, memory_bank, memory_lengths, pad_idx, step=None, ctx=EVAL):
    """onmt/decoders/transformer.py:289-337 -> (dec_outs [T,B,d], attn_std [T,B,L])."""
    L = n_layers(P, "transformer_layers.")
    if step == 0:
        state.init_cache(L)
    tgt_words = tgt[:, :, 0].transpose(0, 1)
    emb = embeddings(sub(P, "embeddings."), tgt, pad_idx, step=step, ctx=ctx)
    out = emb.transpose(0, 1).contiguous()
    mem = memory_bank.transpose(0, 1).contiguous()
    src_pad_mask = ~sequence_mask(memory_lengths, state.src.shape[0]).unsqueeze(1)
    tgt_pad_mask = tgt_words.detach().eq(pad_idx).unsqueeze(1)        # quirk 4 for float tgt
    attn = None
    for i in range(L):
        cache = state.cache[i] if step is not None else None
        out, attn = decoder_layer(sub(P, "transformer_layers.%d." % i), heads, out, mem, src_pad_mask,
                                  tgt_pad_mask, layer_cache=cache, step=step, ctx=ctx)
    out = layer_norm(out, P["layer_norm.weight"], P["layer_norm.bias"])
    return out.transpose(0, 1).contiguous(), attn.transpose(0, 1).contiguous()


# ------------------------------------------------------------------ callers of the path
def style_generator(P, heads, desired_style, src, lengths, pad_idx, ctx=EVAL):
    """onmt/style_generator.py:26-49: style token prepended un-scaled and without PE (quirk 10)."""
    E = sub(P, "encoder.")
    emb = embeddings(sub(E, "embeddings."), src, pad_idx, ctx=ctx)
    cls = P["style_cls.weight"][desired_style]
    out = torch.cat((cls.unsqueeze(0), emb), 0).transpose(0, 1).contiguous()
    mask = ~sequence_mask(lengths + 1, src.size(0) + 1).unsqueeze(1)
    for i in range(n_layers(E, "transformer.")):
        out = encoder_layer(sub(E, "transformer.%d." % i), heads, out, mask, ctx)
    out = layer_norm(out, E["layer_norm.weight"], E["layer_norm.bias"])
    return out.transpose(0, 1).contiguous()[1:]


def context_reps2(P, x, lengths):
    """onmt/style_generator_builder.py:243-256: masked softmax pooling over time (batch first)."""
    w = linear(torch.relu(linear(x, P, "w_fn.0")), P, "w_fn.2")            # [B, L, 1]
    mask = ~sequence_mask(lengths, w.size(1)).unsqueeze(2)
    w = w.masked_fill(mask, float("-inf"))
    return torch.sum(torch.softmax(w, 1) * x, 1)


def classifier_do_cls(P, style_representation, lengths):
    """StyleClassifier.do_cls (sgb:176-183)."""
    return linear(context_reps2(sub(P, "sent_fn."), style_representation, lengths), P, "sent_out_layer")


def style_classifier(P, heads, src, lengths, pad_idx, ctx=EVAL):
    """StyleClassifier.forward (sgb:185-193); teacher and text discriminator."""
    _, mem, _ = encoder(sub(P, "encoder."), heads, src, lengths, pad_idx, ctx)
    return classifier_do_cls(P, mem.transpose(0, 1), lengths)


def style_discriminator(P, x, lengths):
    """StyleDiscriminator.forward (sgb:275-286): pooling + [Linear, ReLU]*n + Linear."""
    h = context_reps2(sub(P, "sent_fn."), x, lengths)
    i = 0
    while "sent_middle_fn.%d.weight" % i in P:
        h = torch.relu(linear(h, P, "sent_middle_fn.%d" % i))
        i += 2
    return linear(h, P, "sent_out_layer")


def bond_layer(P, memory_bank, style_represents):
    """BondLayer.forward (sgb:531-534)."""
    sf = linear(style_represents, P, "w")
    return memory_bank + layer_norm(sf, P["layer_norm.weight"], P["layer_norm.bias"])


def generator_logits(P, dec_out):
    """generator[0] (Linear d->V) and the fp32 cast (model_builder.py:180-185)."""
    return linear(dec_out, P, "generator.0").float()
