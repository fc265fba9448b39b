"""CPU oracle (TEST INFRASTRUCTURE): the callers of the hot path -- soft step-wise decode,
WGAN-GP penalties and the three training phases -- restated from the reference trainer.

Follows onmt/trainer_wgan_wh.py (train_discriminator :505-629, train_generator :865-982,
train_reconstruct :729-863, init_launch :465-503, penalties :1217-1331),
onmt/train_and_translate.py (decode_batch_ori :202-231, fetch_dec_output_length :289-326),
onmt/trainer_teacher.py:229-267 and the greedy loop of
onmt/translate/translator_wh.py:665-756 with the README flag set (README.md:54-84).
Networks are dicts of tensors keyed like the reference state-dicts (see functional.py).
"""
import math

import torch

from . import functional as F_


class Net:
    """Parameters of one network + its train/eval mode + head count."""

    def __init__(self, state, heads, cfg, train=True):
        self.P = {}
        for k, v in state.items():
            t = v.detach().clone().float() if v.is_floating_point() else v.clone()
            if not (k.endswith(".pe.pe")):
                t.requires_grad_(True)
            self.P[k] = t
        self.heads = heads
        self.cfg = cfg
        self.training = train
        self.adam = None

    def named_params(self):
        return [(k, v) for k, v in self.P.items() if v.requires_grad]

    def params(self):
        return [v for _, v in self.named_params()]

    def ctx(self):
        return F_.Ctx(self.cfg.dropout, self.cfg.attention_dropout) if self.training else F_.EVAL

    def zero_grad(self):
        for p in self.params():
            p.grad = None

    def grads(self):
        return {k: (v.grad.detach().clone() if v.grad is not None else None) for k, v in self.named_params()}


def make_nets(states, cfg):
    return dict(
        model=Net(states["model"], cfg.heads, cfg),
        style_generator=Net(states["style_generator"], cfg.sgen[1], cfg),
        style_discriminator=Net(states["style_discriminator"], 1, cfg),
        text_discriminator=Net(states["text_discriminator"], cfg.tdisc[1], cfg),
        teacher=Net(states["teacher"], cfg.cls[1], cfg, train=False))


# ------------------------------------------------------------------ optimiser pieces
def clip_grad_norm(params, max_norm):
    """torch.nn.utils.clip_grad_norm_ (L2): scale by max_norm/(total+1e-6) when that is < 1."""
    gs = [p.grad for p in params if p.grad is not None]
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in gs))
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        for g in gs:
            g.mul_(coef)
    return total


def adam_step(net, lr, beta1, beta2, eps=1e-8):
    """torch.optim.Adam(lr, betas) without weight decay / amsgrad; params without grad are skipped."""
    if net.adam is None:
        net.adam = {}
    with torch.no_grad():
        for k, p in net.named_params():
            if p.grad is None:
                continue
            st = net.adam.setdefault(k, dict(t=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
            st["t"] += 1
            st["m"].mul_(beta1).add_(p.grad, alpha=1 - beta1)
            st["v"].mul_(beta2).addcmul_(p.grad, p.grad, value=1 - beta2)
            bc1 = 1 - beta1 ** st["t"]
            bc2 = 1 - beta2 ** st["t"]
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(st["m"], denom, value=-lr / bc1)


# ------------------------------------------------------------------ small helpers
def disc_gain(logits, cfg, mode):
    """_postprocess_discriminator_gain (:402-416)."""
    if mode == "normal":
        return logits
    if mode == "sigmoid":
        return torch.sigmoid(logits) * cfg.gain_ratio
    raise ValueError(mode)


def pick(probs, labels):
    """cal_disc_loss (:1192-1200): probs[i, labels[i]]."""
    idx = torch.as_tensor(labels, device=probs.device).view(-1, 1)
    return probs.gather(1, idx).squeeze(1)


def one_hot(ids, V):
    return torch.zeros(ids.numel(), V).scatter_(1, ids.reshape(-1, 1), 1.0).view(*ids.shape, V)


def unpack_batch(batch, cfg):
    """Trainer.unpack_batch (:432-454)."""
    src, src_lengths = batch.src, batch.src_lengths
    real_src = src[1:]
    real_len = src_lengths - 1
    lab0 = cfg.label_token_ids
    labels_list = [lab0.index(int(t)) for t in src[0, :, 0].tolist()]
    labels = torch.tensor(labels_list, dtype=torch.long)
    pad_row = torch.full((1, batch.tgt.size(1), 1), cfg.pad_idx, dtype=torch.long)
    vocab_target = torch.cat([batch.tgt[1:], pad_row])
    return labels, labels_list, real_src, real_len, vocab_target


def vocab_loss(log_probs, target, batch_size, cfg=None):
    """VocabLossFn(token_mode=False, all_tokens_aware=sent_stop_aware) (train_util.py:13-32):
    summed NLL / batch; every position counts (pads too) when sent_stop_aware, else pads are ignored."""
    nll = -log_probs.gather(1, target.view(-1, 1)).squeeze(1)
    if cfg is not None and not cfg.sent_stop_aware:
        nll = nll * target.ne(cfg.pad_idx).to(nll.dtype)
    return nll.sum() / batch_size


def nll_mean(logits, labels):
    return -torch.log_softmax(logits, -1).gather(1, labels.view(-1, 1)).mean()


def style_gap(real, gen, lengths):
    """cal_style_gap (:456-463): mean over valid tokens of the squared L2 gap."""
    gap = ((real - gen) ** 2).sum(2).transpose(0, 1)
    return gap[F_.sequence_mask(lengths, real.size(0))].mean()


def fetch_dec_output_length(vocab_probs, cfg):
    """train_and_translate.py:289-326 (host-side length rule; longest forced to seq_len)."""
    seq_len, B = vocab_probs.size(0), vocab_probs.size(1)
    ids = torch.argmax(vocab_probs, 2)
    rows = ids.transpose(0, 1).tolist()
    stops = [cfg.eos_idx] + ([cfg.dot_idx, cfg.excl_idx, cfg.qst_idx] if cfg.sent_stop_aware else [])
    lens, best, best_i = [], 0, 0
    for i, sent in enumerate(rows):
        cut = min([sent.index(s) + 1 for s in stops if s in sent] + [seq_len])
        if cut <= 1:
            cut = seq_len
        lens.append(cut)
        if cut > best:
            best, best_i = cut, i
    lens[best_i] = seq_len
    return lens, ids


# ------------------------------------------------------------------ model pieces
def gen_real_style(nets, real_src, real_len):
    t = nets["teacher"]
    with torch.no_grad():
        _, mem, _ = F_.encoder(F_.sub(t.P, "encoder."), t.heads, real_src, real_len, t.cfg.pad_idx, F_.EVAL)
    return mem.detach()


def gen_style(nets, desired, real_src, real_len):
    g = nets["style_generator"]
    return F_.style_generator(g.P, g.heads, desired, real_src, real_len, g.cfg.pad_idx, g.ctx())


def encode_with_style(nets, real_src, real_len, style_rep):
    m = nets["model"]
    enc_state, mem, lengths = F_.encoder(F_.sub(m.P, "encoder."), m.heads, real_src, real_len,
                                         m.cfg.pad_idx, m.ctx())
    return enc_state, F_.bond_layer(F_.sub(m.P, "bond_layer."), mem, style_rep), lengths


def self_reconstruct(nets, real_src, real_len, tgt, style_rep, temperature=1.0):
    """_self_reconstruct (:648-666): teacher-forced decoder on the FULL tgt, log-softmax."""
    m = nets["model"]
    _, mem, lengths = encode_with_style(nets, real_src, real_len, style_rep)
    st = F_.DecoderState(real_src)
    dec_out, _ = F_.decoder(F_.sub(m.P, "decoder."), m.heads, st, tgt, mem, lengths, m.cfg.pad_idx, ctx=m.ctx())
    return torch.log_softmax(F_.generator_logits(m.P, dec_out) / temperature, -1)


def dynamic_decoding(nets, real_src, real_len, style_rep, max_dec_steps):
    """_dynamic_decoding (:668-697) + decode_batch_ori: one-hot <s>, feed the softmax back."""
    m = nets["model"]
    cfg = m.cfg
    _, mem, _ = encode_with_style(nets, real_src, real_len, style_rep)
    st = F_.DecoderState(real_src)
    B = real_len.size(0)
    dec_in = one_hot(torch.full((B,), cfg.bos_idx, dtype=torch.long), cfg.vocab_size).unsqueeze(0)
    outs = []
    for step in range(max_dec_steps):
        dec_out, _ = F_.decoder(F_.sub(m.P, "decoder."), m.heads, st, dec_in, mem, real_len, cfg.pad_idx,
                                step=step, ctx=m.ctx())
        probs = torch.softmax(F_.generator_logits(m.P, dec_out) / cfg.dec_temperature, -1)
        outs.append(probs[-1:])
        dec_in = outs[-1]
    vocab_probs = torch.cat(outs, 0)
    lens, ids = fetch_dec_output_length(vocab_probs, cfg)
    return vocab_probs, torch.tensor(lens, dtype=torch.long), ids


def call_text_disc(nets, x, lengths):
    d = nets["text_discriminator"]
    logits = F_.style_classifier(d.P, d.heads, x, lengths, d.cfg.pad_idx, d.ctx())
    return disc_gain(logits, d.cfg, d.cfg.gain_mode_t)


def call_style_disc(nets, x, lengths):
    d = nets["style_discriminator"]
    return disc_gain(F_.style_discriminator(d.P, x, lengths), d.cfg, d.cfg.gain_mode_s)


def call_teacher(nets, x, lengths):
    t = nets["teacher"]
    return F_.style_classifier(t.P, t.heads, x, lengths, t.cfg.pad_idx, F_.EVAL)


# ------------------------------------------------------------------ gradient penalties
def penalize_style_grad(d_fn, real, fake, lengths, labels, k, lamb, gp_norm_seq=True):
    """penalize_gen_style_grad (:1217-1263); real/fake [B, L, dim]; alpha from the CPU RNG."""
    alpha = torch.rand(real.size(0), 1, 1).type_as(real)
    inter = (alpha * real + (1 - alpha) * fake).detach().requires_grad_(True)
    err = pick(d_fn(inter, lengths), labels)
    g, = torch.autograd.grad(err, inter, torch.ones_like(err), create_graph=True)
    if gp_norm_seq:
        g = g.contiguous().view(-1, g.size(-1))[F_.sequence_mask(lengths).contiguous().view(-1)]
    return ((g.norm(2, dim=1) - k) ** 2).mean() * lamb


def penalize_text_grad(d_fn, real, fake, lengths, labels, k, lamb, gp_norm_seq=True):
    """penalize_gen_text_grad (:1283-1331); real/fake [B, T, V], D takes [T, B, V]."""
    x_len = real.size(1)
    alpha = torch.rand(real.size(0), 1, 1).type_as(real)
    inter = (alpha * real + (1 - alpha) * fake).transpose(0, 1).detach().requires_grad_(True)
    err = pick(d_fn(inter, lengths), labels)
    g, = torch.autograd.grad(err, inter, torch.ones_like(err), create_graph=True)
    g = g.transpose(0, 1)
    if gp_norm_seq:
        g = g.contiguous().view(-1, g.size(-1))[F_.sequence_mask(lengths, x_len).contiguous().view(-1)]
    return ((g.norm(2, dim=1) - k) ** 2).mean() * lamb


# ------------------------------------------------------------------ phases
def _step(nets, names, cfg, skip=()):
    for n in names:
        clip_grad_norm(nets[n].params(), cfg.max_grad_norm)
    for n in names:
        if n not in skip:
            adam_step(nets[n], cfg.learning_rate, cfg.adam_beta1, cfg.adam_beta2)


def train_discriminator(nets, batch, cfg):
    nets["style_generator"].training = nets["model"].training = False
    nets["style_discriminator"].training = nets["text_discriminator"].training = True
    nets["style_discriminator"].zero_grad()
    nets["text_discriminator"].zero_grad()
    labels, labels_list, real_src, real_len, vocab_target = unpack_batch(batch, cfg)
    text_len = (real_len + 1).detach()
    log = {}
    # text critic: real, fake, penalty
    g_real = call_text_disc(nets, vocab_target, text_len)
    g_real = pick(g_real, labels_list).mean()
    g_real.backward(torch.tensor(-1.0))
    with torch.no_grad():
        rep = gen_style(nets, labels, real_src, real_len)
        probs, _, _ = dynamic_decoding(nets, real_src, real_len, rep, vocab_target.size(0))
    g_fake = pick(call_text_disc(nets, probs, text_len), labels_list).mean()
    g_fake.backward(torch.tensor(1.0))
    with torch.no_grad():
        probs_t = probs.transpose(0, 1)
        real_onehot = one_hot(vocab_target.transpose(0, 1).squeeze(2), cfg.vocab_size)
    gp = penalize_text_grad(lambda x, l: call_text_disc(nets, x, l), real_onehot, probs_t, text_len,
                            labels_list, cfg.wgan_text_k, cfg.wgan_text_lamb, cfg.use_gp_norm_seq)
    gp.backward()
    log.update(TR_gain=g_real.item(), TF_gain=g_fake.item(), T_gp=gp.item(),
               T_cost=g_fake.item() - g_real.item() + gp.item(), T_Wgap=g_real.item() - g_fake.item())
    # style critic
    real_rep_t = gen_real_style(nets, real_src, real_len).transpose(0, 1)
    s_real = pick(call_style_disc(nets, real_rep_t, real_len), labels_list).mean()
    s_real.backward(torch.tensor(-1.0))
    with torch.no_grad():
        gen_rep_t = gen_style(nets, labels, real_src, real_len).transpose(0, 1)
    s_fake = pick(call_style_disc(nets, gen_rep_t, real_len), labels_list).mean()
    s_fake.backward(torch.tensor(1.0))
    sgp = penalize_style_grad(lambda x, l: call_style_disc(nets, x, l), real_rep_t, gen_rep_t, real_len,
                              labels_list, cfg.wgan_style_k, cfg.wgan_style_lamb, cfg.use_gp_norm_seq)
    sgp.backward()
    log.update(SR_gain=s_real.item(), SF_gain=s_fake.item(), S_gp=sgp.item(),
               S_cost=s_fake.item() - s_real.item() + sgp.item(), S_Wgap=s_real.item() - s_fake.item())
    _step(nets, ("style_discriminator", "text_discriminator"), cfg)
    return log


def train_generator(nets, batch, cfg):
    nets["style_generator"].training = nets["model"].training = True
    nets["style_discriminator"].training = nets["text_discriminator"].training = False
    nets["style_generator"].zero_grad()
    nets["model"].zero_grad()
    labels, labels_list, real_src, real_len, vocab_target = unpack_batch(batch, cfg)
    tgt_len, B = vocab_target.size(0), vocab_target.size(1)
    desired = 1 - labels                                   # num_styles == 2 (tat:72-73)
    desired_list = desired.tolist()
    max_dec_steps = tgt_len + cfg.max_extra_dec_steps
    log = {}
    rep = gen_style(nets, desired, real_src, real_len)
    probs, seq_lens, _ = dynamic_decoding(nets, real_src, real_len, rep, max_dec_steps)
    g_t = pick(call_text_disc(nets, probs, seq_lens), desired_list).mean()
    g_t.backward(torch.tensor(-1.0 * cfg.text_ap_scale))
    with torch.no_grad():
        acc = call_teacher(nets, probs, seq_lens).argmax(-1).eq(desired).float().mean().item()
    rep2_t = gen_style(nets, desired, real_src, real_len).transpose(0, 1)
    g_s = pick(call_style_disc(nets, rep2_t, real_len), desired_list).mean()
    g_s.backward(torch.tensor(-1.0 * cfg.style_ap_scale))
    log.update(AT_gain=g_t.item(), AT_acc=acc, AS_gain=g_s.item())
    if cfg.ap_gen_self_style_text:
        rep3 = gen_style(nets, labels, real_src, real_len)
        sprobs, _, _ = dynamic_decoding(nets, real_src, real_len, rep3, tgt_len)
        vl = vocab_loss(sprobs.log().view(-1, sprobs.size(-1)), vocab_target.contiguous().view(-1), B, cfg)
        vl.backward()
        log.update(T_selfstyle_vloss=vl.item())
    _step(nets, ("style_generator", "model"), cfg)
    return log


def train_reconstruct(nets, batch, cfg):
    nets["style_generator"].training = nets["model"].training = True
    nets["style_generator"].zero_grad()
    nets["model"].zero_grad()
    labels, labels_list, real_src, real_len, vocab_target = unpack_batch(batch, cfg)
    tgt_len, B = vocab_target.size(0), vocab_target.size(1)
    flat_target = vocab_target.contiguous().view(-1)
    desired = 1 - labels
    max_dec_steps = tgt_len + cfg.max_extra_dec_steps
    teacher = nets["teacher"]
    log = {}
    real_rep = gen_real_style(nets, real_src, real_len)
    lp = self_reconstruct(nets, real_src, real_len, batch.tgt, real_rep)        # sr_noise_prob == 0
    l_noised = vocab_loss(lp.view(-1, lp.size(-1)), flat_target, B, cfg)
    total = cfg.sr_loss_ratio * l_noised
    rep_src = gen_style(nets, labels, real_src, real_len)
    rep_tgt = gen_style(nets, desired, real_src, real_len)
    probs, _, _ = dynamic_decoding(nets, real_src, real_len, rep_src, tgt_len)
    l_rec = vocab_loss(probs.log().view(-1, probs.size(-1)), flat_target, B, cfg)
    total = total + cfg.sr_loss_ratio * l_rec
    log.update(T_CONSTRUCT_R_VL=l_noised.item(), T_CONSTRUCT_F_VL=l_rec.item())
    if cfg.use_cls_feedback:
        s_src = F_.classifier_do_cls(teacher.P, rep_src.transpose(0, 1), real_len)
        l_s_src = nll_mean(s_src, labels)
        s_tgt = F_.classifier_do_cls(teacher.P, rep_tgt.transpose(0, 1), real_len)
        l_s_tgt = nll_mean(s_tgt, desired)
        total = total + cfg.cls_loss_ratio * l_s_src + cfg.cls_loss_ratio * l_s_tgt
    if not cfg.without_stl_penalty:
        l_gap = style_gap(real_rep, rep_src, real_len)
        total = total + cfg.alpha_style_gap * l_gap
    if cfg.use_cls_feedback:
        cprobs, clens, _ = dynamic_decoding(nets, real_src, real_len, rep_tgt, max_dec_steps)
        t_logits = call_teacher(nets, cprobs, clens)
        l_t = nll_mean(t_logits, desired)
        total = total + cfg.cls_loss_ratio * l_t
        log.update(T_CLS_loss=l_t.item(), T_CLS_acc=t_logits.argmax(-1).eq(desired).float().mean().item(),
                   S_CLS_loss=l_s_tgt.item(), S_CLS_acc=s_tgt.argmax(-1).eq(desired).float().mean().item(),
                   SCur_CLS_loss=l_s_src.item(), SCur_CLS_acc=s_src.argmax(-1).eq(labels).float().mean().item())
    if not cfg.without_stl_penalty:
        log.update(S_GAP_loss=l_gap.item())
    total.backward()
    # note: no gradient clipping in this phase (:859-863)
    if cfg.use_cls_feedback or not cfg.without_stl_penalty:
        adam_step(nets["style_generator"], cfg.learning_rate, cfg.adam_beta1, cfg.adam_beta2)
    adam_step(nets["model"], cfg.learning_rate, cfg.adam_beta1, cfg.adam_beta2)
    return log


def init_launch(nets, batch, cfg):
    nets["style_generator"].training = nets["model"].training = True
    nets["style_generator"].zero_grad()
    nets["model"].zero_grad()
    labels, labels_list, real_src, real_len, vocab_target = unpack_batch(batch, cfg)
    B = len(labels_list)
    real_rep = gen_real_style(nets, real_src, real_len)
    l_style = style_gap(real_rep, gen_style(nets, labels, real_src, real_len), real_len)
    l_style.backward()
    lp = self_reconstruct(nets, real_src, real_len, batch.tgt, real_rep)
    l_v = vocab_loss(lp.view(-1, lp.size(-1)), vocab_target.contiguous().view(-1), B, cfg)
    l_v.backward()
    _step(nets, ("style_generator", "model"), cfg)
    return dict(INIT_SL=l_style.item(), INIT_VL=l_v.item())


def teacher_step_loss(nets, batch, cfg, net="teacher"):
    """trainer_teacher.py:238-252: ids / ids+</s> / full padded length, NLL summed."""
    labels, _, real_src, real_len, vocab_target = unpack_batch(batch, cfg)
    t = nets[net]
    f = lambda x, l: F_.style_classifier(t.P, t.heads, x, l, cfg.pad_idx, t.ctx())  # noqa: E731
    lg = f(real_src, real_len)
    loss = nll_mean(lg, labels) + nll_mean(f(vocab_target, real_len + 1), labels)
    loss = loss + nll_mean(f(vocab_target, torch.ones_like(real_len) * vocab_target.size(0)), labels)
    return loss, lg


def greedy_decode(nets, src, lens, desired, cfg, max_length):
    """translator_wh.py:665-756 with GreedySearch (greedy_search.py:135-190), beam 1:
    topk(1) per step, finished sentences leave the batch (map_state index_select)."""
    m, g = nets["model"], nets["style_generator"]
    with torch.no_grad():
        rep = F_.style_generator(g.P, g.heads, desired, src, lens, cfg.pad_idx, F_.EVAL)
        _, mem, _ = F_.encoder(F_.sub(m.P, "encoder."), m.heads, src, lens, cfg.pad_idx, F_.EVAL)
        mem = F_.bond_layer(F_.sub(m.P, "bond_layer."), mem, rep)
        st = F_.DecoderState(src)
        B = src.size(1)
        alive = torch.full((B, 1), cfg.bos_idx, dtype=torch.long)
        orig = torch.arange(B)
        preds = [None] * B
        mlen = lens
        for step in range(max_length):
            dec_in = alive[:, -1].view(1, -1, 1)
            dec_out, _ = F_.decoder(F_.sub(m.P, "decoder."), m.heads, st, dec_in, mem, mlen, cfg.pad_idx,
                                    step=step, ctx=F_.EVAL)
            logp = torch.log_softmax(F_.generator_logits(m.P, dec_out.squeeze(0)), -1)
            top = logp.argmax(-1, keepdim=True)
            alive = torch.cat([alive, top], -1)
            fin = top.view(-1).eq(cfg.eos_idx)
            if step + 1 == max_length:
                fin = torch.ones_like(fin)
            if fin.any():
                for b in fin.nonzero().view(-1).tolist():
                    preds[int(orig[b])] = alive[b, 1:].tolist()
                if fin.all():
                    break
                keep = (~fin).nonzero().view(-1)
                alive, orig = alive[keep], orig[keep]
                mem = mem.index_select(1, keep)
                mlen = mlen.index_select(0, keep)
                st.map(lambda t, dim: t.index_select(dim, keep))
    return preds
