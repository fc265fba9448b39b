"""The WGAN-GP training phases of MSSRNet driven over the gfx950 modules.

Counterpart of onmt/trainer_wgan_wh.py (``train_discriminator`` :505-629, ``train_generator``
:865-982, ``train_reconstruct`` :729-863, ``init_launch`` :465-503, penalties :1217-1331,
``cal_disc_loss`` :1192-1200) and of the teacher step (onmt/trainer_teacher.py:229-267), for
the README flag set (README.md:54-84; carried by ``MSSRConfig``).  Same order of forwards and
backwards, ``no_grad`` regions, train/eval modes, losses, clipping and Adam settings; logged
scalars are returned as a dict instead of going to InfoLogger / TensorBoard.

Data parallelism: ``grad_sync`` (see parallel.py) is called once per optimizer step, after the
last backward of the phase and before clipping, with the networks that are about to step.
"""
from collections.abc import Mapping

import torch
import torch.nn.functional as F

from . import ops
from .decode_engine import soft_decode
from .optim import Adam
from .train_and_translate import dec_output_length, generator_probs
from .utils.misc import sequence_mask


_mode_lists = {}


def set_mode(module, training):
    """``module.train(training)`` without nn.Module's recursive walk: the phases flip the modes of all four networks
    several times per step (onmt/trainer_wgan_wh.py:507-510, 866-869), and ``Module.train`` re-walks ~1 900 submodules
    through ``__setattr__`` every time (7 ms of host time per WGAN step).  The submodule list is cached per network
    (rebuilt when the module count changes) and ``training`` is written into each instance dict; nothing happens when
    the root already has the requested mode."""
    training = bool(training)
    if module.training == training:
        return module
    key = id(module)
    hit = _mode_lists.get(key)
    if hit is None or hit[0]() is not module:
        import weakref
        mods = list(module.modules())
        _mode_lists[key] = hit = (weakref.ref(module), mods)
    for m in hit[1]:
        m.__dict__["training"] = training
    return module


class _held_out:
    """Forward passes inside the block record no gradient for the modules' own parameters (the graph is decided at
    forward time; the block ends before backward())."""

    def __init__(self, on, *modules):
        self.ps = [p for m in modules for p in m.parameters() if p.requires_grad] if on else ()

    def __enter__(self):
        for p in self.ps:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.ps:
            p.requires_grad_(True)


def cal_disc_loss(probs, labels):
    """probs[i, labels[i]] as a [N] tensor (one gather instead of the reference's per-row loop).  ``labels``:
    a device tensor (no host round trip) or a Python list as in the reference."""
    idx = labels if torch.is_tensor(labels) else torch.as_tensor(labels, device=probs.device)
    return probs.gather(1, idx.view(-1, 1)).squeeze(1)


def _to_device_async(t, like):
    """Host tensor -> like's device and dtype without blocking the host (pinned staging + non_blocking copy):
    a plain .to(device) of pageable memory waits for the stream."""
    if not like.is_cuda:
        return t.to(like)
    return t.to(like.dtype).pin_memory().to(like.device, non_blocking=True)


def _masked_penalty(g, mask, k, lamb):
    """((||g_i|| - k)^2 over the rows where mask is set).mean() * lamb for g [n0, n1, dim] and mask Bool [n0, n1], as
    g[mask] would give it but without boolean indexing (nonzero = a device sync in the middle of the phase) and in
    one launch, reading g in whatever row order the graph left it (ops.gp_penalty; the gradient of a zero row is 0,
    as with the reference's gradients.norm(2, dim=1))."""
    return ops.gp_penalty(g, mask, k, lamb)


def convert_sent_ids_to_onehot(sent_ids, vocab_size):
    B, L = sent_ids.shape
    oh = torch.zeros(B * L, vocab_size, device=sent_ids.device)
    return oh.scatter_(1, sent_ids.reshape(-1, 1), 1).view(B, L, vocab_size)


def penalize_gen_style_grad(d_model, real, fake, lengths, labels, k=1.0, lamb=10.0, gp_norm_seq=True, alpha=None):
    """real/fake [B, L, dim]; alpha is drawn from the CPU generator exactly as the reference does
    (torch.rand(B,1,1).type_as(real), trainer_wgan_wh.py:1238) unless the caller drew it already."""
    alpha = _to_device_async(torch.rand(real.size(0), 1, 1) if alpha is None else alpha, real)
    inter = (alpha * real + (1 - alpha) * fake).detach().requires_grad_(True)
    with ops.second_order():
        err = cal_disc_loss(d_model(inter, lengths), labels)
    with ops.input_grads_only():
        g, = torch.autograd.grad(err, inter, torch.ones_like(err), create_graph=True, retain_graph=True)
    if gp_norm_seq:
        return _masked_penalty(g, sequence_mask(lengths, max_len=real.size(1)), k, lamb)
    return ((g.norm(2, dim=1) - k) ** 2).mean() * lamb


def penalize_gen_text_grad(d_model, real, fake, lengths, labels, k=1.0, lamb=10.0, gp_norm_seq=True, alpha=None):
    """real/fake [B, T, V]; the critic takes [T, B, V] (trainer_wgan_wh.py:1283-1331)."""
    x_len = real.size(1)
    alpha = _to_device_async(torch.rand(real.size(0), 1, 1) if alpha is None else alpha, real)
    inter = (alpha * real + (1 - alpha) * fake).transpose(0, 1).detach().requires_grad_(True)
    with ops.second_order():
        err = cal_disc_loss(d_model(inter, lengths), labels)
    with ops.input_grads_only():
        g, = torch.autograd.grad(err, inter, torch.ones_like(err), create_graph=True, retain_graph=True)
    g = g.transpose(0, 1)
    if gp_norm_seq:
        return _masked_penalty(g, sequence_mask(lengths, max_len=x_len), k, lamb)
    return ((g.norm(2, dim=1) - k) ** 2).mean() * lamb


class PhaseLog(Mapping):
    """The logged scalars of one phase, read from the device on FIRST ACCESS.  The reference calls .item() on
    every scalar every step (InfoLogger, trainer_wgan_wh.py:275-316), which drains the GPU queue at the end of
    each phase; a training loop that reports every N steps only pays for the steps it looks at, and the host
    starts the next phase while the GPU finishes this one."""

    def __init__(self, names, values, derive=None, check=None):
        """``check``: (device bool scalar, message) -- raised as a KeyError when the log is first read."""
        vals = [v.detach().float() for v in values]
        self._check = None
        if check is not None and check[0] is not None:
            vals.append(check[0].detach().float())
            self._check = check[1]
        self._names, self._dev, self._derive, self._d = list(names), torch.stack(vals), derive, None

    def _get(self):
        if self._d is None:
            raw = self._dev.tolist()
            if self._check is not None and raw.pop() != 0.0:
                raise KeyError(self._check)
            d = dict(zip(self._names, raw))
            if self._derive is not None:
                d.update(self._derive(d))
            self._d, self._dev = d, None
        return self._d

    def __getitem__(self, k):
        return self._get()[k]

    def __iter__(self):
        return iter(self._get())

    def __len__(self):
        return len(self._get())

    def __repr__(self):
        return "PhaseLog(%r)" % (self._get(),)


class Trainer:
    def __init__(self, nets, cfg, grad_sync=None):
        self.cfg = cfg
        self.model = nets["model"]
        self.style_generator = nets["style_generator"]
        self.style_discriminator = nets["style_discriminator"]
        self.text_discriminator = nets["text_discriminator"]
        self.classifier = nets["teacher"]
        self.classifier.eval()
        self.grad_sync = grad_sync
        # in-tree multi-tensor Adam (optim.py / csrc/optim.hip): norm + clip + update of a whole network in ~11 launches
        adam = lambda m: Adam(m, lr=cfg.learning_rate, betas=(cfg.adam_beta1, cfg.adam_beta2))  # noqa: E731
        self.style_generator_optim = adam(self.style_generator)
        self.style_discriminator_optim = adam(self.style_discriminator)
        self.text_discriminator_optim = adam(self.text_discriminator)
        self.model_optim = adam(self.model)
        dev = next(self.model.parameters()).device
        self.one = torch.ones([], device=dev)
        self.mone = -torch.ones([], device=dev)
        self.pad_tensor = torch.full((1, 1, 1), cfg.pad_idx, dtype=torch.long, device=dev)
        self._label_table = None
        self._bad_labels = None

    # ------------------------------------------------------------------ pieces
    def _gain(self, logits, mode):
        if mode == "normal":
            return logits
        if mode == "sigmoid":
            return torch.sigmoid(logits) * self.cfg.gain_ratio
        raise ValueError("unsupported force_positive_gain_mode: %s" % mode)

    def _pick(self, logits, labels, mode):
        """cal_disc_loss(self._gain(logits, mode), labels) in one launch (ops.gain_pick, twice differentiable)."""
        if mode not in ("normal", "sigmoid"):
            raise ValueError("unsupported force_positive_gain_mode: %s" % mode)
        idx = labels if torch.is_tensor(labels) else torch.as_tensor(labels, device=logits.device)
        return ops.gain_pick(logits, idx, self.cfg.gain_ratio, mode == "sigmoid")

    def _style_logits(self, x, lengths):
        """The style critic's raw logits (fp32 island, see _call_style_discriminator)."""
        if self.cfg.style_critic_fp32 and ops.compute_dtype() != torch.float32:
            with ops.compute_as(torch.float32):
                return self.style_discriminator(x, lengths)
        return self.style_discriminator(x, lengths)

    def _call_text_discriminator(self, x, lengths):
        return self._gain(self.text_discriminator(x, lengths), self.cfg.gain_mode_t)

    def _call_style_discriminator(self, x, lengths):
        # cfg.style_critic_fp32: the style critic (an MLP on [B, L, 256] representations, < 1 % of the step's
        # arithmetic) always computes in fp32, forward, backward and the gradient penalty's double backward
        # (ops.compute_as: every op's backward follows its forward's dtype).  In bf16 its parameter gradients
        # missed the 2e-2 bar against the reference (tests/test_gpu_parity_readme.py) by 2-3x.
        if self.cfg.style_critic_fp32 and ops.compute_dtype() != torch.float32:
            with ops.compute_as(torch.float32):
                return self._gain(self.style_discriminator(x, lengths), self.cfg.gain_mode_s)
        return self._gain(self.style_discriminator(x, lengths), self.cfg.gain_mode_s)

    def unpack_batch(self, batch):
        cfg = self.cfg
        held = getattr(self, "_unpacked", None)
        if held is not None and held[0] is batch:       # train_wgan_step: one unpacking for its three users
            return held[1]
        src, src_lengths = batch.src, batch.src_lengths
        real_src = src[1:]
        real_len = src_lengths - 1
        # style label = position of the sentence's label token in cfg.label_token_ids, looked up on the device
        # (the reference walks a Python list, :389-400; a .tolist() here would drain the GPU queue every phase)
        if self._label_table is None or self._label_table.device != src.device:
            self._label_table = torch.tensor(cfg.label_token_ids, dtype=torch.long, device=src.device)
        hit = src[0, :, 0].unsqueeze(1) == self._label_table.unsqueeze(0)
        labels = hit.long().argmax(1)
        # the reference's convert_label_idx raises a KeyError for an unknown label token; here the check stays on the
        # device (no sync in the phase) and is raised when a PhaseLog of this trainer is next read
        self._bad_labels = (~hit.any(1)).any() if self._bad_labels is None else self._bad_labels | (~hit.any(1)).any()
        labels_list = labels            # the reference's Python list; cal_disc_loss gathers with the tensor itself
        vocab_target = torch.cat([batch.tgt[1:], self.pad_tensor.repeat(1, batch.tgt.size(1), 1)])
        return labels, labels_list, real_src, real_len, vocab_target

    def _label_check(self):
        bad, self._bad_labels = self._bad_labels, None
        return bad, "a sentence's first token is not one of cfg.label_token_ids %r" % (self.cfg.label_token_ids,)

    def vocab_loss(self, log_probs, target, batch_size):
        """VocabLossFn(token_mode=False, all_tokens_aware=sent_stop_aware), onmt/train_util.py:13-32."""
        ignore = -100 if self.cfg.sent_stop_aware else self.cfg.pad_idx
        return F.nll_loss(log_probs, target, ignore_index=ignore, reduction="sum") / batch_size

    def vocab_loss_from_probs(self, probs, target, batch_size):
        """The same loss on probabilities: NLL(log p) only reads p at the target ids, so the log is
        taken after the gather (T*B values) instead of over the whole [T*B, V] tensor."""
        # probs [T, B, V] is usually a slice of a sweep's output: gather in place, no contiguous copy of the slice
        p = probs.gather(2, target.view(probs.size(0), probs.size(1), 1)).reshape(-1)
        nll = -p.log()
        if not self.cfg.sent_stop_aware:
            nll = nll * target.ne(self.cfg.pad_idx).to(nll.dtype)
        return nll.sum() / batch_size

    def cal_style_gap(self, real, gen, lengths):
        gap = ((real.float() - gen.float()) ** 2).sum(2).transpose(0, 1)
        mask = sequence_mask(lengths, max_len=real.size(0))
        return torch.where(mask, gap, torch.zeros_like(gap)).sum() / mask.sum()      # gap[mask].mean(), no sync

    def _gen_real_style_represents(self, real_src, real_len):
        with torch.no_grad():
            _, memory_bank, _ = self.classifier.encoder(real_src, real_len)
        return memory_bank.detach()

    def _gen_style_represents(self, desired, real_src, real_len, eval_rows=0):
        if eval_rows:
            return self.style_generator(desired, real_src, real_len, eval_rows=eval_rows)
        return self.style_generator(desired, real_src, real_len)

    def _encode(self, real_src, real_len, style_rep, eval_rows=0):
        if eval_rows:
            enc_state, memory_bank, lengths = self.model.encoder(real_src, real_len, eval_rows=eval_rows)
        else:
            enc_state, memory_bank, lengths = self.model.encoder(real_src, real_len)
        return enc_state, self.model.bond_layer(memory_bank, style_rep), lengths

    def _self_reconstruct(self, real_src, real_len, tgt, style_rep):
        enc_state, mem, lengths = self._encode(real_src, real_len, style_rep)
        self.model.decoder.init_state(real_src, mem, enc_state)
        dec_out, _ = self.model.decoder(tgt, mem, memory_lengths=lengths)
        return generator_probs(self.model, dec_out, self.cfg.dec_temperature, log=True)

    def _dynamic_decoding(self, real_src, real_len, style_rep, max_dec_steps):
        """-> (probs [steps, B, V], lengths Long[B] on the device, argmax ids [steps, B]); no host sync."""
        enc_state, mem, _ = self._encode(real_src, real_len, style_rep)
        self.model.decoder.init_state(real_src, mem, enc_state)
        probs = soft_decode(self.model, self.cfg, mem, real_len, max_dec_steps)
        lens, ids = dec_output_length(probs, self.cfg)
        return probs, lens, ids

    def _arm(self, *modules):
        """Start overlapping the gradient all-reduce of ``modules`` with their LAST backward."""
        if self.grad_sync is not None:
            for m in modules:
                self.grad_sync.arm(m)

    def _finish(self, modules, optims, clip=True):
        if self.grad_sync is not None:
            self.grad_sync(modules)
        # clip_grad_norm_(net, max_grad_norm) per network, then its Adam step (trainer_wgan_wh.py:623-628,973-977);
        # the step itself tells the weight caches about the update
        for o in optims:
            o.step(self.cfg.max_grad_norm if clip else 0.0)

    # ------------------------------------------------------------------ phases
    def train_discriminator(self, batch, _shared=None):
        """``_shared`` = (style representation, fake-text probabilities) computed by train_wgan_step in the sweep it
        shares with the generator phase; None: computed here, as the reference does (:530-535)."""
        cfg = self.cfg
        if _shared is None:                 # with the shared sweep the generator side does not run in this phase
            set_mode(self.style_generator, False)
            set_mode(self.model, False)
        set_mode(self.style_discriminator, True)
        set_mode(self.text_discriminator, True)
        self.style_discriminator_optim.zero_grad()
        self.text_discriminator_optim.zero_grad()
        labels, labels_list, real_src, real_len, vocab_target = self.unpack_batch(batch)
        text_len = (real_len + 1).detach()
        if cfg.discr_sent_stop_aware:
            text_len = torch.ones_like(real_len) * vocab_target.size(0)
        B = real_len.numel()
        alpha_t, alpha_s = torch.rand(B, 1, 1), torch.rand(B, 1, 1)       # in the order the reference draws them
        # The generator side is in eval mode: its style representation is deterministic, so the reference's two
        # no_grad calls (:533 for the fake text, :571 for the fake style) are one here.
        if _shared is not None:
            rep = _shared[0]
        else:
            with torch.no_grad():
                rep = self._gen_style_represents(labels, real_src, real_len)
        real_rep_t = self._gen_real_style_represents(real_src, real_len).transpose(0, 1)
        gen_rep_t = rep.transpose(0, 1)

        if cfg.batch_critic_passes:
            if _shared is not None:
                probs = _shared[1]
            else:
                with torch.no_grad():
                    probs, _, _ = self._dynamic_decoding(real_src, real_len, rep, vocab_target.size(0))
            gain_t_real, gain_t_fake, t_gp = self._text_critic_batched(vocab_target, probs, text_len, labels_list, alpha_t)
            gain_s_real, gain_s_fake, s_gp = self._style_critic_batched(real_rep_t, gen_rep_t, real_len, labels_list,
                                                                        alpha_s)
            return self._finish_discriminator(gain_t_real, gain_t_fake, t_gp, gain_s_real, gain_s_fake, s_gp)
        # text critic: real / fake / gradient penalty
        gain_t_real = cal_disc_loss(self._call_text_discriminator(vocab_target, text_len), labels_list).mean()
        gain_t_real.backward(self.mone)
        if _shared is not None:
            probs = _shared[1]
        else:
            with torch.no_grad():
                probs, _, _ = self._dynamic_decoding(real_src, real_len, rep, vocab_target.size(0))
        gain_t_fake = cal_disc_loss(self._call_text_discriminator(probs, text_len), labels_list).mean()
        gain_t_fake.backward(self.one)
        with torch.no_grad():
            probs_t = probs.transpose(0, 1)
            real_onehot = convert_sent_ids_to_onehot(vocab_target.transpose(0, 1).squeeze(2), cfg.vocab_size)
        t_gp = penalize_gen_text_grad(self._call_text_discriminator, real_onehot, probs_t, text_len, labels_list,
                                      k=cfg.wgan_text_k, lamb=cfg.wgan_text_lamb, gp_norm_seq=cfg.use_gp_norm_seq,
                                      alpha=alpha_t)
        self._arm(self.text_discriminator)
        t_gp.backward()
        # style critic (run on a helper thread / side stream during the decode it measured 2 % SLOWER: launches
        # from two threads contend in the runtime and stretch the GPU-bound decode)
        gain_s_real = cal_disc_loss(self._call_style_discriminator(real_rep_t, real_len), labels_list).mean()
        gain_s_real.backward(self.mone)
        gain_s_fake = cal_disc_loss(self._call_style_discriminator(gen_rep_t, real_len), labels_list).mean()
        gain_s_fake.backward(self.one)
        s_gp = penalize_gen_style_grad(self._call_style_discriminator, real_rep_t.float(), gen_rep_t.float(),
                                       real_len, labels_list, k=cfg.wgan_style_k, lamb=cfg.wgan_style_lamb,
                                       gp_norm_seq=cfg.use_gp_norm_seq, alpha=alpha_s)
        self._arm(self.style_discriminator)
        s_gp.backward()
        return self._finish_discriminator(gain_t_real, gain_t_fake, t_gp, gain_s_real, gain_s_fake, s_gp)

    def _penalty(self, g, lengths, max_len, k, lamb):
        """gradient penalty of per-token input gradients g [B, L, dim] (penalize_gen_*_grad, :1253-1261, :1321-1329)."""
        if self.cfg.use_gp_norm_seq:
            return _masked_penalty(g, sequence_mask(lengths, max_len=max_len), k, lamb)
        return ((g.norm(2, dim=1) - k) ** 2).mean() * lamb

    def _critic_weights(self, B, like):
        """Row weights of a batched critic pass [real | fake | interpolated]: the cotangent that selects the
        interpolated rows (grad_outputs of the penalty's input gradient) and the first-order objective
        mean(fake) - mean(real) as a dot product -- no slices of the gains, whose backward passes are a fill and a
        copy each."""
        key = (B, like.device, like.dtype)
        hit = getattr(self, "_cw", None)
        if hit is None or hit[0] != key:
            w_err = torch.cat([like.new_zeros(2 * B), like.new_ones(B)])
            w_obj = torch.cat([like.new_full((B,), -1.0 / B), like.new_full((B,), 1.0 / B), like.new_zeros(B)])
            self._cw = hit = (key, w_err, w_obj)
        return hit[1], hit[2]

    def _text_critic_batched(self, vocab_target, probs, text_len, labels, alpha_t):
        """The text critic's three passes of train_discriminator (real ids :545-551, fake soft words :553-560,
        interpolated :562-569 / :1283-1331) as ONE pass on [T, 3B] rows: the real ids go through the embedding lookup
        (so the padding row still receives no gradient from them), fake and interpolated words through the soft
        product, and the encoder runs once on the concatenation.  Sentences are independent in the critic, so the
        input gradient of the interpolated rows is the gradient penalty's and one backward of
        fake - real + penalty accumulates what the reference's three backward calls do."""
        cfg = self.cfg
        B = labels.numel()
        critic = self.text_discriminator
        T = vocab_target.size(0)
        with torch.no_grad():
            # soft words [T, 2B, V] = fake | interpolated, written in place (no concatenation).  alpha * onehot(real) +
            # (1 - alpha) * fake (:1298-1300) without the one-hot tensor: the fake words scaled, then alpha added at
            # each real id
            alpha = _to_device_async(alpha_t, self.one).view(1, B, 1)
            soft = torch.empty(T, 2 * B, cfg.vocab_size, dtype=torch.float32, device=probs.device)
            soft[:, :B].copy_(probs)
            torch.mul(probs, 1 - alpha, out=soft[:, B:])
            soft[:, B:].scatter_add_(2, vocab_target, alpha.expand(T, B, 1))
        soft.requires_grad_(True)      # the critic's sentences are independent: rows B.. of d/d soft are d/d interpolated
        with ops.second_order():
            emb = torch.cat([critic.encoder.embeddings(vocab_target), critic.encoder.embeddings(soft)], 1)
            picked = self._pick(critic.forward_embedded(emb, text_len.repeat(3)), labels.repeat(3), cfg.gain_mode_t)
        w_err, w_obj = self._critic_weights(B, picked)
        with ops.input_grads_only():
            g, = torch.autograd.grad(picked, soft, w_err, create_graph=True, retain_graph=True)
        gp = self._penalty(g[:, B:].transpose(0, 1), text_len, T, cfg.wgan_text_k, cfg.wgan_text_lamb)
        gains = picked.detach().view(3, B)[:2].mean(1)
        self._arm(critic)
        (torch.dot(picked, w_obj) + gp).backward()
        return gains[0], gains[1], gp

    def _style_critic_batched(self, real_rep_t, gen_rep_t, lengths, labels, alpha_s):
        """The style critic on real, generated and interpolated representations [B, L, dim] in one pass (:571-620)."""
        cfg = self.cfg
        B = labels.numel()
        with torch.no_grad():
            # real | fake | interpolated rows written in place (fp32): two casting copies and one lerp
            # (fake + alpha (real - fake) = alpha real + (1 - alpha) fake, :1238-1240), no concatenation
            x = torch.empty((3 * B,) + tuple(real_rep_t.shape[1:]), dtype=torch.float32, device=real_rep_t.device)
            x[:B].copy_(real_rep_t)
            x[B:2 * B].copy_(gen_rep_t)
            torch.lerp(x[B:2 * B], x[:B], _to_device_async(alpha_s, x), out=x[2 * B:])
        x.requires_grad_(True)         # sentences are independent in the critic: rows 2B.. of d/dx are d/d interpolated
        with ops.second_order():
            picked = self._pick(self._style_logits(x, lengths.repeat(3)), labels.repeat(3), cfg.gain_mode_s)
        w_err, w_obj = self._critic_weights(B, picked)
        with ops.input_grads_only():
            g, = torch.autograd.grad(picked, x, w_err, create_graph=True, retain_graph=True)
        gp = self._penalty(g[2 * B:], lengths, x.size(1), cfg.wgan_style_k, cfg.wgan_style_lamb)
        gains = picked.detach().view(3, B)[:2].mean(1)
        self._arm(self.style_discriminator)
        (torch.dot(picked, w_obj) + gp).backward()
        return gains[0], gains[1], gp

    def _finish_discriminator(self, gain_t_real, gain_t_fake, t_gp, gain_s_real, gain_s_fake, s_gp):
        self._finish((self.style_discriminator, self.text_discriminator),
                     (self.style_discriminator_optim, self.text_discriminator_optim))
        return PhaseLog(("TR_gain", "TF_gain", "T_gp", "SR_gain", "SF_gain", "S_gp"),
                        (gain_t_real, gain_t_fake, t_gp, gain_s_real, gain_s_fake, s_gp), check=self._label_check(),
                        derive=lambda d: dict(T_cost=d["TF_gain"] - d["TR_gain"] + d["T_gp"], T_Wgap=d["TR_gain"] - d["TF_gain"],
                                       S_cost=d["SF_gain"] - d["SR_gain"] + d["S_gp"], S_Wgap=d["SR_gain"] - d["SF_gain"]))

    def _generator_forward_batched(self, labels, real_src, real_len, tgt_len, with_eval=False):
        """Style representations, bonded encoder states and ONE soft decode for both halves of the generator phase
        (adversarial text: desired style, tgt_len + 10 steps; self-style text: own style, tgt_len steps).
        ``with_eval``: the critics' inputs -- the own-style representation and the text decoded from it, which
        train_discriminator computes in eval mode under no_grad (:530-535, :571) -- ride along as B more sentences at
        the front of every call (``eval_rows``: no dropout, no gradient): returns them as well."""
        cfg = self.cfg
        B = real_len.numel()
        desired = 1 - labels
        max_dec_steps = tgt_len + cfg.max_extra_dec_steps
        if not with_eval:
            rep_all = self._gen_style_represents(torch.cat([desired, desired, labels]), real_src.repeat(1, 3, 1),
                                                 real_len.repeat(3))
            rep, rep2, rep3 = rep_all[:, :B], rep_all[:, B:2 * B], rep_all[:, 2 * B:]
            src2, len2 = real_src.repeat(1, 2, 1), real_len.repeat(2)
            enc_state, mem, _ = self._encode(src2, len2, torch.cat([rep3, rep], 1))      # the shorter decode first
            self.model.decoder.init_state(src2, mem, enc_state)
            sprobs, probs = soft_decode(self.model, cfg, mem, len2, max_dec_steps,
                                        groups=[(tgt_len, 0, B), (max_dec_steps, B, 2 * B)])
            return rep2, probs, sprobs, None, None
        src3, len3 = real_src.repeat(1, 3, 1), real_len.repeat(3)
        if 3 * B * real_src.size(0) > cfg.eval_rows_max_tokens:
            # long batches fill the chip with the training sentences alone: the eval-mode sentences would only add
            # their rows to the encoders' backward.  Encode them on their own (forward-only plans), share the decode.
            set_mode(self.style_generator, False)
            set_mode(self.model, False)
            with torch.no_grad():
                rep_d = self._gen_style_represents(labels, real_src, real_len)
                _, mem_d, _ = self._encode(real_src, real_len, rep_d)
            set_mode(self.style_generator, True)
            set_mode(self.model, True)
            rep_all = self._gen_style_represents(torch.cat([desired, desired, labels]), real_src.repeat(1, 3, 1), len3)
            rep, rep2, rep3 = rep_all[:, :B], rep_all[:, B:2 * B], rep_all[:, 2 * B:]
            _, mem, _ = self._encode(src3[:, B:], len3[B:], torch.cat([rep3, rep], 1))
            mem = torch.cat([mem_d.to(mem.dtype), mem], 1)
        else:
            rep_all = self._gen_style_represents(torch.cat([labels, desired, desired, labels]), real_src.repeat(1, 4, 1),
                                                 real_len.repeat(4), eval_rows=B)
            rep_d = rep_all[:, :B].detach()
            rep, rep2, rep3 = rep_all[:, B:2 * B], rep_all[:, 2 * B:3 * B], rep_all[:, 3 * B:]
            _, mem, _ = self._encode(src3, len3, torch.cat([rep_d, rep3, rep], 1), eval_rows=B)
        self.model.decoder.init_state(src3, None, None)
        # sentence order: critics' fake text (eval, tgt_len steps) | self-style text (tgt_len steps) | adversarial text
        # (all steps) -- the longest decode last, so that the late backward steps run on it alone
        fake, sprobs, probs = soft_decode(self.model, cfg, mem, len3, max_dec_steps, eval_rows=B,    # no gradient into
                                          groups=[(tgt_len, 0, B), (tgt_len, B, 2 * B),           # mem[:, :B]
                                                  (max_dec_steps, 2 * B, 3 * B)])
        return rep2, probs, sprobs, fake.detach(), rep_d

    def _train_generator_batched(self, batch, _shared=None):
        """train_generator with its independent pieces batched (cfg.batch_generator_phase): the reference
        runs the style generator three times and encoder + soft decode twice on the SAME weights and source
        (onmt/trainer_wgan_wh.py:898-968); here they are one style-generator call on 3B rows and one
        encoder + decode on 2B rows (the self-style half only needs the first tgt_len steps), and the three
        backward() calls become one backward of the summed objective.  Same losses and gradients up to
        floating-point summation order (the golden test holds both variants to the same tolerance); with
        dropout on, the masks differ as they would between any two runs."""
        cfg = self.cfg
        labels, labels_list, real_src, real_len, vocab_target = self.unpack_batch(batch)
        tgt_len, B = vocab_target.size(0), vocab_target.size(1)
        desired = 1 - labels
        desired_list = desired
        max_dec_steps = tgt_len + cfg.max_extra_dec_steps
        if _shared is not None:
            rep2, probs, sprobs = _shared
        else:
            rep2, probs, sprobs, _, _ = self._generator_forward_batched(labels, real_src, real_len, tgt_len)
        seq_lens, _ = dec_output_length(probs, cfg)
        lens = seq_lens if not cfg.discr_sent_stop_aware else torch.ones_like(real_len) * max_dec_steps
        with _held_out(cfg.critics_frozen_in_generator_phase, self.text_discriminator, self.style_discriminator):
            gain_t = self._pick(self.text_discriminator(probs, lens), desired_list, cfg.gain_mode_t).mean()
            gain_s = self._pick(self._style_logits(rep2.transpose(0, 1), real_len), desired_list, cfg.gain_mode_s).mean()
        with torch.no_grad():
            acc = self.classifier(probs, lens).argmax(-1).eq(desired).float().mean()
        vl = self.vocab_loss_from_probs(sprobs, vocab_target.contiguous().view(-1), B)
        total = vl - cfg.text_ap_scale * gain_t - cfg.style_ap_scale * gain_s
        self._arm(self.style_generator, self.model)
        total.backward()
        self._finish((self.style_generator, self.model), (self.style_generator_optim, self.model_optim))
        return PhaseLog(("AT_gain", "AT_acc", "AS_gain", "T_selfstyle_vloss"), (gain_t, acc, gain_s, vl),
                        check=self._label_check())

    def train_wgan_step(self, batch):
        """``train_discriminator(batch)`` then ``train_generator(batch)`` -- one WGAN step -- with the same results
        as the two calls (same weights read, same order of optimizer updates; the bits of the critics' fake text are
        unchanged, the generator's dropout masks differ as between any two runs) and ONE decode sweep instead of two:
        the critics' fake text (eval mode, no gradient, tgt_len steps; onmt/trainer_wgan_wh.py:530-535) and the
        generator phase's two decodes (train mode, :917-920, :955-968) read the same generator weights, and a
        sweep's cost is its ~100 dependent launches per step, not its row count.  Returns (d_log, g_log)."""
        cfg = self.cfg
        if not (cfg.ap_gen_self_style_text and cfg.batch_generator_phase and cfg.share_decode_sweep):
            return self.train_discriminator(batch), self.train_generator(batch)
        unpacked = self.unpack_batch(batch)
        labels, labels_list, real_src, real_len, vocab_target = unpacked
        self._unpacked = (batch, unpacked)
        try:
            return self._wgan_step_shared(batch, labels, real_src, real_len, vocab_target)
        finally:
            self._unpacked = None

    def _wgan_step_shared(self, batch, labels, real_src, real_len, vocab_target):
        tgt_len = vocab_target.size(0)
        # generator side in train mode: the forward half of train_generator.  What train_discriminator computes in
        # eval mode under no_grad (:530-535: own-style representation -> bonded encoder states -> decoded text) rides
        # along as the first B sentences of the same style-generator, encoder and decode calls (eval_rows: those
        # sentences skip every dropout site and receive no gradient)
        set_mode(self.style_generator, True)
        set_mode(self.model, True)
        self.style_generator_optim.zero_grad()
        self.model_optim.zero_grad()
        rep2, probs, sprobs, fake, rep_d = self._generator_forward_batched(labels, real_src, real_len, tgt_len,
                                                                          with_eval=True)
        d_log = self.train_discriminator(batch, _shared=(rep_d, fake))
        g_log = self.train_generator(batch, _shared=(rep2, probs, sprobs))
        return d_log, g_log

    def train_generator(self, batch, _shared=None):
        cfg = self.cfg
        set_mode(self.style_generator, True)
        set_mode(self.model, True)
        set_mode(self.style_discriminator, False)
        set_mode(self.text_discriminator, False)
        if _shared is None:
            self.style_generator_optim.zero_grad()
            self.model_optim.zero_grad()
        if cfg.ap_gen_self_style_text and cfg.batch_generator_phase:
            return self._train_generator_batched(batch, _shared)
        labels, labels_list, real_src, real_len, vocab_target = self.unpack_batch(batch)
        tgt_len, B = vocab_target.size(0), vocab_target.size(1)
        desired = 1 - labels                                   # num_styles == 2 (train_and_translate.py:72-73)
        desired_list = desired
        max_dec_steps = tgt_len + cfg.max_extra_dec_steps
        rep = self._gen_style_represents(desired, real_src, real_len)
        probs, seq_lens, _ = self._dynamic_decoding(real_src, real_len, rep, max_dec_steps)
        lens = seq_lens if not cfg.discr_sent_stop_aware else torch.ones_like(real_len) * max_dec_steps
        gain_t = cal_disc_loss(self._call_text_discriminator(probs, lens), desired_list).mean()
        if not cfg.ap_gen_self_style_text:
            self._arm(self.model)
        gain_t.backward(self.mone * cfg.text_ap_scale)
        with torch.no_grad():
            acc = self.classifier(probs, lens).argmax(-1).eq(desired).float().mean()
        rep2_t = self._gen_style_represents(desired, real_src, real_len).transpose(0, 1)
        gain_s = cal_disc_loss(self._call_style_discriminator(rep2_t, real_len), desired_list).mean()
        if not cfg.ap_gen_self_style_text:
            self._arm(self.style_generator)
        gain_s.backward(self.mone * cfg.style_ap_scale)
        vals = [gain_t, acc, gain_s]
        if cfg.ap_gen_self_style_text:
            rep3 = self._gen_style_represents(labels, real_src, real_len)
            sprobs, _, _ = self._dynamic_decoding(real_src, real_len, rep3, tgt_len)
            vl = self.vocab_loss_from_probs(sprobs, vocab_target.contiguous().view(-1), B)
            self._arm(self.style_generator, self.model)
            vl.backward()
            vals.append(vl)
        self._finish((self.style_generator, self.model), (self.style_generator_optim, self.model_optim))
        names = ["AT_gain", "AT_acc", "AS_gain"] + (["T_selfstyle_vloss"] if cfg.ap_gen_self_style_text else [])
        return PhaseLog(names, vals, check=self._label_check())

    def _train_reconstruct_batched(self, batch):
        """train_reconstruct with its independent pieces batched (cfg.batch_generator_phase), as in
        _train_generator_batched: the reference (onmt/trainer_wgan_wh.py:729-863) runs the style generator
        twice, the main encoder three times and the soft decode twice on the same weights and source; here
        they are one style-generator call on 2B rows, one encoder call on 3B rows (teacher-forced third
        first) and one soft decode on 2B rows (the reconstruction half only needs the first tgt_len steps).
        This phase is 25 of the 31 calls of the README schedule."""
        cfg = self.cfg
        labels, labels_list, real_src, real_len, vocab_target = self.unpack_batch(batch)
        tgt_len, B = vocab_target.size(0), vocab_target.size(1)
        flat_target = vocab_target.contiguous().view(-1)
        desired = 1 - labels
        max_dec_steps = tgt_len + cfg.max_extra_dec_steps
        real_rep = self._gen_real_style_represents(real_src, real_len)
        rep_both = self._gen_style_represents(torch.cat([labels, desired]), real_src.repeat(1, 2, 1), real_len.repeat(2))
        rep_src, rep_tgt = rep_both[:, :B], rep_both[:, B:]
        src3, len3 = real_src.repeat(1, 3, 1), real_len.repeat(3)
        enc_state, mem, lengths = self._encode(src3, len3, torch.cat([real_rep.to(rep_both.dtype), rep_both], 1))
        if cfg.forced_rows_in_sweep and batch.tgt.size(0) <= max_dec_steps:
            # the teacher-forced self reconstruction of the first third rides in the sweep of the two soft decodes
            # (decode_engine.soft_decode, forced_ids): one pass over the decoder's weights instead of two
            self.model.decoder.init_state(src3, None, None)
            forced, probs, cprobs = soft_decode(self.model, cfg, mem, len3, max_dec_steps, forced_ids=batch.tgt[:, :, 0],
                                                groups=[(tgt_len, 0, B), (tgt_len, B, 2 * B), (max_dec_steps, 2 * B, 3 * B)])
            l_noised = self.vocab_loss_from_probs(forced, flat_target, B)
        else:
            # teacher-forced self reconstruction on the first third
            self.model.decoder.init_state(real_src, mem[:, :B], enc_state[:, :B])
            dec_out, _ = self.model.decoder(batch.tgt, mem[:, :B], memory_lengths=lengths[:B])
            lp = generator_probs(self.model, dec_out, cfg.dec_temperature, log=True)
            l_noised = self.vocab_loss(lp.view(-1, lp.size(-1)), flat_target, B)
            # both soft decodes at once
            src2, len2 = src3[:, B:], len3[B:]
            self.model.decoder.init_state(src2, mem[:, B:], enc_state[:, B:])
            probs_all = soft_decode(self.model, cfg, mem[:, B:], len2, max_dec_steps)
            probs, cprobs = probs_all[:tgt_len, :B], probs_all[:, B:]
        l_rec = self.vocab_loss_from_probs(probs, flat_target, B)
        total = cfg.sr_loss_ratio * l_noised + cfg.sr_loss_ratio * l_rec
        s_src = self.classifier.do_cls(rep_src.transpose(0, 1), real_len)
        l_s_src = F.nll_loss(F.log_softmax(s_src, -1), labels)
        s_tgt = self.classifier.do_cls(rep_tgt.transpose(0, 1), real_len)
        l_s_tgt = F.nll_loss(F.log_softmax(s_tgt, -1), desired)
        total = total + cfg.cls_loss_ratio * l_s_src + cfg.cls_loss_ratio * l_s_tgt
        names, vals = ["T_CONSTRUCT_R_VL", "T_CONSTRUCT_F_VL"], [l_noised, l_rec]
        if not cfg.without_stl_penalty:
            l_gap = self.cal_style_gap(real_rep, rep_src, real_len)
            total = total + cfg.alpha_style_gap * l_gap
        clens, _ = dec_output_length(cprobs, cfg)
        t_logits = self.classifier(cprobs, clens)
        l_t = F.nll_loss(F.log_softmax(t_logits, -1), desired)
        total = total + cfg.cls_loss_ratio * l_t
        names += ["T_CLS_loss", "T_CLS_acc", "S_CLS_loss", "S_CLS_acc", "SCur_CLS_loss", "SCur_CLS_acc"]
        vals += [l_t, t_logits.argmax(-1).eq(desired).float().mean(), l_s_tgt,
                 s_tgt.argmax(-1).eq(desired).float().mean(), l_s_src, s_src.argmax(-1).eq(labels).float().mean()]
        if not cfg.without_stl_penalty:
            names.append("S_GAP_loss")
            vals.append(l_gap)
        self._arm(self.style_generator, self.model)
        total.backward()
        self._finish((self.style_generator, self.model), [self.style_generator_optim, self.model_optim], clip=False)
        return PhaseLog(names, vals, check=self._label_check())

    def train_reconstruct(self, batch):
        cfg = self.cfg
        set_mode(self.style_generator, True)
        set_mode(self.model, True)
        self.style_generator_optim.zero_grad()
        self.model_optim.zero_grad()
        if cfg.batch_generator_phase and cfg.use_cls_feedback and cfg.sr_noise_prob == 0:
            return self._train_reconstruct_batched(batch)
        labels, labels_list, real_src, real_len, vocab_target = self.unpack_batch(batch)
        tgt_len, B = vocab_target.size(0), vocab_target.size(1)
        flat_target = vocab_target.contiguous().view(-1)
        desired = 1 - labels
        max_dec_steps = tgt_len + cfg.max_extra_dec_steps
        real_rep = self._gen_real_style_represents(real_src, real_len)
        lp = self._self_reconstruct(real_src, real_len, batch.tgt, real_rep)         # sr_noise_prob == 0
        l_noised = self.vocab_loss(lp.view(-1, lp.size(-1)), flat_target, B)
        total = cfg.sr_loss_ratio * l_noised
        rep_src = self._gen_style_represents(labels, real_src, real_len)
        rep_tgt = self._gen_style_represents(desired, real_src, real_len)
        probs, _, _ = self._dynamic_decoding(real_src, real_len, rep_src, tgt_len)
        l_rec = self.vocab_loss_from_probs(probs, flat_target, B)
        total = total + cfg.sr_loss_ratio * l_rec
        names, vals = ["T_CONSTRUCT_R_VL", "T_CONSTRUCT_F_VL"], [l_noised, l_rec]
        if cfg.use_cls_feedback:
            s_src = self.classifier.do_cls(rep_src.transpose(0, 1), real_len)
            l_s_src = F.nll_loss(F.log_softmax(s_src, -1), labels)
            s_tgt = self.classifier.do_cls(rep_tgt.transpose(0, 1), real_len)
            l_s_tgt = F.nll_loss(F.log_softmax(s_tgt, -1), desired)
            total = total + cfg.cls_loss_ratio * l_s_src + cfg.cls_loss_ratio * l_s_tgt
        if not cfg.without_stl_penalty:
            l_gap = self.cal_style_gap(real_rep, rep_src, real_len)
            total = total + cfg.alpha_style_gap * l_gap
        if cfg.use_cls_feedback:
            cprobs, clens, _ = self._dynamic_decoding(real_src, real_len, rep_tgt, max_dec_steps)
            t_logits = self.classifier(cprobs, clens)
            l_t = F.nll_loss(F.log_softmax(t_logits, -1), desired)
            total = total + cfg.cls_loss_ratio * l_t
            names += ["T_CLS_loss", "T_CLS_acc", "S_CLS_loss", "S_CLS_acc", "SCur_CLS_loss", "SCur_CLS_acc"]
            vals += [l_t, t_logits.argmax(-1).eq(desired).float().mean(), l_s_tgt,
                     s_tgt.argmax(-1).eq(desired).float().mean(), l_s_src,
                     s_src.argmax(-1).eq(labels).float().mean()]
        if not cfg.without_stl_penalty:
            names.append("S_GAP_loss")
            vals.append(l_gap)
        self._arm(self.style_generator, self.model)
        total.backward()
        # no clipping in this phase; the style generator steps only if it received a loss (:859-863)
        optims = [self.model_optim]
        if cfg.use_cls_feedback or not cfg.without_stl_penalty:
            optims.insert(0, self.style_generator_optim)
        self._finish((self.style_generator, self.model), optims, clip=False)
        return PhaseLog(names, vals, check=self._label_check())

    def init_launch(self, batch):
        set_mode(self.style_generator, True)
        set_mode(self.model, True)
        self.style_generator_optim.zero_grad()
        self.model_optim.zero_grad()
        labels, labels_list, real_src, real_len, vocab_target = self.unpack_batch(batch)
        B = len(labels_list)
        real_rep = self._gen_real_style_represents(real_src, real_len)
        l_style = self.cal_style_gap(real_rep, self._gen_style_represents(labels, real_src, real_len), real_len)
        self._arm(self.style_generator)
        l_style.backward()
        lp = self._self_reconstruct(real_src, real_len, batch.tgt, real_rep)
        l_v = self.vocab_loss(lp.view(-1, lp.size(-1)), vocab_target.contiguous().view(-1), B)
        self._arm(self.model)
        l_v.backward()
        self._finish((self.style_generator, self.model), (self.style_generator_optim, self.model_optim))
        return PhaseLog(("INIT_SL", "INIT_VL"), (l_style, l_v), check=self._label_check())


def teacher_step_loss(classifier, batch, cfg, batched=True):
    """onmt/trainer_teacher.py:238-252: three forwards (ids / ids+</s> / full padded length), NLL summed.
    ``batched``: the three forwards are one call on 3B sentences (same weights; the shorter input is padded with
    <blank> rows, which every length mask of the classifier ignores) -- the same loss and gradients up to summation
    order, a third of the launches of a step that is launch-bound at bs16.  False: the reference's three calls."""
    src, src_lengths = batch.src, batch.src_lengths
    real_src, real_len = src[1:], src_lengths - 1
    labels = torch.tensor([cfg.label_token_ids.index(t) for t in src[0, :, 0].tolist()], device=src.device)
    pad_row = torch.full((1, batch.tgt.size(1), 1), cfg.pad_idx, dtype=torch.long, device=src.device)
    new_x = torch.cat([batch.tgt[1:], pad_row])
    nll = lambda lg: F.nll_loss(F.log_softmax(lg, -1), labels)  # noqa: E731
    if batched:
        B, Lmax = labels.numel(), max(real_src.size(0), new_x.size(0))
        grow = lambda x: x if x.size(0) == Lmax else torch.cat([x, pad_row.expand(Lmax - x.size(0), -1, -1)])  # noqa: E731
        x3 = torch.cat([grow(real_src), grow(new_x), grow(new_x)], 1)
        len3 = torch.cat([real_len, real_len + 1, torch.ones_like(src_lengths) * new_x.size(0)])
        lg3 = classifier(x3, len3)
        return nll(lg3[:B]) + nll(lg3[B:2 * B]) + nll(lg3[2 * B:]), lg3[:B]
    logits = classifier(real_src, real_len)
    loss = nll(logits) + nll(classifier(new_x, real_len + 1))
    loss = loss + nll(classifier(new_x, torch.ones_like(src_lengths) * new_x.size(0)))
    return loss, logits
