"""Shapes and training flags of the MSSRNet networks on the hot path.

Defaults are the README recipe of the reference (README.md:54-84): main model 12 layers,
d=512, 8 heads, ff=1024; teacher / style generator 3 layers 8x32 (ff = 4*d,
onmt/style_generator_builder.py:310,558); text discriminator 1 layer 8x32; style
discriminator 3 hidden layers of 8x64; Adam(lr 4e-5, betas 0.5/0.9), clip 100.
"""
from dataclasses import dataclass, replace


@dataclass
class MSSRConfig:
    vocab_size: int = 9606
    pad_idx: int = 1
    unk_idx: int = 0
    bos_idx: int = 2
    eos_idx: int = 3
    dot_idx: int = 6
    excl_idx: int = 7
    qst_idx: int = 8
    label_token_ids: tuple = (4, 5)
    num_styles: int = 2
    faketruthstyle_mode: bool = True          # opts.py:41-44 (critics emit 2*num_styles logits)

    # main encoder/decoder
    layers: int = 12
    d_model: int = 512
    heads: int = 8
    d_ff: int = 1024
    # teacher classifier, style generator, text / style discriminators: (layers, heads, dim/head)
    cls: tuple = (3, 8, 32)
    sgen: tuple = (3, 8, 32)
    tdisc: tuple = (1, 8, 32)
    sdisc: tuple = (3, 8, 64)

    dropout: float = 0.1
    attention_dropout: float = 0.1

    # trainer flags (README.md:64-83, opts.py:425-507)
    learning_rate: float = 4e-5
    adam_beta1: float = 0.5
    adam_beta2: float = 0.9
    max_grad_norm: float = 100.0
    use_gp_norm_seq: bool = True
    gain_mode_t: str = "sigmoid"
    gain_mode_s: str = "sigmoid"
    gain_ratio: float = 1.0
    wgan_text_k: float = 1.0
    wgan_text_lamb: float = 10.0
    wgan_style_k: float = 1.0
    wgan_style_lamb: float = 10.0
    sent_stop_aware: bool = True
    discr_sent_stop_aware: bool = False
    ap_gen_self_style_text: bool = True
    use_cls_feedback: bool = True
    without_stl_penalty: bool = False
    sr_noise_prob: float = 0.0
    sr_loss_ratio: float = 1.0
    cls_loss_ratio: float = 1.0
    alpha_style_gap: float = 1.0
    alpha_text_ap: float = 1.0
    alpha_style_ap: float = 1.0
    text_ap_scale: float = 1.0
    style_ap_scale: float = 1.0
    max_extra_dec_steps: int = 10             # trainer_wgan_wh.py:149
    dec_temperature: float = 1.0
    max_dec_steps: int = 40
    # not a reference flag: batch the independent style-generator / encoder / decode calls of the generator
    # phase (same weights, same source) into single calls; results equal up to summation order
    batch_generator_phase: bool = True
    # not a reference flag: Trainer.train_wgan_step decodes the critics' fake text and the generator phase's texts in
    # one sweep (False: the two phases run back to back exactly as train_discriminator + train_generator)
    share_decode_sweep: bool = True
    # not a reference flag: train_discriminator runs each critic ONCE on its real, fake and interpolated inputs
    # batched behind the embedding layer and backpropagates the summed objective once (the reference: three forward /
    # backward passes per critic, onmt/trainer_wgan_wh.py:545-620); same losses and gradients up to summation order
    batch_critic_passes: bool = True
    # not a reference flag: in Trainer.train_wgan_step the critics' eval-mode inputs ride through the style generator and
    # the encoder as eval_rows of the generator phase's calls while the batched call has at most this many tokens
    # (3 * B * source length: short batches leave the chip under-filled, the extra rows are free); beyond it they are
    # encoded on their own, forward-only, as train_discriminator does
    eval_rows_max_tokens: int = 16384
    # not a reference flag: train_reconstruct's teacher-forced decoder pass (onmt/trainer_wgan_wh.py:632-664) runs as
    # B more sentences of the sweep that decodes its two soft texts (mssr_decode_plan.forced_rows) instead of as a
    # full-sequence pass of its own; same losses and gradients up to summation order
    forced_rows_in_sweep: bool = True
    # not a reference flag: the style critic computes in fp32 even when the compute dtype is bf16
    style_critic_fp32: bool = True
    # not a reference flag: in the generator phase the critics' own parameters are held out of the graph (the reference
    # lets autograd accumulate gradients into them that the next train_discriminator zeroes unread,
    # onmt/trainer_wgan_wh.py:866-869 eval modes + :511-512 zero_grad): no weight-gradient products for them there
    critics_frozen_in_generator_phase: bool = True

    @property
    def d_cls(self):
        return self.cls[1] * self.cls[2]

    @property
    def d_sgen(self):
        return self.sgen[1] * self.sgen[2]

    @property
    def d_tdisc(self):
        return self.tdisc[1] * self.tdisc[2]

    @property
    def d_sdisc_hidden(self):
        return self.sdisc[1] * self.sdisc[2]

    @property
    def n_disc_out(self):
        return self.num_styles * (2 if self.faketruthstyle_mode else 1)

    def with_(self, **kw):
        return replace(self, **kw)

    @classmethod
    def small(cls, **kw):
        """Reduced shapes used by the parity fixtures (tests/golden)."""
        base = cls(vocab_size=200, layers=2, d_model=64, heads=4, d_ff=128,
                   cls=(2, 2, 16), sgen=(2, 2, 16), tdisc=(1, 2, 16), sdisc=(2, 4, 16),
                   dropout=0.0, attention_dropout=0.0)
        return replace(base, **kw)
