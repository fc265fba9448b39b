"""Differentiable step-wise soft decoding used inside the training step.

Counterpart of onmt/train_and_translate.py: ``decode_batch_ori`` (:202-231),
``_decode_and_generate`` (:95-116) and ``fetch_dec_output_length`` (:289-326).  The decoder is
driven one step at a time with the previous step's probabilities as (soft) input, through the
module-level cache protocol of TransformerDecoder.
"""
import torch

from . import ops
from .decode_engine import soft_decode


def convert_to_onehot(labels, vocab_size, device, dtype=torch.float32):
    """onmt/tensor_utils.py:14-23."""
    return torch.zeros(labels.size(0), vocab_size, device=device, dtype=dtype).scatter_(1, labels.unsqueeze(1), 1)


def generator_probs(model, dec_out, temperature=1.0, log=False):
    """generator[0] Linear -> fp32 -> softmax(/T) (or log-softmax): train_and_translate.py:110-115,
    trainer_wgan_wh.py:659-664."""
    lin = model.generator[0]
    logits = ops.linear(dec_out, lin.weight, lin.bias)
    if log:
        return ops.log_softmax(logits, 1.0 / temperature, out_dtype=torch.float32)
    return ops.softmax(logits, None, 1.0 / temperature, out_dtype=torch.float32)


def fetch_dec_output_length(vocab_probs, cfg):
    """Host-side length rule: first </s> (or . ! ? when sent_stop_aware) + 1, 'cut <= 1' and the
    longest sentence forced to seq_len (train_and_translate.py:289-326)."""
    seq_len = vocab_probs.size(0)
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


def dec_output_length(vocab_probs, cfg):
    """fetch_dec_output_length on the device: (lengths Long[B], argmax ids [steps, B]) with no host round trip --
    first stop token + 1 (or seq_len), 'cut <= 1 -> seq_len', and the FIRST longest sentence forced to seq_len
    (the reference's `if cut > best` scan keeps the first maximum, as torch.argmax does)."""
    seq_len = vocab_probs.size(0)
    ids = torch.argmax(vocab_probs, 2)
    stop = ids == cfg.eos_idx
    if cfg.sent_stop_aware:
        stop = stop | (ids == cfg.dot_idx) | (ids == cfg.excl_idx) | (ids == cfg.qst_idx)
    pos = torch.arange(1, seq_len + 1, device=ids.device).unsqueeze(1)
    cut = torch.where(stop, pos, torch.full_like(pos, seq_len)).min(0).values
    cut = torch.where(cut <= 1, torch.full_like(cut, seq_len), cut)
    cut.scatter_(0, cut.argmax().view(1), seq_len)
    return cut, ids


def decode_batch_ori(model, cfg, src, enc_states, memory_bank, src_lengths, max_dec_steps=None):
    """Returns (vocab_probs [steps,B,V] fp32, seq_lens list, pred_ids [steps,B]).  The whole
    step loop runs inside the decode engine (decode_engine.soft_decode: one autograd node, one C
    call per direction); ``decode_batch_modular`` below is the same loop over the module-level
    cache protocol, kept as the drop-in surface and as the engine's cross-check."""
    if max_dec_steps is None:
        max_dec_steps = cfg.max_dec_steps
    model.decoder.init_state(src, memory_bank, enc_states)
    vocab_probs = soft_decode(model, cfg, memory_bank, src_lengths, max_dec_steps)
    seq_lens, pred_ids = fetch_dec_output_length(vocab_probs, cfg)
    return vocab_probs, seq_lens, pred_ids


def decode_batch_modular(model, cfg, src, enc_states, memory_bank, src_lengths, max_dec_steps=None):
    """decode_batch_ori through TransformerDecoder.forward(step=s) and the layer_cache dicts."""
    if max_dec_steps is None:
        max_dec_steps = cfg.max_dec_steps
    B = src_lengths.size(0)
    model.decoder.init_state(src, memory_bank, enc_states)
    bos = torch.full((B,), cfg.bos_idx, dtype=torch.long, device=src.device)
    decoder_input = convert_to_onehot(bos, cfg.vocab_size, src.device).unsqueeze(0)
    outs = []
    for step in range(max_dec_steps):
        dec_out, _ = model.decoder(decoder_input, memory_bank, memory_lengths=src_lengths, step=step)
        probs = generator_probs(model, dec_out, cfg.dec_temperature)
        outs.append(probs[-1:])
        decoder_input = outs[-1]
    vocab_probs = torch.cat(outs, 0)
    seq_lens, pred_ids = fetch_dec_output_length(vocab_probs, cfg)
    return vocab_probs, seq_lens, pred_ids
