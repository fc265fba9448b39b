"""Greedy style-transfer decoding (trans_beam_size 1).

Counterpart of onmt/translate/translator_wh.py: ``_run_encoder`` (:576-605: StyleGenerator2 +
TransformerEncoder + BondLayer) and the step loop of ``_translate_batch_with_strategy`` (:665-756)
with ``GreedySearch`` (onmt/translate/greedy_search.py:59-190).  Text I/O (fields, TranslationBuilder,
post_handle_sent) is outside the path: the result is token ids per sentence.
"""
import torch

from .decode_engine import greedy_decode_ids


def greedy_translate(model, style_generator, cfg, src, src_lengths, desired_styles, max_length=100):
    """src Long [L, B, 1], src_lengths Long [B], desired_styles Long [B] -> list of B token-id lists, each
    ending at its first </s> (inclusive) or cut at ``max_length`` (GreedySearch.update_finished semantics)."""
    model.eval()
    style_generator.eval()
    with torch.no_grad():
        rep = style_generator(desired_styles, src, src_lengths)
        enc_state, memory_bank, lengths = model.encoder(src, src_lengths)
        memory_bank = model.bond_layer(memory_bank, rep)
        model.decoder.init_state(src, memory_bank, enc_state)
        ids = greedy_decode_ids(model, cfg, memory_bank, src_lengths, max_length)
    rows = ids.t().tolist()
    out = []
    for sent in rows:
        cut = sent.index(cfg.eos_idx) + 1 if cfg.eos_idx in sent else len(sent)
        out.append(sent[:cut])
    return out
