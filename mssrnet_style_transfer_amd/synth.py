"""Synthetic Yelp-shaped batches (the inputs BASELINE.md section 3 prescribes).

Batch layout follows the reference data pipeline (onmt/inputters/inputter.py:682-773,
SURVEY.md section 2 row 13): ``src = (Long[L+1, B, 1], lengths+1)`` whose row 0 is the style
label token, ``tgt = Long[L+2, B, 1] = <s> w.. </s> <blank>..``; sentences sorted by length,
longest first.  Only torch CPU RNG is used so the same seed gives the same batch everywhere.
"""
from types import SimpleNamespace

import torch

UNK, PAD, BOS, EOS, SL0, SL1, DOT, EXCL, QST = range(9)
N_SPECIAL = 9
YELP_VOCAB = 9606


class Vocab(SimpleNamespace):
    """Token-id table with the special ids fixed as in BASELINE.md."""

    def __init__(self, size=YELP_VOCAB):
        super().__init__(size=size, unk=UNK, pad=PAD, bos=BOS, eos=EOS,
                         labels=(SL0, SL1), dot=DOT, excl=EXCL, qst=QST)

    def __len__(self):
        return self.size


def make_batch(batch_size=80, max_words=16, vocab_size=YELP_VOCAB, seed=123,
               min_words=3, device="cpu", lengths=None):
    """One batch. ``lengths`` ~ U{min_words..max_words-1} with one sentence forced to max_words.

    Returns a namespace with ``src``, ``src_lengths``, ``tgt``, ``labels`` (0/1),
    ``n_tokens`` (= tgt[1:] non-pad count, the reference's token normaliser,
    onmt/trainer_wgan_wh.py:389-391) and ``batch_size``.
    """
    g = torch.Generator().manual_seed(seed)
    if lengths is None:
        hi = max(min_words + 1, max_words)
        lens = torch.randint(min_words, hi, (batch_size,), generator=g)
        lens[0] = max_words
    else:
        lens = torch.as_tensor(lengths, dtype=torch.long)
        batch_size = lens.numel()
        max_words = int(lens.max())
    lens, _ = torch.sort(lens, descending=True)
    labels = torch.randint(0, 2, (batch_size,), generator=g)
    words = torch.randint(N_SPECIAL, vocab_size, (max_words, batch_size), generator=g)
    pos = torch.arange(max_words).unsqueeze(1)
    valid = pos < lens.unsqueeze(0)
    words = torch.where(valid, words, torch.full_like(words, PAD))

    src = torch.empty(max_words + 1, batch_size, dtype=torch.long)
    src[0] = labels + SL0
    src[1:] = words
    tgt = torch.full((max_words + 2, batch_size), PAD, dtype=torch.long)
    tgt[0] = BOS
    tgt[1:max_words + 1] = words
    tgt[lens + 1, torch.arange(batch_size)] = EOS
    n_tokens = int((tgt[1:] != PAD).sum())
    return SimpleNamespace(
        src=src.unsqueeze(2).to(device), src_lengths=(lens + 1).to(device),
        tgt=tgt.unsqueeze(2).to(device), labels=labels.to(device),
        n_tokens=n_tokens, batch_size=batch_size)
