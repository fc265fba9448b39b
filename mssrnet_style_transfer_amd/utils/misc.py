"""Helpers with the semantics of onmt/utils/misc.py that the hot path uses."""
import torch


_positions = {}


def _arange(n, device):
    """[1, n] position indices, kept per (n, device): a training step asks for the same few lengths ~30 times."""
    key = (n, device)
    t = _positions.get(key)
    if t is None:
        if len(_positions) > 256:
            _positions.clear()
        t = _positions[key] = torch.arange(n, device=device).unsqueeze(0)
    return t


def sequence_mask(lengths, max_len=None):
    """Bool[B, max_len], True on valid positions (onmt/utils/misc.py:45-56).  Without ``max_len``
    the longest length is used, which costs a device sync exactly as in the reference."""
    max_len = int(max_len if max_len is not None else lengths.max())
    return _arange(max_len, lengths.device) < lengths.unsqueeze(1)


def padding_mask(lengths, max_len):
    """~sequence_mask(lengths, max_len) in one launch: Bool[B, max_len], True on padding positions."""
    return _arange(int(max_len), lengths.device) >= lengths.unsqueeze(1)


def aeq(*args):
    first = args[0]
    assert all(a == first for a in args[1:]), "Not all arguments have the same value: " + str(args)
