"""Helpers with the semantics of onmt/utils/misc.py that the hot path uses."""
import torch


def sequence_mask(lengths, max_len=None):
    """Bool[B, max_len], True on valid positions (onmt/utils/misc.py:45-56).  Without ``max_len``
    the longest length is used, which costs a device sync exactly as in the reference."""
    max_len = int(max_len if max_len is not None else lengths.max())
    return torch.arange(max_len, device=lengths.device).unsqueeze(0) < lengths.unsqueeze(1)


def aeq(*args):
    first = args[0]
    assert all(a == first for a in args[1:]), "Not all arguments have the same value: " + str(args)
