"""EncoderBase (onmt/encoders/encoder.py:8-58): the interface checks the callers rely on."""
import torch.nn as nn

from ..utils.misc import aeq


class EncoderBase(nn.Module):
    @classmethod
    def from_opt(cls, opt, embeddings=None):
        raise NotImplementedError

    def _check_args(self, src, lengths=None, hidden=None):
        n_batch = src.size(1)
        if lengths is not None:
            n_batch_, = lengths.size()
            aeq(n_batch, n_batch_)

    def forward(self, src, lengths=None):
        raise NotImplementedError
