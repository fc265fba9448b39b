"""DecoderBase (onmt/decoders/decoder.py:11-29)."""
import torch.nn as nn


class DecoderBase(nn.Module):
    def __init__(self, attentional=True):
        super().__init__()
        self.attentional = attentional

    @classmethod
    def from_opt(cls, opt, embeddings):
        raise NotImplementedError
