"""Containers kept for state-dict compatibility (onmt/modules/util_class.py)."""
import torch
import torch.nn as nn


class Elementwise(nn.ModuleList):
    """List of per-feature look-up tables; only the single-feature ('first'/'concat' of one
    table) case occurs on the MSSRNet path.  Kept so that the embedding weight stays under the
    key ``make_embedding.emb_luts.0.weight``."""

    def __init__(self, merge=None, *args):
        assert merge in [None, "first", "concat", "sum", "mlp"]
        self.merge = merge
        super().__init__(*args)

    def forward(self, inputs):
        feats = [f.squeeze(2) for f in inputs.split(1, dim=2)]
        assert len(self) == len(feats)
        outs = [m(x) for m, x in zip(self, feats)]
        if self.merge == "first":
            return outs[0]
        if self.merge in ("concat", "mlp"):
            return torch.cat(outs, 2)
        if self.merge == "sum":
            return sum(outs)
        return outs


class Cast(nn.Module):
    """Casts its input to ``dtype`` (the generator's fp32 cast, onmt/model_builder.py:183)."""

    def __init__(self, dtype):
        super().__init__()
        self._dtype = dtype

    def forward(self, x):
        return x.to(self._dtype)
