"""Deterministic parameter fill shared by make_golden.py and the tests.

Large-shape fixtures do not store their weights: both sides rebuild them from a seed with
this function (same image => same torch CPU RNG stream), so only inputs/outputs are kept.
"""
import torch


def seeded_state(template, seed):
    """Return {key: tensor} with the shapes of ``template`` (a state-dict); buffers named
    ``*.pe`` (the sinusoid table) are kept as they are."""
    out = {}
    for i, k in enumerate(sorted(template.keys())):
        t = template[k]
        if k.endswith(".pe") or k == "pe":
            out[k] = t.clone()
            continue
        g = torch.Generator().manual_seed(seed * 1000 + i)
        r = torch.randn(t.shape, generator=g)
        if t.dim() > 1:
            r = r / (t.shape[-1] ** 0.5)
        elif k.endswith("weight"):
            r = 1.0 + 0.3 * r
        else:
            r = 0.3 * r
        out[k] = r
    return out
