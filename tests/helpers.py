"""Shared comparison helpers for the parity tests."""
import torch


def rel_err(a, b):
    """max|a-b| / max(max|b|, 1e-12): the relative error the north_star tolerances are stated in."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def assert_close(a, b, tol, name="", floor=0.0):
    """|a-b|_inf <= tol * |b|_inf (+ floor for tensors that are ~0)."""
    a_ = a.detach().double().cpu()
    b_ = b.detach().double().cpu()
    assert a_.shape == b_.shape, "%s: shape %s vs %s" % (name, tuple(a_.shape), tuple(b_.shape))
    assert torch.isfinite(a_).all(), "%s: non-finite values" % name
    err = float((a_ - b_).abs().max()) if b_.numel() else 0.0
    ref = float(b_.abs().max()) if b_.numel() else 0.0
    assert err <= tol * ref + floor, "%s: max abs err %.3e vs ref max %.3e (rel %.3e > %.1e)" % (
        name, err, ref, err / max(ref, 1e-30), tol)


def assert_checksum(t, cs, tol, name=""):
    """Compare a tensor with a stored {sum, abssum, samples} checksum (tests/golden/make_golden.py)."""
    t = t.detach().double().cpu().reshape(-1)
    idx = torch.linspace(0, t.numel() - 1, 16).long()
    scale = cs["abssum"] / max(t.numel(), 1)
    assert abs(float(t.abs().sum()) - cs["abssum"]) <= tol * cs["abssum"] + 1e-12, name + " abssum"
    assert abs(float(t.sum()) - cs["sum"]) <= tol * cs["abssum"] + 1e-12, name + " sum"
    assert float((t[idx] - cs["samples"].double()).abs().max()) <= tol * max(
        float(cs["samples"].abs().max()), scale) + 1e-12, name + " samples"


def assert_grads(got, want, tol, name="", floor=0.0, floor_rel=1e-5):
    """Compare a dict of gradients.  Gradients that are mathematically zero (e.g. the key-bias
    gradient of softmax attention) hold only rounding noise, so errors below ``floor_rel`` x the
    largest gradient of the set are accepted."""
    scale = max([float(w.abs().max()) for w in want.values() if torch.is_tensor(w) and w.numel()] + [0.0])
    floor = max(floor, floor_rel * scale)
    for k, w in want.items():
        g = got.get(k)
        if w is None:
            assert g is None or float(g.abs().max()) == 0.0, "%s.%s should have no grad" % (name, k)
            continue
        assert g is not None, "%s.%s: missing grad" % (name, k)
        if isinstance(w, dict):
            assert_checksum(g, w, tol, "%s.%s" % (name, k))
        else:
            assert_close(g, w, tol, "%s.%s" % (name, k), floor=floor)


def assert_post_step(named_params, post, lr, name="", frac=0.9):
    """Post-Adam parameters of one network vs their stored checksums.  Adam turns a gradient g
    into a step lr*g/(|g|+eps): where g is mathematically zero (rounding noise only, e.g. the
    key-projection bias) the step's sign is arbitrary.  So every sampled element must lie within
    2*lr of the reference, and >= ``frac`` of all samples of the network must match closely."""
    n_close = n_all = 0
    for k, t in named_params:
        cs = post[k]
        t = t.detach().double().cpu().reshape(-1)
        idx = torch.linspace(0, t.numel() - 1, 16).long()
        ref = cs["samples"].double()
        err = (t[idx] - ref).abs()
        assert float(err.max()) <= 2.05 * lr + 1e-7, "%s.%s: post-step sample off by %.3e" % (
            name, k, float(err.max()))
        n_close += int((err <= 0.02 * lr + 2e-6 * ref.abs()).sum())
        n_all += err.numel()
    assert n_close >= frac * n_all, "%s: only %d of %d post-step samples match" % (name, n_close, n_all)
