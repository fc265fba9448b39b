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


def assert_mixed_grads(net, want, tol, name, outlier_frac=0.0, cap_mult=2.0, l2_tol=None):
    """Every parameter gradient against the reference's, errors measured against the network's largest gradient
    (helpers.assert_grads' convention).  fp32 (outlier_frac = 0): EVERY element within tol.  bf16: the relative L2
    error over the network <= tol, at most ``outlier_frac`` of the checked elements beyond tol and none beyond 2 tol
    (the maximum over 1e4..1e7 elements of 8-bit-mantissa noise sits at 4-5 sigma: at README widths the deepest
    tensor of the style generator, style_cls.weight, reaches 2.7e-2 with a network L2 error of 7e-3).
    Returns (worst max-abs ratio, relative L2 error, share of elements beyond tol)."""
    top = want["_max"]
    worst, e2, w2, n_out, n_all = 0.0, 0.0, 0.0, 0, 0
    # ``net``: a module (its parameters' .grad) or a {name: gradient} snapshot
    pairs = [(k, p.grad) for k, p in net.named_parameters()] if hasattr(net, "named_parameters") else list(net.items())
    for k, grad in pairs:
        w = want[k]
        if w is None:
            assert grad is None or float(grad.abs().max()) == 0.0, "%s.%s should have no gradient" % (name, k)
            continue
        assert grad is not None, "%s.%s: missing gradient" % (name, k)
        g = grad.detach().double().cpu()
        assert torch.isfinite(g).all(), "%s.%s: non-finite gradient" % (name, k)
        if isinstance(w, dict):
            t = g.reshape(-1)
            assert t.numel() == w["numel"]
            idx = torch.linspace(0, t.numel() - 1, w["samples"].numel()).long()
            d = t[idx] - w["samples"].double()
            ref = w["samples"].double()
            slack = tol * max(w["abssum"], t.numel() * top * 1e-3)
            assert abs(float(t.abs().sum()) - w["abssum"]) <= slack, "%s.%s: |g| sum %.6g vs %.6g" % (
                name, k, float(t.abs().sum()), w["abssum"])
            assert abs(float(t.sum()) - w["sum"]) <= slack, "%s.%s: g sum" % (name, k)
        else:
            d = g - w.double()
            ref = w.double()
        err = float(d.abs().max())
        e2 += float((d * d).sum())
        w2 += float((ref * ref).sum())
        n_out += int((d.abs() > tol * top).sum())
        n_all += d.numel()
        worst = max(worst, err / top)
        cap = tol if outlier_frac == 0.0 else cap_mult * tol
        assert err <= cap * top, "%s.%s: gradient off by %.3e = %.2e of the network's largest gradient %.3e (cap %.0e)" % (
            name, k, err, err / top, top, cap)
    l2 = (e2 / max(w2, 1e-300)) ** 0.5
    assert n_out <= outlier_frac * n_all, "%s: %d of %d checked gradient elements beyond %.0e of the largest gradient" % (
        name, n_out, n_all, tol)
    l2_tol = tol if l2_tol is None else l2_tol
    assert l2 <= l2_tol, "%s: relative L2 gradient error %.3e > %.0e" % (name, l2, l2_tol)
    return worst, l2, n_out / max(n_all, 1)


def assert_scalars(log, want, tol, name):
    """Logged losses.  `*_Wgap` / `*_cost` are differences and sums of the gains and the penalty logged beside them:
    their error is measured against the largest of those operands; accuracies are counts of argmax agreements."""
    for k, v in want.items():
        if k.endswith("_acc"):
            continue
        scale = abs(v)
        if k.endswith("_Wgap") or k.endswith("_cost"):
            p = k.split("_")[0]
            scale = max(abs(want[p + "R_gain"]), abs(want[p + "F_gain"]), abs(want[p + "_gp"]) if k.endswith("_cost") else 0.0)
        assert abs(log[k] - v) <= tol * max(scale, 1e-3), "%s.%s: %.6g vs %.6g (tol %.0e of %.3g)" % (name, k, log[k], v, tol, scale)


def wgan_step_with_capture(tr, batch, nets, critics=("text_discriminator", "style_discriminator")):
    """Trainer.train_wgan_step (what bench.py times) -> (d_log, g_log, critic gradients of the D phase).  The G phase's
    backward adds to the critics' .grad (as in the reference, which zeroes them at the start of the next D phase), so
    the D phase's gradients are cloned at the moment the step enters its generator half."""
    snap = {}
    inner = tr.train_generator

    def spy(b, _shared=None):
        for n in critics:
            snap[n] = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in nets[n].named_parameters()}
        return inner(b, _shared=_shared)

    tr.train_generator = spy
    try:
        d_log, g_log = tr.train_wgan_step(batch)
    finally:
        del tr.train_generator
    assert snap, "train_wgan_step did not reach its generator half"
    return d_log, g_log, snap
