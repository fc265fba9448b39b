"""HIP kernels (through the C ABI / ctypes) against the oracle and the reference's golden vectors.

Tolerances (north_star): fp32 relative 1e-4, bf16 relative 2e-2.  GPU only (-m gpu).
"""
import math

import pytest
import torch

from helpers import assert_close, assert_grads
from _seeded import seeded_state

pytestmark = pytest.mark.gpu

TOL32 = 1e-4
TOL16 = 2e-2


@pytest.fixture(scope="module")
def ops():
    from mssrnet_style_transfer_amd import ops as o
    o.set_compute_dtype(torch.float32)
    yield o
    o.set_compute_dtype(torch.float32)


def dev(x):
    return x.cuda() if torch.is_tensor(x) else x


def gp_probe(g):
    return ((g.reshape(-1, g.size(-1)).norm(2, dim=1) - 1.0) ** 2).mean()


def test_library_loads_and_reports_errors():
    from mssrnet_style_transfer_amd import _lib
    lib = _lib.load()
    assert lib.mssr_abi_version() == 2
    x = torch.zeros(4, 6, device="cuda")     # d = 6 is not a multiple of 4 -> rejected, nothing launched
    rc = lib.mssr_layernorm_fwd(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                4, 6, 1e-6, 0, None)
    assert rc != 0 and b"multiple of 4" in lib.mssr_last_error()


@pytest.mark.parametrize("d", [64, 256, 512])
def test_layernorm_golden(ops, golden_ops, d):
    c = golden_ops["ln_%d" % d]
    x = dev(c["x"]).requires_grad_()
    g = dev(c["gamma"]).requires_grad_()
    b = dev(c["beta"]).requires_grad_()
    y = ops.layer_norm(x, g, b, 1e-6)
    assert_close(y, c["y"], TOL32, "y")
    gx, gg, gb = torch.autograd.grad((y * dev(c["w"])).sum(), (x, g, b), create_graph=True)
    assert_close(gx, c["gx"], TOL32, "gx")
    assert_close(gg, c["ggamma"], TOL32, "ggamma")
    assert_close(gb, c["gbeta"], TOL32, "gbeta")
    probe = gp_probe(gx)
    assert_close(probe, c["probe"], TOL32, "probe")
    g2x, g2g = torch.autograd.grad(probe, (x, g))
    assert_close(g2x, c["g2x"], TOL32, "g2x")
    assert_close(g2g, c["g2gamma"], TOL32, "g2gamma")


def test_layernorm_bf16_and_dgamma_path(ops):
    torch.manual_seed(0)
    x = torch.randn(1000, 512) * 2 + 0.5
    g, b = torch.randn(512) * 0.3 + 1, torch.randn(512) * 0.3
    w = torch.randn(1000, 512)
    xr, gr, br = x.clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    yr = torch.nn.functional.layer_norm(xr, (512,), gr, br, 1e-6)
    (yr * w).sum().backward()
    ops.set_compute_dtype(torch.bfloat16)
    try:
        xc, gc, bc = x.cuda().requires_grad_(), g.cuda().requires_grad_(), b.cuda().requires_grad_()
        y = ops.layer_norm(xc, gc, bc, 1e-6)
        assert y.dtype == torch.bfloat16
        (y.float() * w.cuda()).sum().backward()
        assert_close(y.float(), yr, TOL16, "y bf16")
        assert_close(xc.grad, xr.grad, TOL16, "dx bf16")
        assert_close(gc.grad, gr.grad, TOL16, "dgamma bf16")
        assert_close(bc.grad, br.grad, TOL16, "dbeta bf16")
    finally:
        ops.set_compute_dtype(torch.float32)
    # second-order terms through dgamma / dbeta (ggg, ggb inputs of the bwd_bwd kernel), fp32 vs torch
    x2, g2, b2 = x[:50].clone().requires_grad_(), g.clone().requires_grad_(), b.clone().requires_grad_()
    w2 = w[:50]
    y2 = torch.nn.functional.layer_norm(x2, (512,), g2, b2, 1e-6)
    a1 = torch.autograd.grad((y2 * w2).sum(), (x2, g2, b2), create_graph=True)
    s_ref = (a1[0] ** 2).sum() + (a1[1] ** 3).sum() + (a1[2] ** 2).sum()
    r_ref = torch.autograd.grad(s_ref, (x2, g2))
    xc, gc, bc = x[:50].cuda().requires_grad_(), g.cuda().requires_grad_(), b.cuda().requires_grad_()
    y = ops.layer_norm(xc, gc, bc, 1e-6)
    a2 = torch.autograd.grad((y * w2.cuda()).sum(), (xc, gc, bc), create_graph=True)
    s = (a2[0] ** 2).sum() + (a2[1] ** 3).sum() + (a2[2] ** 2).sum()
    r = torch.autograd.grad(s, (xc, gc))
    assert_close(r[0], r_ref[0], TOL32, "g_x via all three")
    assert_close(r[1], r_ref[1], TOL32, "g_gamma via all three")


GEMM_SHAPES = [
    # (batch dims, M, N, K, a transposed view, b given as [N,K])
    ((), 80, 512, 512, False, True),
    ((), 1440, 1024, 512, False, True),
    ((), 80, 9606, 512, False, True),          # generator Linear, odd N
    ((), 1440, 256, 9606, False, False),       # soft embedding: K = V, unaligned rows
    ((), 512, 1024, 1440, True, False),        # dW = dY^T X  (A transposed view)
    ((), 37, 129, 65, False, True),            # ragged everything
    ((), 1, 5, 7, False, False),
    ((5, 8), 18, 18, 64, False, True),         # scores Q K^T per (b, h)
    ((5, 8), 18, 64, 18, False, False),        # context P V
    ((3, 4), 130, 130, 32, True, True),
]


@pytest.mark.parametrize("shape", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_shapes(ops, shape, dtype):
    bd, M, N, K, ta, tb = shape
    torch.manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(*bd, K, M).transpose(-1, -2) if ta else torch.randn(*bd, M, K)
    b = torch.randn(*bd, N, K) if tb else torch.randn(*bd, K, N)
    a, b = a.to(dtype), b.to(dtype)
    ref = a.double() @ (b.double().transpose(-1, -2) if tb else b.double())
    ac = (a.transpose(-1, -2).contiguous().cuda().transpose(-1, -2) if ta else a.cuda())
    out = ops.gemm(ac, b.cuda(), tb, dtype)
    assert_close(out.float(), ref, 1e-5 if dtype == torch.float32 else 1e-2, "gemm %s" % (shape,))


def test_gemm_epilogue_and_heads_layout(ops):
    torch.manual_seed(3)
    M, N, K = 100, 72, 40
    a, w, bias = torch.randn(M, K), torch.randn(N, K), torch.randn(N)
    res, gate = torch.randn(M, N), torch.randn(M, N)
    ref = torch.relu(a @ w.t() + bias)
    out = ops.gemm(a.cuda(), w.cuda(), True, torch.float32, bias=bias.cuda(), relu=True)
    assert_close(out, ref, 1e-5)
    ref = (a @ w.t() + bias + res) * (gate > 0)
    out = ops.gemm(a.cuda(), w.cuda(), True, torch.float32, bias=bias.cuda(), residual=res.cuda(),
                   gate=gate.cuda())
    assert_close(out, ref, 1e-5)
    c0 = torch.randn(M, N)
    out = ops.gemm(a.cuda(), w.cuda(), True, torch.float32, alpha=0.5, beta=1.0, out=c0.cuda())
    assert_close(out, 0.5 * (a @ w.t()) + c0, 1e-5)
    # attention layout: operands are [B, L, H, dh] memory viewed [B, H, L, dh]; context stored heads-last
    B, H, L, dh = 3, 4, 9, 16
    q = torch.randn(B, L, H * dh).cuda()
    k = torch.randn(B, L, H * dh).cuda()
    qv, kv = (t.view(B, L, H, dh).transpose(1, 2) for t in (q, k))
    s = ops.matmul(qv, kv, tb=True)
    assert_close(s, torch.matmul(qv, kv.transpose(2, 3)), 1e-5)
    ctx = ops.matmul(s, kv, tb=False, heads_last=True)
    assert ctx.permute(0, 2, 1, 3).is_contiguous()
    assert_close(ctx, torch.matmul(s, kv), 1e-5)


def test_softmax_family(ops):
    torch.manual_seed(5)
    for n, rows in ((7, 33), (130, 20), (1000, 9), (9606, 11)):
        x = torch.randn(rows, n) * 3
        xr = x.clone().requires_grad_()
        w = torch.randn(rows, n)
        yr = torch.softmax(xr * 0.7, -1)
        gr, = torch.autograd.grad((yr * w).sum(), xr, create_graph=True)
        g2r, = torch.autograd.grad((gr ** 2).sum(), xr)
        xc = x.cuda().requires_grad_()
        y = ops.softmax(xc, None, 0.7)
        g, = torch.autograd.grad((y * w.cuda()).sum(), xc, create_graph=True)
        g2, = torch.autograd.grad((g ** 2).sum(), xc)
        assert_close(y, yr, TOL32, "softmax n=%d" % n)
        assert_close(g, gr, TOL32, "softmax bwd n=%d" % n)
        assert_close(g2, g2r, TOL32, "softmax bwd_bwd n=%d" % n)
        ly = ops.log_softmax(xc, 1.3)
        lr = torch.log_softmax(xr * 1.3, -1)
        assert_close(ly, lr, TOL32, "log_softmax")
        gl, = torch.autograd.grad((ly * w.cuda()).sum(), xc)
        glr, = torch.autograd.grad((lr * w).sum(), xr)
        assert_close(gl, glr, TOL32, "log_softmax bwd")
    # masks: [B,1,Lk] and [B,Lq,Lk]; a fully masked row gives a uniform distribution (finite fill)
    B, H, Lq, Lk = 3, 2, 5, 6
    s = torch.randn(B, H, Lq, Lk)
    m1 = torch.zeros(B, 1, Lk, dtype=torch.bool)
    m1[1, 0, 4:] = True
    m1[2, 0, :] = True
    m2 = torch.rand(B, Lq, Lk) > 0.6
    for m in (m1, m2):
        ref = torch.softmax(s.masked_fill(m.unsqueeze(1), -1e18), -1)
        got = ops.softmax(s.cuda(), m.cuda(), 1.0)
        assert_close(got, ref, TOL32, "masked softmax")
    ops.set_compute_dtype(torch.bfloat16)
    try:
        y = ops.softmax(s.cuda().bfloat16(), m1.cuda(), 1.0)
        assert y.dtype == torch.bfloat16
        assert_close(y.float(), torch.softmax(s.bfloat16().float().masked_fill(m1.unsqueeze(1), -1e18), -1), TOL16)
    finally:
        ops.set_compute_dtype(torch.float32)


def test_dropout_mask_replay_and_stats(ops):
    torch.manual_seed(1)
    x = torch.randn(1 << 20, device="cuda")
    x = torch.where(x == 0, torch.ones_like(x), x).requires_grad_()     # randn can emit an exact 0
    r = torch.randn(1 << 20, device="cuda")
    y = ops.dropout(x, 0.1, True)
    sel = y.detach() != 0
    assert abs(sel.float().mean().item() - 0.9) < 5e-3
    g, = torch.autograd.grad(y.sum(), x, create_graph=True)
    assert_close(g, sel.float() / 0.9, 1e-6)
    assert_close(y[sel], x.detach()[sel] / 0.9, 1e-6)
    assert not g.requires_grad          # dropout is linear in x: its gradient does not depend on x
    yr = ops.dropout(x, 0.1, True, residual=r).detach()
    dropped = yr == r
    kept_ok = (yr - (x.detach() / 0.9 + r)).abs() <= 1e-6 * (1 + r.abs() + x.detach().abs())
    assert bool((dropped | kept_ok).all()) and abs(dropped.float().mean().item() - 0.1) < 5e-3
    y2 = ops.dropout(x, 0.1, True)
    assert not torch.equal((y2 != 0), sel)               # a fresh mask per call
    assert torch.equal(ops.dropout(x, 0.1, False), x)    # eval: identity
    xo = torch.randn(1003, device="cuda")                # odd length tail
    yo = ops.dropout(xo, 0.5, True)
    nz = yo != 0
    assert_close(yo[nz], xo[nz] * 2, 1e-6)


def test_embedding_and_pe(ops, golden_ops):
    from mssrnet_style_transfer_amd.modules.embeddings import Embeddings
    e = golden_ops["emb"]
    emb = Embeddings(64, 50, 1, position_encoding=True, dropout=0.0)
    emb.load_state_dict(e["sd"])
    emb.cuda()
    W = emb.word_lut.weight
    y = emb(dev(e["ids"]))
    assert_close(y, e["y_ids"], TOL32)
    g, = torch.autograd.grad((y * dev(e["w"])).sum(), W)
    assert_close(g, e["g_ids_w"], TOL32)
    soft = dev(e["soft"]).requires_grad_()
    y = emb(soft)
    assert_close(y, e["y_soft"], TOL32)
    gi, gw = torch.autograd.grad((y * dev(e["w"])).sum(), (soft, W))
    assert_close(gi, e["g_soft_in"], TOL32)
    assert_close(gw, e["g_soft_w"], TOL32)
    assert_close(emb(soft[2:3], step=2), e["y_soft_step2"], TOL32)
    c = golden_ops["pe"]
    pe = emb.make_embedding[1]
    assert_close(pe(dev(c["emb"])), c["full"], TOL32)
    assert_close(pe(dev(c["emb"])[:1], step=3), c["step3"], TOL32)


def mha_template(d):
    t = {}
    for n in ("linear_keys", "linear_values", "linear_query", "final_linear"):
        t[n + ".weight"] = torch.empty(d, d)
        t[n + ".bias"] = torch.empty(d)
    return t


def run_mha(c, dtype=torch.float32):
    from mssrnet_style_transfer_amd.modules.multi_headed_attn import MultiHeadedAttention
    d = c["d"]
    m = MultiHeadedAttention(c["H"], d, dropout=0.0)
    m.load_state_dict(c["sd"] if c["sd"] is not None else seeded_state(mha_template(d), c["seed"]))
    m.cuda()
    ins = {k: dev(v).requires_grad_() for k, v in c["inputs"].items()}
    if "x" in ins:
        key = value = query = ins["x"]
        at = "self"
    else:
        key = value = ins["mem"]
        query = ins["q"]
        at = "context"
    o, attn = m(key, value, query, mask=dev(c["mask"]), attn_type=at)
    return m, ins, o, attn


@pytest.mark.parametrize("name", ["self_pad", "self_causal", "cross", "nomask", "self_pad_d256", "cross_d512"])
def test_mha_golden_fp32(ops, golden_ops, name):
    """Composed path (ops.second_order(): what the WGAN-GP critics run): values, gradients through both
    outputs and the double-backward probe against the reference's vectors."""
    c = golden_ops["mha_" + name]
    with ops.second_order():
        m, ins, o, attn = run_mha(c)
    assert_close(o, c["out"], TOL32, "out")
    assert_close(attn, c["attn"], TOL32, "attn")
    names = list(ins.keys()) + [n for n, _ in m.named_parameters()]
    tens = list(ins.values()) + list(m.parameters())
    scalar = (o * dev(c["wo"])).sum() + (attn * dev(c["wa"])).sum()
    g = torch.autograd.grad(scalar, tens, create_graph=True)
    assert_grads(dict(zip(names, g)), c["g"], TOL32, "g")
    probe = sum(gp_probe(gi) for gi in g[:len(ins)])
    assert_close(probe, c["probe"], TOL32, "probe")
    g2 = torch.autograd.grad(probe, tens, allow_unused=True)
    assert_grads(dict(zip(names, g2)), c["g2"], 2e-4, "g2")


def test_mha_bf16(ops, golden_ops):
    ops.set_compute_dtype(torch.bfloat16)
    try:
        for name in ("self_pad", "cross_d512"):
            c = golden_ops["mha_" + name]
            with ops.second_order():
                m, ins, o, attn = run_mha(c)
            assert o.dtype == torch.bfloat16
            assert_close(o.float(), c["out"], TOL16, "out bf16")
            assert_close(attn.float(), c["attn"], TOL16, "attn bf16")
            tens = list(ins.values()) + list(m.parameters())
            names = list(ins.keys()) + [n for n, _ in m.named_parameters()]
            scalar = (o.float() * dev(c["wo"])).sum() + (attn.float() * dev(c["wa"])).sum()
            g = torch.autograd.grad(scalar, tens)
            for n_, t in zip(names, g):
                assert t.dtype == torch.float32
            assert_grads(dict(zip(names, g)), c["g"], 4e-2, "g bf16", floor_rel=2e-2)
    finally:
        ops.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("name", ["self_pad", "self_causal", "cross", "nomask", "self_pad_d256", "cross_d512"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mha_fused_kernel(ops, golden_ops, name, dtype):
    """The fused attention kernel (csrc/attention.hip, the default path) against the reference's vectors
    (outputs, probabilities) and against the composed path's first-order gradients."""
    c = golden_ops["mha_" + name]
    tol, gtol = (TOL32, 2e-4) if dtype == torch.float32 else (TOL16, 4e-2)
    ops.set_compute_dtype(dtype)
    try:
        m, ins, o, attn = run_mha(c)
        assert o.grad_fn is not None and not attn.requires_grad          # fused: probabilities carry no graph
        assert_close(o.float(), c["out"], tol, "out")
        assert_close(attn.float(), c["attn"], tol, "attn")
        tens = list(ins.values()) + list(m.parameters())
        names = list(ins.keys()) + [n for n, _ in m.named_parameters()]
        g = torch.autograd.grad((o.float() * dev(c["wo"])).sum(), tens)
        with ops.second_order():
            m2, ins2, o2, _ = run_mha(c)
        g2 = torch.autograd.grad((o2.float() * dev(c["wo"])).sum(), list(ins2.values()) + list(m2.parameters()))
        assert_grads(dict(zip(names, g)), dict(zip(names, g2)), gtol, "fused vs composed",
                     floor_rel=1e-5 if dtype == torch.float32 else 2e-2)
    finally:
        ops.set_compute_dtype(torch.float32)


def test_fused_attention_dropout_and_limits(ops):
    """Dropout inside the fused kernel: the backward regenerates the forward's mask (directional finite
    difference under a fixed seed); shapes beyond the kernel's limits take the composed path."""
    torch.manual_seed(0)
    B, H, L, dh = 3, 4, 9, 16
    q, k, v = (torch.randn(B, L, H * dh, device="cuda").view(B, L, H, dh).transpose(1, 2) for _ in range(3))
    w = torch.randn(B, H, L, dh, device="cuda")
    d = torch.randn_like(q)

    def f(qq):
        ops._state["seed_counter"] = 77
        out, _ = ops.attention(qq, k, v, None, 0.4, True)
        return out

    qg = q.clone().requires_grad_(True)
    out = f(qg)
    (out * w).sum().backward()
    with torch.no_grad():
        assert torch.equal(f(q), out.detach())
        assert float((f(q) - ops.attention(q, k, v, None, 0.4, False)[0]).abs().max()) > 1e-4
        eps = 1e-2
        fd = float(((f(q + eps * d) - f(q - eps * d)) * w).sum() / (2 * eps))
    an = float((qg.grad * d).sum())
    assert abs(an - fd) <= 2e-2 * max(abs(fd), 1e-3), (an, fd)
    from mssrnet_style_transfer_amd import _lib
    lib = _lib.load()
    assert lib.mssr_attention_fits(17, 17, 64, _lib.BF16, 1) == 1
    assert lib.mssr_attention_fits(130, 130, 64, _lib.BF16, 1) == 1          # matrix-core kernels (attention_mfma.hip)
    assert lib.mssr_attention_fits(130, 130, 64, _lib.F32, 0) == 0
    assert lib.mssr_attention_fits(161, 161, 64, _lib.BF16, 0) == 0
    big = torch.randn(2, 2, 70, 16, device="cuda")
    o_big, p_big = ops.attention(big, big, big, None, 0.0, False)          # composed path, same contract
    ref = torch.softmax(big @ big.transpose(-1, -2) / 4.0, -1) @ big
    assert_close(o_big, ref, 1e-4, "beyond the fused limits")


def _attn_ref(q, k, v, mask, keep=None):
    """fp64 reference of the attention chain on the bf16-rounded operands; keep = dropout multipliers or None."""
    qr, kr, vr = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    sc = (qr / q.shape[-1] ** 0.5) @ kr.transpose(-1, -2)
    if mask is not None:
        sc = sc.masked_fill(mask.unsqueeze(1), -1e18)
    pr = torch.softmax(sc, -1)
    pd = pr if keep is None else pr * keep.double()
    return (qr, kr, vr), pr, pd @ vr


@pytest.mark.parametrize("B,H,Lq,Lk,dh,kind", [
    (2, 2, 16, 16, 64, "pad"), (3, 2, 17, 17, 64, "pad"), (2, 4, 33, 50, 32, "none"), (2, 8, 130, 130, 64, "pad"),
    (2, 2, 160, 160, 64, "causal"), (2, 3, 100, 37, 64, "pad"), (3, 2, 1, 140, 32, "pad"), (2, 2, 75, 129, 64, "full"),
    (80, 8, 18, 18, 64, "pad")])
def test_mfma_attention_bf16(ops, B, H, Lq, Lk, dh, kind):
    """Matrix-core fused attention (csrc/attention_mfma.hip; bf16, up to 160 keys: the config-5 lengths) against fp64
    torch on the same bf16 operands: context, probabilities, dq / dk / dv; key-padding, causal and per-query masks,
    a fully masked sentence (uniform row, not NaN), ragged tile edges, one query row."""
    from mssrnet_style_transfer_amd import _lib
    assert _lib.load().mssr_attention_fits(Lq, Lk, dh, _lib.BF16, 1) == 1
    torch.manual_seed(Lq * 1000 + Lk)
    ops.set_compute_dtype(torch.bfloat16)
    try:
        q, k, v = (torch.randn(B, n, H * dh, device="cuda").bfloat16().view(B, n, H, dh).transpose(1, 2).detach()
                   .requires_grad_(True) for n in (Lq, Lk, Lk))
        mask = None
        if kind == "pad":
            lens = torch.randint(1, Lk + 1, (B,), device="cuda")
            lens[0] = Lk
            mask = (torch.arange(Lk, device="cuda").unsqueeze(0) >= lens.unsqueeze(1)).unsqueeze(1)
            mask[B - 1] = True                                                           # a fully masked sentence
        elif kind == "causal":
            mask = torch.triu(torch.ones(Lq, Lk, dtype=torch.bool, device="cuda"), 1).unsqueeze(0).repeat(B, 1, 1)
        elif kind == "full":
            mask = torch.rand(B, Lq, Lk, device="cuda") < 0.3
        w = torch.randn(B, H, Lq, dh, device="cuda").bfloat16()
        out, probs = ops.attention(q, k, v, mask, 0.0, False)
        assert out.dtype == torch.bfloat16 and probs.dtype == torch.bfloat16
        g = torch.autograd.grad((out * w).sum(), (q, k, v))
        leaves, pr, ref = _attn_ref(q, k, v, mask)
        gr = torch.autograd.grad((ref * w.double()).sum(), leaves)
        assert torch.isfinite(out.float()).all() and torch.isfinite(probs.float()).all()
        assert_close(out.float(), ref.float(), TOL16, "out")
        assert_close(probs.float(), pr.float(), TOL16, "probs")
        if kind == "pad":
            assert_close(probs[B - 1].float(), torch.full_like(probs[B - 1], 1.0 / Lk).float(), 1e-2, "masked row uniform")
        for name, a, b in zip("qkv", g, gr):
            assert_close(a.float(), b.float(), 3e-2, "d" + name, floor=1e-6)
    finally:
        ops.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("Lq,Lk", [(40, 48), (130, 130)])
def test_mfma_attention_dropout_replay(ops, Lq, Lk):
    """Dropout inside the matrix-core kernels: the keep decisions are read off the forward (one-hot values: the context
    IS the dropped probability row, 64 keys at a time under a fixed seed), then the backward of a random problem with
    the same seed must equal fp64 autograd with exactly that mask -- both backward phases regenerate it."""
    B, H, dh, p = 2, 3, 64, 0.3
    torch.manual_seed(5)
    ops.set_compute_dtype(torch.bfloat16)
    try:
        q, k, v = (torch.randn(B, n, H * dh, device="cuda").bfloat16().view(B, n, H, dh).transpose(1, 2).detach()
                   .requires_grad_(True) for n in (Lq, Lk, Lk))
        lens = torch.tensor([Lk, Lk - 5], device="cuda")
        mask = (torch.arange(Lk, device="cuda").unsqueeze(0) >= lens.unsqueeze(1)).unsqueeze(1)
        pd = torch.zeros(B, H, Lq, Lk, device="cuda")
        with torch.no_grad():
            for c0 in range(0, Lk, dh):
                n = min(dh, Lk - c0)
                onehot = torch.zeros(B, H, Lk, dh, device="cuda", dtype=torch.bfloat16)
                onehot[:, :, c0:c0 + n, :n] = torch.eye(n, device="cuda", dtype=torch.bfloat16)
                oh = onehot.transpose(1, 2).contiguous().transpose(1, 2)
                ops._state["seed_counter"] = 123
                o, probs = ops.attention(q.detach(), k.detach(), oh, mask, p, True)
                pd[..., c0:c0 + n] = o[..., :n].float()
        keep = (pd != 0).float() / (1.0 - p)
        live = probs.float() > 1e-4
        rate = float(((pd != 0) & live).sum()) / float(live.sum())
        assert abs(rate - (1 - p)) < 0.02, rate
        assert_close(pd, probs.float() * keep, TOL16, "dropped probabilities")
        w = torch.randn(B, H, Lq, dh, device="cuda").bfloat16()
        ops._state["seed_counter"] = 123
        out, _ = ops.attention(q, k, v, mask, p, True)
        g = torch.autograd.grad((out * w).sum(), (q, k, v))
        leaves, _, ref = _attn_ref(q, k, v, mask, keep)
        gr = torch.autograd.grad((ref * w.double()).sum(), leaves)
        assert_close(out.float(), ref.float(), TOL16, "out")
        for name, a, b in zip("qkv", g, gr):
            assert_close(a.float(), b.float(), 3e-2, "d" + name, floor=1e-6)
    finally:
        ops.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("B,H,Lq,Lk,dh", [(1, 1, 1, 1, 8), (2, 3, 1, 64, 16), (2, 2, 64, 64, 64), (3, 2, 5, 7, 32),
                                          (4, 8, 17, 17, 64)])
def test_fused_attention_edge_shapes_and_masked_rows(ops, B, H, Lq, Lk, dh):
    """Fused attention at the corners of its range (one key, one query, the 64 x 64 x 64 maximum, ragged pad masks)
    and with a sentence whose keys are ALL masked: the finite -1e18 fill makes that row a uniform distribution, not
    NaN (onmt/modules/multi_headed_attn.py:195-202).  Values, probabilities and gradients against plain torch."""
    torch.manual_seed(B * 1000 + Lk)
    q, k, v = (torch.randn(B, n, H * dh, device="cuda").view(B, n, H, dh).transpose(1, 2).detach().requires_grad_(True)
               for n in (Lq, Lk, Lk))
    lens = torch.randint(1, Lk + 1, (B,), device="cuda")
    lens[0] = Lk
    mask = torch.arange(Lk, device="cuda").unsqueeze(0) >= lens.unsqueeze(1)            # True = masked
    if B > 1:
        mask[B - 1] = True                                                              # a fully masked sentence
    mask3 = mask.unsqueeze(1)
    w = torch.randn(B, H, Lq, dh, device="cuda")
    out, probs = ops.attention(q, k, v, mask3, 0.0, False)
    g = torch.autograd.grad((out * w).sum(), (q, k, v))
    qr, kr, vr = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    sc = (qr / dh ** 0.5) @ kr.transpose(-1, -2)
    pr = torch.softmax(sc.masked_fill(mask3.unsqueeze(1), -1e18), -1)
    ref = pr @ vr
    gr = torch.autograd.grad((ref * w.double()).sum(), (qr, kr, vr))
    assert torch.isfinite(out).all() and torch.isfinite(probs).all()
    assert_close(out, ref.float(), 1e-4, "out")
    assert_close(probs, pr.float(), 1e-4, "probs")
    if B > 1:
        assert_close(probs[B - 1], torch.full_like(probs[B - 1], 1.0 / Lk), 1e-5, "fully masked row is uniform")
    for name, a, b in zip("qkv", g, gr):
        assert_close(a, b.float(), 2e-4, "d" + name, floor=1e-6)


def test_mha_cached(ops, golden_ops):
    from mssrnet_style_transfer_amd.modules.multi_headed_attn import MultiHeadedAttention
    c = golden_ops["mha_cached"]
    ms, mc = MultiHeadedAttention(c["H"], 64, 0.0), MultiHeadedAttention(c["H"], 64, 0.0)
    ms.load_state_dict(c["sd_self"])
    mc.load_state_dict(c["sd_ctx"])
    ms.cuda()
    mc.cuda()
    mem = dev(c["mem"])
    mask = ~(torch.arange(mem.size(1), device="cuda").unsqueeze(0) < dev(c["lens"]).unsqueeze(1)).unsqueeze(1)
    cache = {"self_keys": None, "self_values": None, "memory_keys": None, "memory_values": None}
    for st in c["steps"]:
        q = dev(st["q"])
        o1, a1 = ms(q, q, q, mask=None, layer_cache=cache, attn_type="self")
        o2, a2 = mc(mem, mem, q, mask=mask, layer_cache=cache, attn_type="context")
        assert_close(o1, st["self_out"], TOL32)
        assert_close(a1, st["self_attn"], TOL32)
        assert_close(o2, st["ctx_out"], TOL32)
        assert_close(a2, st["ctx_attn"], TOL32)
    for k, v in c["cache"].items():
        assert_close(cache[k], v, TOL32, k)


def ffn_template(d, ff):
    return {"w_1.weight": torch.empty(ff, d), "w_1.bias": torch.empty(ff), "w_2.weight": torch.empty(d, ff),
            "w_2.bias": torch.empty(d), "layer_norm.weight": torch.empty(d), "layer_norm.bias": torch.empty(d)}


@pytest.mark.parametrize("d", [64, 256])
def test_ffn_golden(ops, golden_ops, d):
    from mssrnet_style_transfer_amd.modules.position_ffn import PositionwiseFeedForward
    c = golden_ops["ffn_%d" % d]
    ff = c["ff"] if c["sd"] is None else c["sd"]["w_1.weight"].shape[0]
    m = PositionwiseFeedForward(d, ff, 0.0)
    m.load_state_dict(c["sd"] if c["sd"] is not None else seeded_state(ffn_template(d, ff), c["seed"]))
    m.cuda()
    x = dev(c["x"]).requires_grad_()
    y = m(x)
    assert_close(y, c["y"], TOL32, "y")
    names = ["x"] + [n for n, _ in m.named_parameters()]
    tens = [x] + list(m.parameters())
    g = torch.autograd.grad((y * dev(c["w"])).sum(), tens, create_graph=True)
    assert_grads(dict(zip(names, g)), c["g"], TOL32, "g")
    g2 = torch.autograd.grad(gp_probe(g[0]), tens, allow_unused=True)
    assert_grads(dict(zip(names, g2)), c["g2"], 2e-4, "g2")


def test_encoder_golden(ops, golden_modules):
    from mssrnet_style_transfer_amd.encoders.transformer import TransformerEncoder
    from mssrnet_style_transfer_amd.modules.embeddings import Embeddings
    c = golden_modules["encoder"]
    enc = TransformerEncoder(2, 64, 4, 128, 0.0, 0.0, Embeddings(64, 200, 1, position_encoding=True), 0)
    enc.load_state_dict(c["sd"])
    enc.cuda()
    emb, mem, _ = enc(dev(c["src"]), dev(c["lens"]))
    assert_close(emb, c["emb"], TOL32)
    assert_close(mem, c["mem"], TOL32)
    (mem * dev(c["w"])).sum().backward()
    assert_grads({k: p.grad for k, p in enc.named_parameters()}, c["g"], TOL32, "enc")
    soft = dev(c["soft"]).requires_grad_()
    _, mem2, _ = enc(soft, dev(c["lens"]))
    assert_close(mem2, c["mem_soft"], TOL32)
    gs, = torch.autograd.grad((mem2 * dev(c["w"])).sum(), soft)
    assert_close(gs, c["g_soft_in"], TOL32)


def test_decoder_golden(ops, golden_modules):
    from mssrnet_style_transfer_amd.decoders.transformer import TransformerDecoder
    from mssrnet_style_transfer_amd.modules.embeddings import Embeddings
    c = golden_modules["decoder"]
    dec = TransformerDecoder(2, 64, 4, 128, False, "scaled-dot", 0.0, 0.0, Embeddings(64, 200, 1, position_encoding=True),
                             0, False, False, -3, 0)
    dec.load_state_dict(c["sd"])
    dec.cuda()
    src, lens = dev(c["src"]), dev(c["lens"])
    memory = dev(c["memory"]).requires_grad_()
    dec.init_state(src, memory, None)
    outs, attns = dec(dev(c["tgt"]), memory, memory_lengths=lens)
    assert_close(outs, c["dec_outs"], TOL32)
    assert_close(attns["std"], c["attn_std"], TOL32)
    ((outs * dev(c["w"])).sum() + attns["std"].sum() * 0.1).backward()
    assert_grads({k: p.grad for k, p in dec.named_parameters()}, c["g"], TOL32, "dec")
    assert_close(memory.grad, c["g_memory"], TOL32)
    # step-wise with soft inputs, gradient through the growing cache
    dec.zero_grad()
    memory.grad = None
    dec.init_state(src, memory, None)
    so, sa = [], []
    soft_in = dev(c["soft_in"])
    for s in range(soft_in.size(0)):
        o, a = dec(soft_in[s:s + 1], memory, memory_lengths=lens, step=s)
        so.append(o)
        sa.append(a["std"])
    so = torch.cat(so, 0)
    assert_close(so, c["step_outs"], TOL32)
    assert_close(torch.cat(sa, 0), c["step_attn"], TOL32)
    (so * dev(c["ws"])).sum().backward()
    assert_grads({k: p.grad for k, p in dec.named_parameters()}, c["g_step"], TOL32, "dec_step")
    assert_close(memory.grad, c["g_step_memory"], TOL32)
    for k, v in c["cache0"].items():
        assert_close(dec.state["cache"]["layer_0"][k], v, TOL32, k)
    dec.init_state(src, memory, None)
    with torch.no_grad():
        ids = torch.cat([dec(dev(c["tgt"])[s:s + 1], memory, memory_lengths=lens, step=s)[0] for s in range(4)], 0)
    assert_close(ids, c["id_step_outs"], TOL32)


@pytest.mark.parametrize("M,N,K", [(80, 512, 512), (1, 512, 64), (16, 1536, 512), (128, 9606, 512), (80, 512, 9632),
                                   (77, 130, 1024), (80, 1024, 2048), (1280, 512, 512), (1360, 1024, 256),
                                   (129, 48, 1024), (300, 100, 4096)])
def test_gemm_skinny_path(ops, M, N, K):
    """The M <= 128 weight-streaming kernel (gemm_skinny.hip) against fp64 and against the general
    tiled kernel, including split-K (K > 1024) and every epilogue feature."""
    from mssrnet_style_transfer_amd import _lib
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K).bfloat16()
    w = torch.randn(N, K).bfloat16()
    ref = a.double() @ w.double().t()
    ac, wc = a.cuda(), w.cuda()
    out = ops.gemm(ac, wc, True, torch.bfloat16)
    assert_close(out.float(), ref, 1e-2, "skinny")
    out32 = ops.gemm(ac, wc, True, torch.float32)
    assert_close(out32, ref, 2e-5 * math.sqrt(K), "skinny f32 out")
    bias = torch.randn(N).cuda()
    res = torch.randn(M, N).bfloat16().cuda()
    gate = torch.randn(M, N).bfloat16().cuda()
    kw = dict(bias=bias, relu=True, residual=res, gate=gate, alpha=0.5, drop_p=0.25, drop_seed=99)
    o1 = ops.gemm(ac, wc, True, torch.float32, **kw)
    o2 = ops.gemm(ac, wc, True, torch.float32, no_skinny=True, **kw)
    assert torch.equal(o1 == 0, o2 == 0)                    # same keep mask / gate pattern
    assert_close(o1, o2, 1e-4, "skinny vs tiled epilogue")
    kept = ops.gemm(ac, wc, True, torch.float32, drop_p=0.25, drop_seed=5)
    frac = (kept != 0).float().mean().item()
    assert abs(frac - 0.75) < 0.05 or M * N < 2000
    sel = kept != 0
    assert_close(kept[sel], (out32 / 0.75)[sel], 1e-5, "dropout scaling")


def test_gemm_skinny_operand_beyond_4gb(ops):
    """The LDS-DMA form of the K-split kernel addresses its operands with 32-bit byte offsets from the operand base: an
    operand whose rows span 4 GB or more (here 24 rows 2^27 elements apart in a 6.4 GB buffer) has to take the
    register-form kernel, with the same result as the compact copy of the same rows."""
    torch.manual_seed(3)
    M, N, K = 24, 512, 512
    stride = 1 << 27
    big = torch.empty((M - 1) * stride + K, dtype=torch.bfloat16, device="cuda")
    a = big.as_strided((M, K), (stride, 1))
    a.copy_(torch.randn(M, K).bfloat16())
    w = torch.randn(N, K).bfloat16().cuda()
    ops.start_profile()
    far = ops.gemm(a, w, True, torch.float32)
    near = ops.gemm(a.contiguous(), w, True, torch.float32)
    names = [r["kernel"] for r in ops.stop_profile()]
    assert names[0].startswith("gemm_skinny_kernel<") and names[1].startswith("gemm_skinny_dma_kernel<"), names
    assert_close(far, near, 1e-5, "rows 256 MB apart")
    assert_close(far, a.double() @ w.double().t(), 2e-5 * math.sqrt(K), "against fp64")
    del big


def test_gemm_skinny_beta_accumulates(ops):
    """beta * C_old in the skinny epilogue (the decode backward's dprobs feedback, fp32 output)."""
    torch.manual_seed(5)
    a = torch.randn(80, 512).bfloat16().cuda()
    w = torch.randn(1000, 512).bfloat16().cuda()
    c0 = torch.randn(80, 1000).cuda()
    out = c0.clone()
    ops.gemm(a, w, True, torch.float32, out=out, beta=1.0)
    assert_close(out, c0.double() + a.double() @ w.double().t(), 2e-5 * math.sqrt(512), "beta")


@pytest.mark.parametrize("M,N,K", [(80, 1536, 512), (20, 512, 512), (80, 1024, 512), (33, 48, 256), (1360, 768, 256),
                                   (1280, 1536, 512)])
def test_gemm_skinny_layernorm_prologue(ops, M, N, K):
    """LayerNorm folded into the product: raw rows x, gamma-scaled weights, column sums and a folded
    bias give LN(x) W^T + b and the row statistics, against fp64 LayerNorm -> Linear."""
    torch.manual_seed(M * N + K)
    x = (torch.randn(M, K) * 1.7 + 0.3).bfloat16()
    gamma, beta = torch.rand(K) + 0.5, torch.randn(K) * 0.2
    W, b = torch.randn(N, K) / math.sqrt(K), torch.randn(N)
    xd = x.double()
    mean = xd.mean(1, keepdim=True)
    var = xd.var(1, unbiased=False, keepdim=True)
    ref = ((xd - mean) / (var + 1e-6).sqrt() * gamma.double() + beta.double()) @ W.double().t() + b.double()
    wg = (W * gamma).bfloat16()
    colsum = wg.float().sum(1).cuda()
    bias = (W.double() @ beta.double() + b.double()).float().cuda()
    mo, ro = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    out = ops.gemm(x.cuda(), wg.cuda(), True, torch.bfloat16, bias=bias, ln=(colsum, mo, ro, 1e-6))
    assert_close(out.float(), ref, 2e-2, "ln-gemm")
    assert_close(mo, mean.squeeze(1), 1e-4, "mean", floor=1e-5)
    assert_close(ro, 1.0 / (var.squeeze(1) + 1e-6).sqrt(), 1e-4, "rstd")
    with pytest.raises(RuntimeError):
        ops.gemm(x.cuda(), wg.cuda(), True, torch.bfloat16, bias=bias, ln=(colsum, mo, ro, 1e-6), no_skinny=True)


@pytest.mark.parametrize("M,N,K,batch", [(512, 512, 1280, 1), (1536, 512, 2240, 3), (130, 70, 100, 1), (9606, 256, 512, 1),
                                         (64, 64, 64, 2), (48, 1024, 333, 1),
                                         # few output tiles, long reduction: K sliced over blockIdx.z, fp32 atomics
                                         (256, 256, 33280, 1), (130, 70, 9000, 1), (64, 192, 8192, 3)])
def test_gemm_tn_weight_gradient_kernel(ops, M, N, K, batch):
    """dW = dY^T X with both operands k-major (gemm_tn.hip: LDS tiles read back through
    ds_read_b64_tr_b16) against fp64, fp32 and bf16 outputs, beta accumulation, batched over layers,
    ragged edges and unaligned leading dimensions."""
    torch.manual_seed(M + N + K)
    dy = torch.randn(batch, K, M).bfloat16().cuda()
    x = torch.randn(batch, K, N).bfloat16().cuda()
    ref = dy.double().transpose(1, 2) @ x.double()
    from mssrnet_style_transfer_amd import _lib
    ops.start_profile()
    out = ops.gemm(dy.transpose(1, 2), x, False, torch.float32)
    rec = ops.stop_profile()
    # large aligned problems go to the LDS-DMA kernel (gemm_pipe.hip), the rest to the register-staged one (gemm_tn.hip)
    assert rec[0]["kernel"].startswith(("gemm_tn_kernel", "gemm_pipe_kernel")), rec[0]["kernel"]
    assert_close(out, ref, 2e-5 * math.sqrt(K), "tn f32")
    out16 = ops.gemm(dy.transpose(1, 2), x, False, torch.bfloat16)
    assert_close(out16.float(), ref, 1e-2, "tn bf16")
    acc = torch.randn(batch, M, N, device="cuda")
    want = acc.double() * 0.5 + 2.0 * ref
    ops.gemm(dy.transpose(1, 2), x, False, torch.float32, out=acc, alpha=2.0, beta=0.5)
    assert_close(acc, want, 3e-5 * math.sqrt(K), "tn alpha/beta")
    acc1 = torch.randn(batch, M, N, device="cuda")
    want1 = acc1.double() + ref
    ops.gemm(dy.transpose(1, 2), x, False, torch.float32, out=acc1, beta=1.0)
    assert_close(acc1, want1, 3e-5 * math.sqrt(K), "tn accumulate (beta = 1)")
    if batch == 1:          # column slices: leading dimensions that are not multiples of 8, unaligned bases
        dys, xs = dy[0][:, 3:M - 2], x[0][:, 1:N - 4]
        o = ops.gemm(dys.t(), xs, False, torch.float32)
        assert_close(o, ref[0][3:M - 2, 1:N - 4], 2e-5 * math.sqrt(K), "tn sliced")


PIPE_CASES = [
    # form, batch dims, M, N, K      (nt: b is [N,K]; nn: b is [K,N]; tn: a is a transposed view of [K,M], b is [K,N])
    ("nt", (), 8192, 512, 512), ("nt", (), 8200, 520, 544), ("nn", (), 8192, 512, 512), ("nn", (), 4100, 264, 96),
    ("nt", (3,), 2048, 384, 64), ("nn", (2, 2), 1024, 256, 128), ("tn", (), 512, 512, 2240), ("tn", (12,), 512, 256, 1440),
    ("tn", (), 1536, 512, 35840), ("tn", (), 9608, 512, 2240), ("nt", (), 32768, 512, 1024),
]


@pytest.mark.parametrize("form,bd,M,N,K", PIPE_CASES)
def test_gemm_pipe_kernel(ops, form, bd, M, N, K):
    """The LDS-DMA 128x128 kernel (gemm_pipe.hip) in its three operand forms against fp64: plain, fp32 output, ragged
    edges (M, N not multiples of 128, K % 64 == 32), batched, K sliced with atomics; and its vectorised epilogue
    (bias, ReLU, dropout, residual, gate, alpha / beta) against the register-staged kernel of gemm.hip run on the same
    operands and the same dropout seed (no_skinny=True forces that kernel): same masks, same values up to bf16 rounding."""
    torch.manual_seed(M + 3 * N + K)
    if form == "tn":
        at = torch.randn(*bd, K, M).bfloat16().cuda()
        a = at.transpose(-1, -2)
    else:
        a = torch.randn(*bd, M, K).bfloat16().cuda()
    tb = form == "nt"
    b = (torch.randn(*bd, N, K) if tb else torch.randn(*bd, K, N)).bfloat16().cuda()
    ref = a.double() @ (b.double().transpose(-1, -2) if tb else b.double())
    ops.start_profile()
    out = ops.gemm(a, b, tb, torch.float32)
    rec = ops.stop_profile()
    assert rec[0]["kernel"].startswith("gemm_pipe_kernel"), rec[0]["kernel"]
    assert_close(out, ref, 2e-5 * math.sqrt(K), "pipe %s f32" % form)
    out16 = ops.gemm(a, b, tb, torch.bfloat16)
    assert_close(out16.float(), ref, 1e-2, "pipe %s bf16" % form)
    acc = torch.randn(*bd, M, N, device="cuda")
    want = acc.double() + 2.0 * ref
    ops.gemm(a, b, tb, torch.float32, out=acc, alpha=2.0, beta=1.0)
    assert_close(acc, want, 3e-5 * math.sqrt(K), "pipe %s accumulate" % form)
    if form == "tn":
        return
    bias = torch.randn(N, device="cuda")
    res = torch.randn(*bd, M, N, device="cuda").bfloat16()
    gate = torch.randn(*bd, M, N, device="cuda").bfloat16()
    for od in (torch.bfloat16, torch.float32):
        kw = dict(bias=bias, relu=True, residual=res, gate=gate, alpha=0.5, drop_p=0.25, drop_seed=1234)
        got = ops.gemm(a, b, tb, od, **kw)
        old = ops.gemm(a, b, tb, od, no_skinny=True, **kw)
        assert_close(got.float(), old.float(), 1e-2 if od == torch.bfloat16 else 1e-5, "pipe %s epilogue %s" % (form, od))
        assert bool(((got == 0) == (old == 0)).all()), "dropout / gate masks differ"
        c0 = torch.randn(*bd, M, N, device="cuda").to(od)
        got = ops.gemm(a, b, tb, od, bias=bias, out=c0.clone(), beta=0.5)
        assert_close(got.float(), ref + bias.double() + 0.5 * c0.double(), 1e-2 if od == torch.bfloat16 else 3e-5 * math.sqrt(K),
                     "pipe %s beta %s" % (form, od))


@pytest.mark.parametrize("M,N,K,dtype", [(1, 512, 32768, torch.float32), (1, 129, 33280, torch.bfloat16),
                                         (512, 256, 8192, torch.float32), (4, 256, 4100, torch.bfloat16)])
def test_gemm_general_kernel_k_slices(ops, M, N, K, dtype):
    """Few output tiles, long reduction, fp32 output (weight gradients of the critics' score layers): the general
    kernel slices K over blockIdx.z and adds the slices with atomics; beta = 0 and beta = 1."""
    torch.manual_seed(K + N)
    at = torch.randn(K, M).to(dtype).cuda()
    b = torch.randn(K, N).to(dtype).cuda()
    ref = at.double().t() @ b.double()
    out = ops.gemm(at.t(), b, False, torch.float32, no_skinny=True)
    tol = 2e-5 * math.sqrt(K)
    assert_close(out, ref, tol, "k-sliced general kernel")
    acc = torch.randn(M, N, device="cuda")
    want = acc.double() + ref
    ops.gemm(at.t(), b, False, torch.float32, out=acc, beta=1.0, no_skinny=True)
    assert_close(acc, want, tol, "k-sliced general kernel, beta = 1")


def test_gemm_pipe_ragged_width_of_a_padded_operand(ops):
    """dW = dlogits^T x with V = 9606 columns stored in rows padded to 9632 (the vocabulary slabs of the decode engine):
    the 8-column chunks of the last tile read on into the padding, whose garbage only reaches rows that are not stored."""
    torch.manual_seed(1)
    K, V, Vp, N = 2240, 9606, 9632, 512
    buf = torch.full((K, Vp), float("nan")).bfloat16().cuda()
    buf[:, :V] = torch.randn(K, V).bfloat16().cuda()
    x = torch.randn(K, N).bfloat16().cuda()
    dl = buf[:, :V]
    ops.start_profile()
    out = ops.gemm(dl.t(), x, False, torch.float32)
    rec = ops.stop_profile()
    assert rec[0]["kernel"].startswith("gemm_pipe_kernel"), rec[0]["kernel"]
    assert_close(out, dl.double().t() @ x.double(), 2e-5 * math.sqrt(K), "ragged RM operand")


def test_adam_and_global_norm_clip_match_torch():
    """csrc/optim.hip against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (what onmt/trainer_wgan_wh.py:192-211,
    623-628 runs): odd sizes, a parameter without gradient, clipping active and inactive, five steps."""
    from mssrnet_style_transfer_amd.optim import Adam
    torch.manual_seed(3)
    shapes = [(129,), (1, 129), (512, 512), (3,), (9606, 64), (7, 5), (1030,)]
    mine = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt = Adam(mine, lr=4e-3, betas=(0.5, 0.9))
    topt = torch.optim.Adam(ref, lr=4e-3, betas=(0.5, 0.9))
    for step in range(5):
        max_norm = (0.0, 1e9, 5.0, 0.5, 100.0)[step]
        scale = (1.0, 1.0, 3.0, 0.01, 1.0)[step]
        for i, (a, b) in enumerate(zip(mine, ref)):
            if i == 3 and step % 2 == 0:
                a.grad = b.grad = None              # a parameter that received no gradient this step
                continue
            g = torch.randn_like(a) * scale
            a.grad, b.grad = g.clone(), g.clone()
        if max_norm > 0:
            want_norm = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        opt.step(max_norm)
        topt.step()
        if max_norm > 0:
            assert abs(float(opt.last_norm[0]) - float(want_norm)) <= 1e-5 * float(want_norm)
        for i, (a, b) in enumerate(zip(mine, ref)):
            assert_close(a, b, 1e-6, "step %d param %d" % (step, i))
            if a.grad is not None and max_norm > 0:
                assert_close(a.grad, b.grad, 1e-6, "step %d clipped grad %d" % (step, i))
    # determinism of the norm: same gradients, same bits
    n1 = opt.grad_norm(1.0).clone()
    n2 = opt.grad_norm(1.0).clone()
    assert torch.equal(n1, n2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,n,ld", [(4480, 1536, 1536), (1, 512, 512), (37, 520, 528), (1000, 9606, 9606), (333, 50, 50),
                                       (70000, 512, 1536)])
def test_colsum_matches_torch(ops, dtype, rows, n, ld):
    """Bias gradients = column sums over all (step, sentence) rows: the 16-byte wide kernel (aligned shapes) and the
    scalar kernel (ragged widths such as V = 9606), plain and batched over layers, against float64."""
    from mssrnet_style_transfer_amd import _lib
    from mssrnet_style_transfer_amd.decode_engine import _bsum, _colsum
    torch.manual_seed(rows + n)
    x = torch.randn(rows, ld, device="cuda").to(dtype)
    view = x[:, :n]
    want = view.double().sum(0).float()
    tol = 1e-5 if dtype == torch.float32 else 1e-5            # fp32 accumulation of exactly representable inputs
    got = _colsum(view)
    assert_close(got, want, tol, "colsum", floor=1e-3 * rows ** 0.5 * 1e-2)
    if rows <= 5000:
        xb = torch.randn(3, rows, ld, device="cuda").to(dtype)
        gb = _bsum(xb[:, :, :n])
        assert_close(gb, xb[:, :, :n].double().sum(1).float(), tol, "colsum batched", floor=1e-3 * rows ** 0.5 * 1e-2)
    assert _lib.load() is not None


@pytest.mark.parametrize("M,N,K", [(240, 9606, 512), (160, 9606, 512), (80, 2050, 256), (600, 4100, 512), (65, 9606, 256)])
@pytest.mark.parametrize("ln", [False, True])
def test_gemm_wide_vocabulary_kernel(ops, M, N, K, ln):
    """The all-rows x 48-columns kernel (csrc/gemm_wide.hip) for vocabulary-wide products with few rows: generator
    logits with the final LayerNorm folded in (bf16 out, row statistics written), and the soft embedding's input
    gradient accumulated into an fp32 slab (beta = 1, ragged leading dimension V = 9606); against fp64."""
    torch.manual_seed(M + N + K)
    x = (torch.randn(M, K) * 1.3 + 0.2).bfloat16()
    W, b = torch.randn(N, K) / math.sqrt(K), torch.randn(N)
    ops.start_profile()
    if ln:
        gamma, beta = torch.rand(K) + 0.5, torch.randn(K) * 0.2
        xd = x.double()
        mean = xd.mean(1, keepdim=True)
        var = xd.var(1, unbiased=False, keepdim=True)
        ref = ((xd - mean) / (var + 1e-6).sqrt() * gamma.double() + beta.double()) @ W.double().t() + b.double()
        wg = (W * gamma).bfloat16()
        colsum = wg.float().sum(1).cuda()
        bias = (W.double() @ beta.double() + b.double()).float().cuda()
        mo, ro = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
        Np = (N + 31) // 32 * 32
        out = torch.zeros(M, Np, device="cuda", dtype=torch.bfloat16)[:, :N]           # padded rows, as the logits slab
        ops.gemm(x.cuda(), wg.cuda(), True, torch.bfloat16, bias=bias, ln=(colsum, mo, ro, 1e-6), out=out)
        rec = ops.stop_profile()
        assert_close(out.float(), ref, 2e-2, "ln-gemm wide")
        assert_close(mo, mean.squeeze(1), 1e-4, "mean", floor=1e-5)
        assert_close(ro, 1.0 / (var.squeeze(1) + 1e-6).sqrt(), 1e-4, "rstd")
    else:
        wb = W.bfloat16()
        old = torch.randn(M, N, device="cuda")
        out = old.clone()
        ops.gemm(x.cuda(), wb.cuda(), True, torch.float32, out=out, beta=1.0, alpha=0.5)
        rec = ops.stop_profile()
        ref = 0.5 * (x.double() @ wb.double().t()) + old.double().cpu()
        assert_close(out, ref.float(), 1e-4, "wide, fp32 accumulate", floor=1e-4)
        ops.start_profile()
        out2 = ops.gemm(x.cuda(), wb.cuda(), True, torch.bfloat16, bias=b.cuda())      # memory keys|values of all layers
        rec += ops.stop_profile()
        assert_close(out2.float(), (x.double() @ wb.double().t() + b.double()).float(), 1e-2, "wide, bias, bf16 out")
    assert any(r["kernel"].startswith("gemm_wide_kernel") for r in rec), [r["kernel"] for r in rec]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("NL,T,B,b0,H,dh,Ls", [(3, 7, 5, 0, 8, 64, 18), (2, 28, 6, 2, 4, 32, 40), (1, 3, 9, 8, 2, 16, 1)])
def test_ctx_kv_grad_kernel(ops, dtype, NL, T, B, b0, H, dh, Ls):
    """Memory key / value gradients of a decode sweep (mssr_ctx_kv_grad): the per-step score-gradient slabs contracted
    with the saved queries / context gradients over the steps, all layers in one launch, against a float64 einsum;
    sentences below b0 (the eval rows of a mixed sweep) are left untouched."""
    from mssrnet_style_transfer_amd import _lib as L
    torch.manual_seed(NL * 100 + T + B)
    d = H * dh
    ds = torch.randn(NL, 2, T, B * H, Ls, device="cuda")
    act = torch.randn(NL, 8, T * B, d, device="cuda").to(dtype)          # queries live in a strided slab, kind 4
    dctx = torch.randn(NL, 2, T * B, d, device="cuda").to(dtype)
    qc, dc = act[:, 4], dctx[:, 1]
    out = torch.full((B, Ls, NL, 2, H, dh), 7.0, device="cuda", dtype=dtype)
    L.check(L.load().mssr_ctx_kv_grad(ds.data_ptr(), qc.data_ptr(), qc.stride(0), dc.data_ptr(), dc.stride(0), out.data_ptr(),
                                      NL, T, B, b0, H, dh, Ls, L.dt(out), L.stream()), "ctx_kv_grad")
    dsv = ds.view(NL, 2, T, B, H, Ls).double()
    want = torch.stack([torch.einsum("ltbhj,ltbhe->bjlhe", dsv[:, 0], qc.reshape(NL, T, B, H, dh).double()),
                        torch.einsum("ltbhj,ltbhe->bjlhe", dsv[:, 1], dc.reshape(NL, T, B, H, dh).double())], 3)
    assert_close(out[b0:], want[b0:].float(), 1e-5 if dtype == torch.float32 else 8e-3, "ctx_kv_grad")
    assert bool((out[:b0] == 7.0).all())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n0,n1,d,strided", [(18, 80, 9606, True), (80, 16, 256, False), (3, 2, 7, True)])
def test_gp_penalty_kernel(ops, dtype, n0, n1, d, strided):
    """WGAN-GP penalty of per-token gradients in one launch (ops.gp_penalty) against the reference's formula
    ((g[mask].norm(2, dim=1) - k) ** 2).mean() * lamb (trainer_wgan_wh.py:1253-1261) in float64, value and gradient,
    on a strided slice of a larger gradient (what torch.cat's backward hands over) and a transposed mask; a selected
    row whose gradient is exactly zero contributes (0 - k)^2 with a zero subgradient, not inf * 0 = NaN."""
    torch.manual_seed(n0 * n1 + d)
    full = torch.randn(n0, 2 * n1 if strided else n1, d, device="cuda").to(dtype)
    full[1, -1] = 0                                            # a zero gradient row inside the mask
    full.requires_grad_(True)
    g = full[:, n1:] if strided else full
    mask = (torch.rand(n1, n0, device="cuda") < 0.7).t()       # [n0, n1], transposed strides
    mask[1, -1] = True
    mask[0, 0] = False
    k, lamb = 1.0, 10.0
    pen = ops.gp_penalty(g, mask, k, lamb)
    pen.backward(torch.tensor(0.5, device="cuda"))
    ref_in = full.detach().double().requires_grad_(True)
    gr = ref_in[:, n1:] if strided else ref_in
    want = ((gr[mask].norm(2, dim=1) - k) ** 2).mean() * lamb
    want.backward(torch.tensor(0.5, device="cuda", dtype=torch.float64))
    assert abs(float(pen) - float(want)) <= 1e-4 * abs(float(want))
    assert pen.dtype == torch.float32
    assert_close(full.grad.float(), ref_in.grad.float(), 1e-5 if dtype == torch.float32 else 8e-3, "gp_penalty gradient")
    assert bool(torch.isfinite(full.grad).all())
    assert float(full.grad[1, -1].abs().max()) == 0.0 and float(full.grad[:, n1:][0, 0].abs().max() if strided else full.grad[0, 0].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops.gp_penalty(g.detach().cpu(), mask.cpu(), k, lamb)   # no CPU path


@pytest.mark.parametrize("sigmoid", [True, False])
def test_gain_pick_twice_differentiable(ops, sigmoid):
    """ops.gain_pick = the critics' sigmoid gain followed by cal_disc_loss's per-row pick
    (trainer_wgan_wh.py:402-416, :1192-1200) in one launch, against torch in float64: value, gradient, and the
    gradient of a function of the gradient (what the WGAN-GP penalty takes)."""
    torch.manual_seed(7)
    n, C, ratio = 240, 4, 1.5
    base = torch.randn(n, C + 2, device="cuda")
    labels = torch.randint(0, C, (n,), device="cuda")
    w = torch.randn(n, device="cuda")

    def run(x, pick):
        z = (x * x).cumsum(1) * 0.3 + x                          # something upstream with curvature
        picked = pick(z)
        g, = torch.autograd.grad(picked, x, w.to(picked.dtype), create_graph=True)
        loss = (g ** 2).sum() + picked.sum()
        loss.backward()
        return picked.detach(), g.detach(), x.grad

    x32 = base[:, :C].clone().requires_grad_(True)               # contiguous copy; strided logits are covered below
    got = run(x32, lambda z: ops.gain_pick(z, labels, ratio, sigmoid))
    x64 = base[:, :C].double().requires_grad_(True)
    ref = lambda z: ((torch.sigmoid(z) * ratio) if sigmoid else z).gather(1, labels.view(-1, 1)).squeeze(1)  # noqa: E731
    want = run(x64, ref)
    for a, b, name in zip(got, want, ("picked", "first derivative", "second-order gradient")):
        assert_close(a, b.float(), 2e-5, "gain_pick " + name)
    strided = ops.gain_pick(base[:, :C], labels, ratio, sigmoid)  # row stride C + 2
    assert_close(strided, ref(base[:, :C].double()).float(), 2e-5, "gain_pick strided")
    bad = labels.clone()
    bad[3] = C
    assert bool(torch.isnan(ops.gain_pick(base[:, :C], bad, ratio, sigmoid)[3]))


@pytest.mark.parametrize("dtype,time_major", [(torch.float32, False), (torch.float32, True), (torch.bfloat16, True)])
def test_ctx_pool_twice_differentiable(ops, dtype, time_major):
    """ContextReps2 (onmt/style_generator_builder.py:243-256: Linear-ReLU-Linear scores, -inf mask past the length,
    softmax over time, weighted sum) through ops.linear + ops.ctx_pool against the same composition in torch float64:
    the pooled states, every first-order gradient, and the gradients of a function of the input gradient (the shape
    of the WGAN-GP penalty) -- in fp32 at 2e-5, in bf16 within bf16's rounding of the operands."""
    torch.manual_seed(11)
    B, Lx, d, m = 9, 13, 64, 33                                 # m odd, as the reference's d / 2 + 1
    lengths = torch.tensor([13, 1, 7, 12, 13, 2, 5, 9, 10], device="cuda")
    base = torch.randn(Lx, B, d, device="cuda") if time_major else torch.randn(B, Lx, d, device="cuda")
    P = [torch.randn(m, d, device="cuda") * 0.2, torch.randn(m, device="cuda") * 0.1, torch.randn(1, m, device="cuda"),
         torch.randn(1, device="cuda")]
    wv = torch.randn(B, d, device="cuda")
    ops.set_compute_dtype(dtype)

    def run(x0, params, pool):
        x = x0.transpose(0, 1) if time_major else x0
        z = x + 0.1 * x * x                                    # something upstream with curvature
        pooled = pool(z, *params)
        g, = torch.autograd.grad(pooled, x0, wv.to(pooled.dtype), create_graph=True)
        loss = (g.double() ** 2).sum() * 3.0 + (pooled.double() * pooled.double()).sum()
        loss.backward()
        return [pooled.detach(), g.detach(), x0.grad] + [q.grad for q in params]

    def ref_pool(z, w1, b1, w2, b2):
        h = torch.relu(z @ w1.t() + b1)
        sc = (h @ w2.t() + b2).squeeze(2)
        t = torch.arange(Lx, device="cuda").unsqueeze(0)
        sc = sc.masked_fill(t >= lengths.unsqueeze(1), float("-inf"))
        return (torch.softmax(sc, 1).unsqueeze(2) * z).sum(1)

    def our_pool(z, w1, b1, w2, b2):
        return ops.ctx_pool(z, ops.linear(z, w1, b1, relu=True), w2, b2, lengths)

    got = run(base.clone().requires_grad_(True), [q.clone().requires_grad_(True) for q in P], our_pool)
    want = run(base.double().requires_grad_(True), [q.double().requires_grad_(True) for q in P], ref_pool)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    names = ("pooled", "input gradient", "x.grad (second order)", "w1.grad", "b1.grad", "w2.grad", "b2.grad")
    for a, b, name in zip(got, want, names):
        assert a is not None, name
        # b2 shifts every score of a sentence alike and the softmax ignores it: its exact gradient is 0
        assert_close(a.float(), b.float(), tol, "ctx_pool " + name, floor=1e-3 if name == "b2.grad" else 0.0)
    # rows past a sentence's length receive no gradient at all
    xg = got[2].transpose(0, 1) if time_major else got[2]
    assert float(xg[1, 1:].abs().max()) == 0.0 and float(xg[5, 2:].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops.ctx_pool(base.cpu(), base.cpu(), P[2].cpu(), P[3].cpu(), lengths.cpu())   # no CPU path


@pytest.mark.parametrize("M,N,K", [(240, 512, 512), (80, 512, 256), (37, 130, 48), (1, 4, 16), (1000, 64, 1024)])
def test_gemm_skinny_f32_kernel(ops, M, N, K):
    """fp32 K-split kernel (exact f32 MFMA; the style critic's fp32 island and the fp32 parity runs): plain product,
    bias + ReLU, residual + gate + beta accumulation, against float64; and the profile must name the K-split kernel."""
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / math.sqrt(K)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    want = a.double() @ w.double().t()
    ops.start_profile()
    got = ops.gemm(a, w, True, torch.float32)
    rec = ops.stop_profile()
    assert rec and rec[0]["kernel"] == "gemm_skinny_f32_kernel", rec
    assert_close(got, want.float(), 2e-6, "f32 K-split plain", floor=1e-6)
    got = ops.gemm(a, w, True, torch.float32, bias=bias, relu=True)
    assert_close(got, torch.relu(want + bias.double()).float(), 2e-6, "f32 K-split bias relu", floor=1e-6)
    out = torch.randn(M, N, device="cuda")
    ref = torch.where(res.double() > 0, 0.5 * want + res.double(), torch.zeros((), device="cuda", dtype=torch.float64)) \
        + out.double()
    got = ops.gemm(a, w, True, torch.float32, residual=res, gate=res, alpha=0.5, out=out, beta=1.0)
    assert_close(got, ref.float(), 2e-6, "f32 K-split residual gate beta", floor=1e-6)
    # strided operands (row stride > K) keep the kernel; unaligned ones fall back to the general kernel
    big = torch.randn(M, K + 8, device="cuda")
    assert_close(ops.gemm(big[:, :K], w, True, torch.float32), (big[:, :K].double() @ w.double().t()).float(), 2e-6,
                 "strided A", floor=1e-6)
