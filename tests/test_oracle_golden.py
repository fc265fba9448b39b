"""The CPU oracle against the golden vectors the REFERENCE produced (tests/golden/*.pt).

This is the pin of SURVEY.md section 8c: every function in oracle/ is checked against
what the reference's own modules computed on the same inputs (forward, first-order
gradients, WGAN-GP double-backward gradients, per-phase scalars and gradients, greedy
token ids).  Tolerance: 2e-5 relative (both sides are fp32 CPU; only summation order
differs).  Runs without a GPU.
"""
import random
from types import SimpleNamespace

import torch

from helpers import assert_close, assert_grads, assert_post_step
from _seeded import seeded_state

from mssrnet_style_transfer_amd.config import MSSRConfig
from oracle import functional as F_
from oracle import phases as PH

TOL = 2e-5


def leafs(sd):
    out = {}
    for k, v in sd.items():
        t = v.clone()
        if t.is_floating_point() and not k.endswith("pe.pe"):
            t.requires_grad_(True)
        out[k] = t
    return out


def gp_probe(g):
    return ((g.reshape(-1, g.size(-1)).norm(2, dim=1) - 1.0) ** 2).mean()


def mha_template(d):
    t = {}
    for n in ("linear_keys", "linear_values", "linear_query", "final_linear"):
        t[n + ".weight"] = torch.empty(d, d)
        t[n + ".bias"] = torch.empty(d)
    return t


def test_layer_norm(golden_ops):
    for d in (64, 256, 512):
        c = golden_ops["ln_%d" % d]
        x = c["x"].clone().requires_grad_()
        g = c["gamma"].clone().requires_grad_()
        b = c["beta"].clone().requires_grad_()
        y = F_.layer_norm(x, g, b)
        assert_close(y, c["y"], TOL, "y")
        gx, gg, gb = torch.autograd.grad((y * c["w"]).sum(), (x, g, b), create_graph=True)
        assert_close(gx, c["gx"], TOL, "gx")
        assert_close(gg, c["ggamma"], TOL, "ggamma")
        assert_close(gb, c["gbeta"], TOL, "gbeta")
        probe = gp_probe(gx)
        assert_close(probe, c["probe"], TOL, "probe")
        g2x, g2g = torch.autograd.grad(probe, (x, g))
        assert_close(g2x, c["g2x"], 5e-5, "g2x")
        assert_close(g2g, c["g2gamma"], 5e-5, "g2gamma")


def run_mha_case(c, fn):
    d = c["d"]
    sd = c["sd"] if c["sd"] is not None else seeded_state(mha_template(d), c["seed"])
    P = leafs(sd)
    ins = {k: v.clone().requires_grad_() for k, v in c["inputs"].items()}
    if "x" in ins:
        key = value = query = ins["x"]
        at = "self"
    else:
        key = value = ins["mem"]
        query = ins["q"]
        at = "context"
    o, attn = fn(P, c["H"], key, value, query, c["mask"], at)
    return P, ins, o, attn


def check_mha_case(c, fn, tol, tol2):
    P, ins, o, attn = run_mha_case(c, fn)
    assert_close(o, c["out"], tol, "out")
    assert_close(attn, c["attn"], tol, "attn")
    names = list(ins.keys()) + list(P.keys())
    tens = list(ins.values()) + list(P.values())
    scalar = (o * c["wo"]).sum() + (attn * c["wa"]).sum()
    g = torch.autograd.grad(scalar, tens, create_graph=True)
    assert_grads(dict(zip(names, g)), c["g"], tol, "g", floor=2e-6)
    probe = sum(gp_probe(gi) for gi in g[:len(ins)])
    assert_close(probe, c["probe"], tol2, "probe")
    g2 = torch.autograd.grad(probe, tens, allow_unused=True)
    assert_grads(dict(zip(names, g2)), c["g2"], tol2, "g2", floor=1e-9)


def test_mha_nocache(golden_ops):
    fn = lambda P, H, k, v, q, m, at: F_.multi_head_attention(P, H, k, v, q, mask=m, attn_type=at)  # noqa
    for name in ("self_pad", "self_causal", "cross", "nomask", "self_pad_d256", "cross_d512"):
        check_mha_case(golden_ops["mha_" + name], fn, TOL, 1e-4)


def test_mha_cached(golden_ops):
    c = golden_ops["mha_cached"]
    Ps, Pc = c["sd_self"], c["sd_ctx"]
    mask = ~F_.sequence_mask(c["lens"], c["mem"].size(1)).unsqueeze(1)
    cache = {"self_keys": None, "self_values": None, "memory_keys": None, "memory_values": None}
    for st in c["steps"]:
        q = st["q"]
        o1, a1 = F_.multi_head_attention(Ps, c["H"], q, q, q, layer_cache=cache, attn_type="self")
        o2, a2 = F_.multi_head_attention(Pc, c["H"], c["mem"], c["mem"], q, mask=mask, layer_cache=cache,
                                         attn_type="context")
        assert_close(o1, st["self_out"], TOL)
        assert_close(a1, st["self_attn"], TOL)
        assert_close(o2, st["ctx_out"], TOL)
        assert_close(a2, st["ctx_attn"], TOL)
    for k, v in c["cache"].items():
        assert_close(cache[k], v, TOL, k)


def ffn_template(d, ff):
    return {"w_1.weight": torch.empty(ff, d), "w_1.bias": torch.empty(ff), "w_2.weight": torch.empty(d, ff),
            "w_2.bias": torch.empty(d), "layer_norm.weight": torch.empty(d), "layer_norm.bias": torch.empty(d)}


def test_ffn(golden_ops):
    for d in (64, 256):
        c = golden_ops["ffn_%d" % d]
        sd = c["sd"] if c["sd"] is not None else seeded_state(ffn_template(d, c["ff"]), c["seed"])
        P = leafs(sd)
        x = c["x"].clone().requires_grad_()
        y = F_.feed_forward(P, x)
        assert_close(y, c["y"], TOL, "y")
        names = ["x"] + list(P.keys())
        tens = [x] + list(P.values())
        g = torch.autograd.grad((y * c["w"]).sum(), tens, create_graph=True)
        assert_grads(dict(zip(names, g)), c["g"], TOL, "g")
        g2 = torch.autograd.grad(gp_probe(g[0]), tens, allow_unused=True)
        assert_grads(dict(zip(names, g2)), c["g2"], 1e-4, "g2", floor=1e-9)


def test_embeddings_and_pe(golden_ops):
    c = golden_ops["pe"]
    table = F_.sinusoid_table(16, 64)
    assert_close(table, c["table"].squeeze(1), 1e-6, "table")
    e = golden_ops["emb"]
    P = leafs(e["sd"])
    W = P["make_embedding.emb_luts.0.weight"]
    y = F_.embeddings(P, e["ids"], 1)
    assert_close(y, e["y_ids"], TOL)
    g, = torch.autograd.grad((y * e["w"]).sum(), W)
    assert_close(g, e["g_ids_w"], TOL)
    soft = e["soft"].clone().requires_grad_()
    y = F_.embeddings(P, soft, 1)
    assert_close(y, e["y_soft"], TOL)
    gi, gw = torch.autograd.grad((y * e["w"]).sum(), (soft, W))
    assert_close(gi, e["g_soft_in"], TOL)
    assert_close(gw, e["g_soft_w"], TOL)
    assert_close(F_.embeddings(P, soft[2:3], 1, step=2), e["y_soft_step2"], TOL)


def test_encoder(golden_modules):
    c = golden_modules["encoder"]
    P = leafs(c["sd"])
    emb, mem, _ = F_.encoder(P, 4, c["src"], c["lens"], 1)
    assert_close(emb, c["emb"], TOL)
    assert_close(mem, c["mem"], TOL)
    names = [k for k, v in P.items() if v.requires_grad]
    g = torch.autograd.grad((mem * c["w"]).sum(), [P[k] for k in names], allow_unused=True)
    assert_grads(dict(zip(names, g)), c["g"], TOL, "enc")
    soft = c["soft"].clone().requires_grad_()
    _, mem2, _ = F_.encoder(P, 4, soft, c["lens"], 1)
    assert_close(mem2, c["mem_soft"], TOL)
    gs, = torch.autograd.grad((mem2 * c["w"]).sum(), soft)
    assert_close(gs, c["g_soft_in"], TOL)


def test_decoder(golden_modules):
    c = golden_modules["decoder"]
    P = leafs(c["sd"])
    names = [k for k, v in P.items() if v.requires_grad]
    params = [P[k] for k in names]
    memory = c["memory"].clone().requires_grad_()
    st = F_.DecoderState(c["src"])
    outs, attn = F_.decoder(P, 4, st, c["tgt"], memory, c["lens"], 1)
    assert_close(outs, c["dec_outs"], TOL)
    assert_close(attn, c["attn_std"], TOL)
    g = torch.autograd.grad((outs * c["w"]).sum() + attn.sum() * 0.1, params + [memory], allow_unused=True)
    assert_grads(dict(zip(names, g[:-1])), c["g"], TOL, "dec")
    assert_close(g[-1], c["g_memory"], TOL)
    # step-wise, soft inputs, gradient through the growing cache
    st = F_.DecoderState(c["src"])
    so, sa = [], []
    for s in range(c["soft_in"].size(0)):
        o, a = F_.decoder(P, 4, st, c["soft_in"][s:s + 1], memory, c["lens"], 1, step=s)
        so.append(o)
        sa.append(a)
    so = torch.cat(so, 0)
    assert_close(so, c["step_outs"], TOL)
    assert_close(torch.cat(sa, 0), c["step_attn"], TOL)
    g = torch.autograd.grad((so * c["ws"]).sum(), params + [memory], allow_unused=True)
    assert_grads(dict(zip(names, g[:-1])), c["g_step"], TOL, "dec_step")
    assert_close(g[-1], c["g_step_memory"], TOL)
    for k, v in c["cache0"].items():
        assert_close(st.cache[0][k], v, TOL, k)
    st = F_.DecoderState(c["src"])
    with torch.no_grad():
        ids = torch.cat([F_.decoder(P, 4, st, c["tgt"][s:s + 1], memory, c["lens"], 1, step=s)[0]
                         for s in range(4)], 0)
    assert_close(ids, c["id_step_outs"], TOL)


def test_style_modules(golden_modules):
    cfg = MSSRConfig.small()
    c = golden_modules["style_generator"]
    P = leafs(c["sd"])
    rep = F_.style_generator(P, cfg.sgen[1], c["desired"], c["src"], c["lens"], 1)
    assert_close(rep, c["rep"], TOL)
    names = [k for k, v in P.items() if v.requires_grad]
    g = torch.autograd.grad((rep * c["w"]).sum(), [P[k] for k in names], allow_unused=True)
    assert_grads(dict(zip(names, g)), c["g"], TOL, "sgen")

    c = golden_modules["text_disc"]
    P = leafs(c["sd"])
    names = [k for k, v in P.items() if v.requires_grad]
    params = [P[k] for k in names]
    H = cfg.tdisc[1]
    logits = F_.style_classifier(P, H, c["src"], c["lens"], 1)
    assert_close(logits, c["logits"], TOL)
    g = torch.autograd.grad(logits.sum(), params, allow_unused=True)
    assert_grads(dict(zip(names, g)), c["g"], TOL, "tdisc")
    assert_close(F_.style_classifier(P, H, c["fake"], c["tlen"], 1), c["soft_logits"], TOL)
    onehot = PH.one_hot(c["vocab_target"].transpose(0, 1).squeeze(2), cfg.vocab_size)
    torch.manual_seed(c["gp_seed"])
    gp = PH.penalize_text_grad(lambda x, l: torch.sigmoid(F_.style_classifier(P, H, x, l, 1)), onehot,
                               c["fake"].transpose(0, 1), c["tlen"], c["labels"].tolist(), 1.0, 10.0, True)
    assert_close(gp, c["gp"], 5e-5)
    g = torch.autograd.grad(gp, params, allow_unused=True)
    assert_grads(dict(zip(names, g)), c["g_gp"], 1e-4, "tdisc_gp", floor=1e-9)

    c = golden_modules["style_disc"]
    P = leafs(c["sd"])
    names = list(P.keys())
    params = list(P.values())
    lg = F_.style_discriminator(P, c["real"], c["lens"])
    assert_close(lg, c["logits"], TOL)
    g = torch.autograd.grad(lg.sum(), params, allow_unused=True)
    assert_grads(dict(zip(names, g)), c["g"], TOL, "sdisc")
    torch.manual_seed(c["gp_seed"])
    gp = PH.penalize_style_grad(lambda x, l: torch.sigmoid(F_.style_discriminator(P, x, l)), c["real"],
                                c["fake"], c["lens"], c["labels"].tolist(), 1.0, 10.0, True)
    assert_close(gp, c["gp"], 5e-5)
    g = torch.autograd.grad(gp, params, allow_unused=True)
    assert_grads(dict(zip(names, g)), c["g_gp"], 1e-4, "sdisc_gp", floor=1e-9)

    c = golden_modules["bond"]
    assert_close(F_.bond_layer(c["sd"], c["memory"], c["style"]), c["out"], TOL)


def _batch(golden_phases):
    b = golden_phases["batch"]
    return SimpleNamespace(src=b["src"], src_lengths=b["src_lengths"], tgt=b["tgt"], labels=b["labels"])


def _phase(golden_phases, name, fn, trained, tol=1e-4):
    cfg = MSSRConfig.small()
    nets = PH.make_nets(golden_phases["init"], cfg)
    c = golden_phases[name]
    torch.manual_seed(c["seed"])
    random.seed(c["seed"])
    log = fn(nets, _batch(golden_phases), cfg)
    for k, v in c["scalars"].items():
        assert abs(log[k] - v) <= tol * max(abs(v), 1e-3), (name, k, log[k], v)
    for n in trained:
        assert_grads(nets[n].grads(), c["grads"][n], tol, name + "." + n, floor=1e-9)
        assert_post_step(nets[n].named_params(), c["post"][n], cfg.learning_rate, name + ".post." + n)


def test_phase_discriminator(golden_phases):
    _phase(golden_phases, "discriminator", PH.train_discriminator, ("text_discriminator", "style_discriminator"))


def test_phase_generator(golden_phases):
    _phase(golden_phases, "generator", PH.train_generator, ("model", "style_generator"))


def test_phase_reconstruct(golden_phases):
    _phase(golden_phases, "reconstruct", PH.train_reconstruct, ("model", "style_generator"))


def test_phase_init_launch(golden_phases):
    _phase(golden_phases, "init_launch", PH.init_launch, ("model", "style_generator"))


def test_soft_decode_and_teacher_step(golden_phases):
    cfg = MSSRConfig.small()
    nets = PH.make_nets(golden_phases["init"], cfg)
    nets["model"].training = nets["style_generator"].training = False
    labels, _, real_src, real_len, _ = PH.unpack_batch(_batch(golden_phases), cfg)
    c = golden_phases["decode"]
    with torch.no_grad():
        rep = PH.gen_style(nets, 1 - labels, real_src, real_len)
        probs, lens, ids = PH.dynamic_decoding(nets, real_src, real_len, rep, 9)
    assert_close(rep, c["rep"], TOL)
    assert_close(probs, c["probs"], 1e-4)
    assert torch.equal(ids, c["ids"])
    assert torch.equal(lens, c["seq_lens"])
    t = golden_phases["teacher_step"]
    nets["teacher"].training = True          # dropout is 0 in the small config
    loss, lg = PH.teacher_step_loss(nets, _batch(golden_phases), cfg)
    assert_close(loss, t["loss"], TOL)
    assert_close(lg, t["logits"], TOL)
    loss.backward()
    assert_grads(nets["teacher"].grads(), t["g"], 1e-4, "teacher", floor=1e-9)


def test_greedy(golden_phases, golden_greedy):
    cfg = MSSRConfig.small()
    nets = PH.make_nets(golden_phases["init"], cfg)
    with torch.no_grad():
        nets["model"].P["generator.0.bias"][cfg.eos_idx] += golden_greedy["eos_bias"]
    preds = PH.greedy_decode(nets, golden_greedy["src"], golden_greedy["lens"], golden_greedy["desired"], cfg,
                             golden_greedy["max_length"])
    assert preds == golden_greedy["predictions"]


def test_phase_gemm_work_matches_the_reference_count(golden_phases):
    """SURVEY 8d's algorithmic work: GEMM FLOPs (2*M*N*K, torch.utils.flop_counter's formulas) of each phase.
    tests/golden/flops.json holds the count over the REFERENCE trainer; the oracle, counted live on the small config,
    must do exactly that work, and the recorded bs80 counts (configs[1], what bench.py's `algorithmic` field uses)
    must agree between reference and oracle."""
    import json
    import os
    from torch.utils._python_dispatch import TorchDispatchMode
    from torch.utils.flop_counter import flop_registry

    class GemmFlops(TorchDispatchMode):
        total = 0

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            out = func(*args, **kwargs)
            formula = flop_registry.get(func._overloadpacket)
            if formula is not None:
                self.total += int(formula(*args, **kwargs, out_val=out))
            return out

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flops.json")) as f:
        rec = json.load(f)
    cfg = MSSRConfig.small()
    small = rec["small_b6_w7"]
    for fn in ("train_discriminator", "train_generator", "train_reconstruct"):
        nets = PH.make_nets(golden_phases["init"], cfg)
        torch.manual_seed(7)
        random.seed(7)
        with GemmFlops() as fc:
            getattr(PH, fn)(nets, _batch(golden_phases), cfg)
        assert fc.total == small[fn]["reference"], (fn, fc.total, small[fn])
        full = rec["configs1_b80_w16"][fn]
        assert full["reference"] == full["oracle"] and full["reference"] > 1e11, (fn, full)
