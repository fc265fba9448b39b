"""Callers of the hot path on the GPU (critics, style generator, soft decode, the four
training phases, teacher step) against the golden vectors the reference produced.

fp32 compute, dropout 0, tolerance 1e-4 relative on losses and gradients (north_star); the
bf16 run of the phases is checked at 2e-2 on the logged losses.
"""
import random
from types import SimpleNamespace

import pytest
import torch

from helpers import assert_close, assert_grads, assert_post_step
from mssrnet_style_transfer_amd.config import MSSRConfig

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    from mssrnet_style_transfer_amd import ops as o
    o.set_compute_dtype(torch.float32)
    yield o
    o.set_compute_dtype(torch.float32)


def dev(x):
    return x.cuda() if torch.is_tensor(x) else x


def grads_of(m):
    return {k: p.grad for k, p in m.named_parameters()}


def test_style_modules(ops, golden_modules):
    from mssrnet_style_transfer_amd import style_generator_builder as B
    from mssrnet_style_transfer_amd import trainer_wgan_wh as T
    cfg = MSSRConfig.small()
    c = golden_modules["style_generator"]
    sg = B.build_style_generator(cfg)
    sg.load_state_dict(c["sd"])
    sg.cuda()
    rep = sg(dev(c["desired"]), dev(c["src"]), dev(c["lens"]))
    assert_close(rep, c["rep"], TOL)
    (rep * dev(c["w"])).sum().backward()
    assert_grads(grads_of(sg), c["g"], TOL, "sgen")

    c = golden_modules["text_disc"]
    td = B.build_text_discriminator(cfg)
    td.load_state_dict(c["sd"])
    td.cuda()
    logits = td(dev(c["src"]), dev(c["lens"]))
    assert_close(logits, c["logits"], TOL)
    logits.sum().backward()
    assert_grads(grads_of(td), c["g"], TOL, "tdisc")
    assert_close(td(dev(c["fake"]), dev(c["tlen"])), c["soft_logits"], TOL)
    td.zero_grad()
    onehot = T.convert_sent_ids_to_onehot(dev(c["vocab_target"]).transpose(0, 1).squeeze(2), cfg.vocab_size)
    torch.manual_seed(c["gp_seed"])
    gp = T.penalize_gen_text_grad(lambda x, l: torch.sigmoid(td(x, l)), onehot, dev(c["fake"]).transpose(0, 1),
                                  dev(c["tlen"]), c["labels"].tolist(), 1.0, 10.0, True)
    assert_close(gp, c["gp"], TOL)
    gp.backward()
    assert_grads(grads_of(td), c["g_gp"], 2e-4, "tdisc_gp")

    c = golden_modules["style_disc"]
    sd = B.build_style_discriminator(cfg)
    sd.load_state_dict(c["sd"])
    sd.cuda()
    lg = sd(dev(c["real"]), dev(c["lens"]))
    assert_close(lg, c["logits"], TOL)
    lg.sum().backward()
    assert_grads(grads_of(sd), c["g"], TOL, "sdisc")
    sd.zero_grad()
    torch.manual_seed(c["gp_seed"])
    gp = T.penalize_gen_style_grad(lambda x, l: torch.sigmoid(sd(x, l)), dev(c["real"]), dev(c["fake"]),
                                   dev(c["lens"]), c["labels"].tolist(), 1.0, 10.0, True)
    assert_close(gp, c["gp"], TOL)
    gp.backward()
    assert_grads(grads_of(sd), c["g_gp"], 2e-4, "sdisc_gp")

    c = golden_modules["bond"]
    bond = B.BondLayer(cfg.d_sgen, cfg.d_model)
    bond.load_state_dict(c["sd"])
    bond.cuda()
    assert_close(bond(dev(c["memory"]), dev(c["style"])), c["out"], TOL)


def make_nets(golden_phases, cfg):
    from mssrnet_style_transfer_amd import style_generator_builder as B
    nets = B.build_all(cfg, device="cpu")
    for k, m in nets.items():
        m.load_state_dict(golden_phases["init"][k])
        m.cuda()
    return nets


def gpu_batch(golden_phases):
    b = golden_phases["batch"]
    return SimpleNamespace(src=b["src"].cuda(), src_lengths=b["src_lengths"].cuda(), tgt=b["tgt"].cuda(),
                           labels=b["labels"].cuda())


PHASES = [("discriminator", "train_discriminator", ("text_discriminator", "style_discriminator")),
          ("generator", "train_generator", ("model", "style_generator")),
          ("reconstruct", "train_reconstruct", ("model", "style_generator")),
          ("init_launch", "init_launch", ("model", "style_generator"))]


def test_generator_phase_reference_order_fp32(ops, golden_phases):
    """train_generator in the reference's own call order (three style-generator calls, two decodes,
    three backward calls: batch_generator_phase = False) against the same golden vectors."""
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg = MSSRConfig.small(batch_generator_phase=False)
    nets = make_nets(golden_phases, cfg)
    tr = Trainer(nets, cfg)
    c = golden_phases["generator"]
    torch.manual_seed(c["seed"])
    random.seed(c["seed"])
    log = tr.train_generator(gpu_batch(golden_phases))
    for k, v in c["scalars"].items():
        assert abs(log[k] - v) <= TOL * max(abs(v), 1e-3), (k, log[k], v)
    for n in ("model", "style_generator"):
        assert_grads(grads_of(nets[n]), c["grads"][n], 2e-4, "generator(ref order)." + n)


def test_discriminator_phase_reference_order_fp32(ops, golden_phases):
    """train_discriminator in the reference's own call order (each critic three times: real, fake, interpolated, with
    a backward after each; batch_critic_passes = False) against the same golden vectors as the batched default."""
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg = MSSRConfig.small(batch_critic_passes=False)
    nets = make_nets(golden_phases, cfg)
    tr = Trainer(nets, cfg)
    c = golden_phases["discriminator"]
    torch.manual_seed(c["seed"])
    random.seed(c["seed"])
    log = tr.train_discriminator(gpu_batch(golden_phases))
    for k, v in c["scalars"].items():
        assert abs(log[k] - v) <= TOL * max(abs(v), 1e-3), (k, log[k], v)
    for n in ("text_discriminator", "style_discriminator"):
        assert_grads(grads_of(nets[n]), c["grads"][n], 2e-4, "discriminator(ref order)." + n)


def test_reconstruct_phase_reference_order_fp32(ops, golden_phases):
    """train_reconstruct in the reference's own call order (two style-generator calls, three encoder calls,
    two decodes: batch_generator_phase = False) against the same golden vectors."""
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg = MSSRConfig.small(batch_generator_phase=False)
    nets = make_nets(golden_phases, cfg)
    tr = Trainer(nets, cfg)
    c = golden_phases["reconstruct"]
    torch.manual_seed(c["seed"])
    random.seed(c["seed"])
    log = tr.train_reconstruct(gpu_batch(golden_phases))
    for k, v in c["scalars"].items():
        assert abs(log[k] - v) <= TOL * max(abs(v), 1e-3), (k, log[k], v)
    for n in ("model", "style_generator"):
        assert_grads(grads_of(nets[n]), c["grads"][n], 2e-4, "reconstruct(ref order)." + n)


@pytest.mark.parametrize("name,fn,trained", PHASES)
def test_phase_fp32(ops, golden_phases, name, fn, trained):
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg = MSSRConfig.small()
    nets = make_nets(golden_phases, cfg)
    tr = Trainer(nets, cfg)
    c = golden_phases[name]
    torch.manual_seed(c["seed"])
    random.seed(c["seed"])
    log = getattr(tr, fn)(gpu_batch(golden_phases))
    for k, v in c["scalars"].items():
        assert abs(log[k] - v) <= TOL * max(abs(v), 1e-3), (name, k, log[k], v)
    for n in trained:
        assert_grads(grads_of(nets[n]), c["grads"][n], 2e-4, name + "." + n)
        assert_post_step(list(nets[n].named_parameters()), c["post"][n], cfg.learning_rate, name + ".post." + n)


@pytest.mark.parametrize("name,fn,trained", PHASES[:3])
def test_phase_bf16_losses(ops, golden_phases, name, fn, trained):
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg = MSSRConfig.small()
    ops.set_compute_dtype(torch.bfloat16)
    try:
        nets = make_nets(golden_phases, cfg)
        tr = Trainer(nets, cfg)
        c = golden_phases[name]
        torch.manual_seed(c["seed"])
        log = getattr(tr, fn)(gpu_batch(golden_phases))
        for k, v in c["scalars"].items():
            if k.endswith("_acc") or k.endswith("Wgap") or k.endswith("_cost"):
                continue            # accuracies are argmax counts; gaps/costs are differences of gains
            assert abs(log[k] - v) <= 2e-2 * max(abs(v), 1e-2), (name, k, log[k], v)
        for n in trained:
            for p in nets[n].parameters():
                assert p.grad is None or (p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all())
    finally:
        ops.set_compute_dtype(torch.float32)


def test_soft_decode_and_teacher_step(ops, golden_phases):
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer, teacher_step_loss
    cfg = MSSRConfig.small()
    nets = make_nets(golden_phases, cfg)
    tr = Trainer(nets, cfg)
    nets["model"].eval()
    nets["style_generator"].eval()
    batch = gpu_batch(golden_phases)
    labels, _, real_src, real_len, _ = tr.unpack_batch(batch)
    c = golden_phases["decode"]
    with torch.no_grad():
        rep = tr._gen_style_represents(1 - labels, real_src, real_len)
        probs, lens, ids = tr._dynamic_decoding(real_src, real_len, rep, 9)
    assert_close(rep, c["rep"], TOL)
    assert_close(probs, c["probs"], TOL)
    assert torch.equal(ids.cpu(), c["ids"])
    assert torch.equal(lens.cpu(), c["seq_lens"])
    t = golden_phases["teacher_step"]
    teacher = nets["teacher"]
    teacher.train()
    for p in teacher.parameters():
        p.requires_grad_(True)
    for batched in (True, False):       # one call on 3B sentences / the reference's three calls
        teacher.zero_grad()
        loss, lg = teacher_step_loss(teacher, batch, cfg, batched=batched)
        assert_close(loss, t["loss"], TOL)
        assert_close(lg, t["logits"], TOL)
        loss.backward()
        assert_grads(grads_of(teacher), t["g"], 2e-4, "teacher (batched=%s)" % batched)


def test_greedy_translate_matches_reference_tokens(ops, golden_phases, golden_greedy):
    """BASELINE config 4 (greedy decode, beam 1) on the decode engine: predicted token ids per sentence equal
    the reference's (GreedySearch imported from the reference when the fixture was made), fp32: exact."""
    from mssrnet_style_transfer_amd.translate import greedy_translate
    cfg = MSSRConfig.small()
    nets = make_nets(golden_phases, cfg)
    with torch.no_grad():
        nets["model"].generator[0].bias[cfg.eos_idx] += golden_greedy["eos_bias"]
    g = golden_greedy
    preds = greedy_translate(nets["model"], nets["style_generator"], cfg, g["src"].cuda(), g["lens"].cuda(),
                             g["desired"].cuda(), g["max_length"])
    assert preds == g["predictions"]
    ops.set_compute_dtype(torch.bfloat16)
    try:
        preds16 = greedy_translate(nets["model"], nets["style_generator"], cfg, g["src"].cuda(), g["lens"].cuda(),
                                   g["desired"].cuda(), g["max_length"])
    finally:
        ops.set_compute_dtype(torch.float32)
    # bf16: token match >= 99 % is the north_star bar at real scale; on 16 short random-weight sentences
    # (near-uniform distributions) demand that most sentences agree on their first token and length
    same_first = sum(a[0] == b[0] for a, b in zip(preds16, g["predictions"]))
    assert same_first >= 12, (same_first, preds16, g["predictions"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_greedy_batch_compaction_keeps_every_sentence(ops, dtype):
    """Finished sentences leave the batch at the check points (translator_wh.py:744-746 on the engine: survivors'
    memory bank, cache rows and last tokens gathered into a smaller plan): every sentence's tokens up to its first
    </s> are the ones the uncompacted sweep produces -- bit-identical work per sentence, so exact in both dtypes."""
    from mssrnet_style_transfer_amd.decode_engine import greedy_decode_ids
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    cfg = MSSRConfig.small()
    ops.set_compute_dtype(dtype)
    try:
        nets = build_all(cfg, device="cuda", seed=21)
        model = nets["model"].eval()
        B, Ls, T = 96, 9, 40
        torch.manual_seed(4)
        mem = torch.randn(Ls, B, cfg.d_model, device="cuda")
        lens = torch.randint(3, Ls + 1, (B,), device="cuda")
        model.decoder.init_state(torch.zeros(Ls, B, 1, dtype=torch.long, device="cuda"), None, None)
        for bump in (0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.2, 0.2, 0.2, 0.3, 0.3, 0.5):     # raise </s> until sentences end
            with torch.no_grad():                                                    # at scattered steps
                model.generator[0].bias[cfg.eos_idx] += bump
            plain = greedy_decode_ids(model, cfg, mem, lens, T, compact_below=0.0)
            ends = (plain == cfg.eos_idx).int().argmax(0)
            has = (plain == cfg.eos_idx).any(0)
            if int(has.sum()) >= B // 3 and len(set(ends[has].tolist())) >= 4:
                break
        else:
            raise AssertionError("fixture: sentences must end at different steps")
        packed = greedy_decode_ids(model, cfg, mem, lens, T, compact_below=0.95)
        for b in range(B):
            n = int(ends[b]) + 1 if bool(has[b]) else plain.shape[0]
            got = packed[:, b].tolist()
            want = plain[:n, b].tolist()
            assert got[:min(n, len(got))] == want[:len(got)] and (len(got) >= n or all(t == cfg.eos_idx for t in want[len(got):])), b
    finally:
        ops.set_compute_dtype(torch.float32)


def test_long_sequence_stress_phases_bf16(ops):
    """BASELINE config 5 shape class (128-word sentences, T = 130 / 140 decode steps), reduced to 2 layers and
    bs 6 so it runs in seconds: all four phases run end to end in bf16 with finite losses and gradients
    (attention beyond 64 keys takes the composed path, the decode engine its blocked step attention)."""
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    from mssrnet_style_transfer_amd.synth import make_batch
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg = MSSRConfig(layers=2, cls=(1, 8, 32), sgen=(1, 8, 32), vocab_size=1000, dropout=0.1, attention_dropout=0.1)
    ops.set_compute_dtype(torch.bfloat16)
    try:
        nets = build_all(cfg, device="cuda", seed=9)
        b = make_batch(6, 128, cfg.vocab_size, seed=9, device="cuda", min_words=64)
        tr = Trainer(nets, cfg)
        for fn in ("train_discriminator", "train_generator", "train_reconstruct", "init_launch"):
            log = getattr(tr, fn)(b)
            assert all(v == v and abs(v) < 1e6 for v in log.values()), (fn, log)
        for n in ("model", "style_generator", "text_discriminator", "style_discriminator"):
            for p in nets[n].parameters():
                assert p.grad is None or torch.isfinite(p.grad).all(), n
    finally:
        ops.set_compute_dtype(torch.float32)


def test_two_consecutive_steps_follow_the_oracle(ops):
    """D phase then G phase with the optimizer steps in between (what smoke() runs): the weight caches (bf16
    shadows, packed engine weights) must follow the fused-Adam updates, which do not bump Parameter._version."""
    import __graft_entry__ as g
    g.smoke()


def test_training_loop_learns_and_keeps_memory_flat(ops):
    """Forty optimizer steps of the reconstruct / discriminator / generator schedule in bf16 on two alternating
    batches of different shape: the reconstruction loss must fall (weights, packed-weight caches and optimizer state
    stay coherent across steps and shapes) and device memory must not grow."""
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    from mssrnet_style_transfer_amd.synth import make_batch
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    ops.set_compute_dtype(torch.bfloat16)
    cfg = MSSRConfig.small(learning_rate=1e-3)
    nets = build_all(cfg, device="cuda", seed=5)
    tr = Trainer(nets, cfg)
    batches = [make_batch(12, 9, cfg.vocab_size, seed=1, device="cuda"), make_batch(8, 6, cfg.vocab_size, seed=2, device="cuda")]
    first, last, mem = {}, {}, []
    for i in range(40):
        b = batches[i % 2]
        r = tr.train_reconstruct(b)
        if i % 4 == 3:
            tr.train_discriminator(b)
            g = tr.train_generator(b)
            assert all(v == v for v in g.values())
        first.setdefault(i % 2, r["T_CONSTRUCT_R_VL"])
        last[i % 2] = r["T_CONSTRUCT_R_VL"]
        if i in (9, 39):
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
    ops.set_compute_dtype(torch.float32)
    for k in (0, 1):
        assert last[k] < 0.8 * first[k], (k, first[k], last[k])
    assert mem[1] <= mem[0] * 1.05 + (1 << 20), mem


def test_decoder_map_state_compacts_the_cache(ops):
    """The reference's greedy loop drops finished sentences with decoder.map_state(index_select)
    (translator_wh.py:744-746, decoders/transformer.py:274-287): after two cached steps on five sentences the state is
    cut down to three of them and decoding continues; the result must equal decoding those three from the start."""
    from mssrnet_style_transfer_amd import style_generator_builder as SB
    from mssrnet_style_transfer_amd.train_and_translate import convert_to_onehot, generator_probs
    cfg = MSSRConfig.small()
    torch.manual_seed(9)
    model = SB.build_main_model(cfg).cuda().eval()
    B, Ls = 5, 6
    lens = torch.tensor([6, 6, 5, 4, 3], device="cuda")
    src = torch.randint(9, cfg.vocab_size, (Ls, B, 1), device="cuda")
    mem = torch.randn(Ls, B, cfg.d_model, device="cuda")
    keep = torch.tensor([0, 2, 3], device="cuda")

    def steps(src_, mem_, lens_, n, select_after=None):
        model.decoder.init_state(src_, mem_, None)
        bos = torch.full((lens_.numel(),), cfg.bos_idx, dtype=torch.long, device="cuda")
        inp = convert_to_onehot(bos, cfg.vocab_size, "cuda").unsqueeze(0)
        outs = []
        for s in range(n):
            dec_out, _ = model.decoder(inp, mem_, memory_lengths=lens_, step=s)
            probs = generator_probs(model, dec_out, cfg.dec_temperature)
            outs.append(probs[-1])
            inp = probs[-1:]
            if select_after is not None and s == select_after:
                model.decoder.map_state(lambda t, dim: t.index_select(dim, keep))
                mem_, lens_, inp = mem_.index_select(1, keep), lens_.index_select(0, keep), inp.index_select(1, keep)
                assert model.decoder.state["src"].size(1) == keep.numel()
                for layer_cache in model.decoder.state["cache"].values():
                    for k, v in layer_cache.items():
                        assert v is None or v.size(0) == keep.numel(), k
        return outs

    with torch.no_grad():
        cut = steps(src, mem, lens, 5, select_after=1)
        ref = steps(src.index_select(1, keep), mem.index_select(1, keep), lens.index_select(0, keep), 5)
    for s in range(5):
        a = cut[s] if s > 1 else cut[s].index_select(0, keep)
        assert_close(a, ref[s], 1e-4, "step %d" % s)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_wgan_step_with_shared_decode_sweep_equals_the_two_phases(ops, golden_phases, dtype):
    """Trainer.train_wgan_step (the critics' fake text decoded in eval mode in the same sweep as the generator
    phase's texts: mssr_decode_plan.eval_rows / bwd_row0) against train_discriminator + train_generator called one
    after the other: same logged scalars, same gradients of all four networks, same parameters after the two
    optimizer steps (dropout 0 -> the only differences are the row grouping of the launches and, in bf16, the
    segmented reduction of the weight-gradient products, mssr_gemm_desc.k_seg_len, and the eval-mode sentences riding
    through the encoders' saving plan (separate LayerNorm) instead of the forward-only one (LayerNorm folded into the
    projections): bf16 scalars are held to 2e-3 of the gains' scale (sigmoid gains are O(1); T_Wgap / S_Wgap are
    differences of two of them)."""
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    ops.set_compute_dtype(dtype)
    try:
        cfg = MSSRConfig.small()
        runs = []
        for shared in (False, True):
            nets = make_nets(golden_phases, cfg.with_(share_decode_sweep=shared))
            tr = Trainer(nets, cfg.with_(share_decode_sweep=shared))
            torch.manual_seed(5)
            random.seed(5)
            d_log, g_log = tr.train_wgan_step(gpu_batch(golden_phases))
            runs.append((dict(d_log), dict(g_log), nets))
    finally:
        ops.set_compute_dtype(torch.float32)
    (d0, g0, n0), (d1, g1, n1) = runs
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    floor = 1e-3 if dtype == torch.float32 else 1.0
    for k, v in d0.items():
        # bf16: the penalties square a norm of second-order quantities -- 3.5e-3 between the two runs measured on T_gp
        # (both runs are held to 2e-2 of the reference elsewhere)
        tk = 5e-3 if dtype != torch.float32 and k.endswith(("_gp", "_cost")) else tol
        assert abs(d1[k] - v) <= tk * max(abs(v), floor), ("D", k, d1[k], v)
    for k, v in g0.items():
        assert abs(g1[k] - v) <= tol * max(abs(v), floor), ("G", k, g1[k], v)
    for name in ("text_discriminator", "style_discriminator", "model", "style_generator"):
        # bf16: two roundings of the same mathematics (see the docstring) feed five decode steps and the critics'
        # second-order graph at 64-wide layers: up to 6.4e-2 between the two runs on a critic weight gradient (fp32:
        # 2e-5, which is what pins the logic); each bf16 run is held to 2e-2 of the reference at README widths
        # elsewhere (test_gpu_parity_readme.py)
        assert_grads(grads_of(n1[name]), {k: g for k, g in grads_of(n0[name]).items()},
                     2 * tol if dtype == torch.float32 else 1e-1, "shared." + name,
                     floor_rel=1e-5 if dtype == torch.float32 else 3e-2)      # of the network's largest gradient
        # Adam turns a mathematically-zero gradient (rounding noise, e.g. the key-projection bias) into a step of
        # arbitrary sign: every element within 2 lr, as helpers.assert_post_step allows
        for (k, a), (_, b) in zip(n0[name].named_parameters(), n1[name].named_parameters()):
            assert float((a.detach() - b.detach()).abs().max()) <= 2.05 * cfg.learning_rate + 1e-7, "post-step %s.%s" % (name, k)
    assert set(golden_phases["discriminator"]["scalars"]) <= set(d1)


def test_mixed_eval_train_sweep_with_dropout(ops):
    """eval_rows sentences of a mixed sweep must be decoded WITHOUT dropout (bit-identical to an eval-mode decode of
    the same sentences) while the rest are dropped, and the backward over the differentiable sentences must replay
    exactly the masks their forward drew (finite-difference check of a seed-fixed loss along a random direction)."""
    from mssrnet_style_transfer_amd import style_generator_builder as SB
    from mssrnet_style_transfer_amd import ops as O
    from mssrnet_style_transfer_amd.decode_engine import soft_decode
    ops.set_compute_dtype(torch.float32)
    cfg = MSSRConfig.small(dropout=0.3, attention_dropout=0.2)
    torch.manual_seed(11)
    model = SB.build_main_model(cfg).cuda()
    Ls, E, G, T = 6, 3, 4, 5
    mem = torch.randn(Ls, E + G, cfg.d_model, device="cuda")
    lens = torch.tensor([6, 5, 4, 6, 6, 3, 2], device="cuda")
    model.eval()
    with torch.no_grad():
        want_eval = soft_decode(model, cfg, mem[:, :E].contiguous(), lens[:E], T)
    model.train()

    def run(m):
        O._state["seed_counter"] = 4321         # same dropout seeds on every call
        return soft_decode(model, cfg, m, lens, T, eval_rows=E)

    with torch.no_grad():
        both = run(mem)
        assert torch.equal(run(mem), both), "same seed must give the same masks"
    assert torch.equal(both[:, :E], want_eval)
    model.eval()
    with torch.no_grad():
        all_eval = soft_decode(model, cfg, mem, lens, T)
    model.train()
    assert float((both[:, E:] - all_eval[:, E:]).abs().max()) > 1e-4          # the other sentences ARE dropped
    w = torch.randn(T, E + G, cfg.vocab_size, device="cuda")
    m1 = mem.clone().requires_grad_(True)
    loss = (run(m1) * w).sum()
    g, = torch.autograd.grad(loss, m1)
    assert float(g[:, :E].abs().max()) == 0.0                                  # eval sentences carry no gradient
    # along the gradient itself: the best conditioned finite difference (see test_engine_dropout_backward_replays_masks)
    direction = g / g.norm() * float(g[:, E:].numel()) ** 0.5
    analytic = float((g * direction).sum())
    # (ReLU / dropout kinks make the difference quotient drift at larger steps: 2e-3 is off by up to 6 % here, 5e-4 by 3e-4)
    eps = 5e-4
    with torch.no_grad():
        fd = float((((run(mem + eps * direction) - run(mem - eps * direction)) * w).sum() / (2 * eps)).double())
    assert abs(fd - analytic) <= 1e-2 * max(abs(fd), 1e-3), (fd, analytic)


def test_unknown_label_token_is_reported_when_the_log_is_read(ops, golden_phases):
    """The reference's convert_label_idx raises KeyError for a first token outside label_tokens
    (onmt/train_and_translate.py:60-61); here the lookup stays on the device and the error surfaces with the log."""
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg = MSSRConfig.small()
    tr = Trainer(make_nets(golden_phases, cfg), cfg)
    b = gpu_batch(golden_phases)
    b.src = b.src.clone()
    b.src[0, 1, 0] = 9                        # a word where the style label token belongs
    log = tr.init_launch(b)
    with pytest.raises(KeyError, match="label_token_ids"):
        log["INIT_SL"]
    assert tr.init_launch(gpu_batch(golden_phases))["INIT_SL"] > 0       # the flag does not stick
