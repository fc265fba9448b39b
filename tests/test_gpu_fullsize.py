"""Full-size checks (BASELINE.json configs[1] and configs[4] as bench.py runs them), VERDICT round-1 item 1(b).

* the README bs80 WGAN step (12 layers, d512, V=9606, <=16 words) in fp32 against ``oracle.phases`` run on the
  box's CPU with the same weights, batch and interpolation draws: every logged loss at 1e-4, every parameter
  gradient of the four trained networks at 1e-4 of the network's largest gradient (measured: 1.7e-5 at worst;
  the fixtures of tests/golden pin the oracle itself against the reference);
* one full-size configs[4] step (12 layers, bs256, 64..128 words, bf16, dropout on): size-independent properties --
  finite losses and gradients, soft words that sum to one, and a bound on device memory.
"""
from types import SimpleNamespace

import pytest
import torch

from mssrnet_style_transfer_amd.config import MSSRConfig

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    from mssrnet_style_transfer_amd import ops as o
    o.set_compute_dtype(torch.float32)
    yield o
    o.set_compute_dtype(torch.float32)


def assert_net_grads(net, want, name, tol=1e-4, tol_tensor=5e-2):
    """Every element of every gradient within ``tol`` x the NETWORK's largest gradient of the oracle's (the convention of
    tests/test_gpu_parity_readme.py); additionally each tensor within ``tol_tensor`` of its own largest element --
    both sides are fp32 sums over 80 sentences in different orders, and the softmax-pooling and second-order
    gradients are differences of much larger terms (measured: 1e-2 on a tensor 50x below the network's largest)."""
    scale = max(float(w.abs().max()) for w in want.values() if w is not None)
    worst = (0.0, None)
    for k, p in net.named_parameters():
        w = want.get(k)
        if w is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, (name, k)
            continue
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), (name, k)
        err = float((p.grad.detach().cpu().double() - w.double()).abs().max())
        worst = max(worst, (err / scale, k))
        assert err <= tol * scale, "%s.%s: err %.3e > %.1e x largest gradient %.3e" % (name, k, err, tol, scale)
        assert err <= tol_tensor * float(w.abs().max()) + 1e-5 * scale, "%s.%s: err %.3e vs own max %.3e" % (
            name, k, err, float(w.abs().max()))
    print("%s: worst error %.2e of the largest gradient (%s)" % (name, worst[0], worst[1]))


def test_readme_bs80_step_against_the_oracle_fp32(ops):
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    from mssrnet_style_transfer_amd.synth import make_batch
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    from oracle import phases as PH
    cfg = MSSRConfig(dropout=0.0, attention_dropout=0.0)            # README widths, 12 layers
    nets = build_all(cfg, device="cuda", seed=123)
    states = {k: {n: v.detach().cpu() for n, v in m.state_dict().items()} for k, m in nets.items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    onets = PH.make_nets(states, cfg)
    batch = make_batch(80, 16, cfg.vocab_size, seed=123)
    gbatch = SimpleNamespace(src=batch.src.cuda(), src_lengths=batch.src_lengths.cuda(), tgt=batch.tgt.cuda(),
                             labels=batch.labels.cuda())
    tr = Trainer(nets, cfg)
    trained = {"train_discriminator": ("text_discriminator", "style_discriminator"),
               "train_generator": ("model", "style_generator")}
    for name, who in trained.items():
        torch.manual_seed(17)
        got = getattr(tr, name)(gbatch)
        torch.manual_seed(17)
        want = getattr(PH, name)(onets, batch, cfg)
        for k, v in want.items():
            assert abs(got[k] - v) <= 1e-4 * max(abs(v), 1e-3), (name, k, got[k], v)
        for n in who:
            assert_net_grads(nets[n], onets[n].grads(), name + "." + n)


def test_config5_full_size_step_properties_bf16(ops):
    from mssrnet_style_transfer_amd.decode_engine import soft_decode
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    from mssrnet_style_transfer_amd.synth import make_batch
    from mssrnet_style_transfer_amd.trainer_wgan_wh import Trainer
    cfg = MSSRConfig()                                               # 12 layers, dropout 0.1
    ops.set_compute_dtype(torch.bfloat16)
    nets = build_all(cfg, device="cuda", seed=5)
    b = make_batch(256, 128, cfg.vocab_size, seed=5, device="cuda", min_words=64)
    tr = Trainer(nets, cfg)
    torch.cuda.reset_peak_memory_stats()
    d, g = tr.train_wgan_step(b)
    for log in (d, g):
        assert all(v == v and abs(v) < 1e6 for v in log.values()), log
    for n in ("model", "style_generator", "text_discriminator", "style_discriminator"):
        for k, p in nets[n].named_parameters():
            assert p.grad is None or (p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all())), (n, k)
    # 128-word sentences at bs256: the step must stay far inside one GPU's 288 GB (measured ~60 GB)
    assert torch.cuda.max_memory_allocated() < 120 * (1 << 30), torch.cuda.max_memory_allocated()
    # soft words of a short eval decode at the same batch shape are distributions
    with torch.no_grad():
        nets["model"].eval()
        mem = torch.randn(128, 256, cfg.d_model, device="cuda")
        lens = torch.full((256,), 128, device="cuda", dtype=torch.long)
        probs = soft_decode(nets["model"], cfg, mem, lens, 4)
        nets["model"].train()
    probs = probs[0] if isinstance(probs, (tuple, list)) else probs
    s = probs.float().sum(-1)
    assert float((s - 1).abs().max()) < 2e-2, float((s - 1).abs().max())
