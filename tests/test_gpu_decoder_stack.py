"""The teacher-forced decoder-stack engine (csrc/decoder_stack.hip via mssr_decstack_fwd / mssr_decstack_bwd: masked
self attention, context attention over the memory bank, FFN, all layers + final LayerNorm as one autograd node) against
the same stack looped over the modules (TransformerDecoderLayer.forward, the drop-in surface of
onmt/decoders/transformer.py:75-171, which the golden tests pin to the reference; reached here through
ops.second_order()).  fp32: 1e-4 on outputs, 2e-4 on gradients; bf16 at README shapes: 2e-2; dropout: the backward
must replay the forward's masks (directional finite difference under a fixed seed)."""
import pytest
import torch

from helpers import assert_close, assert_grads
from mssrnet_style_transfer_amd.config import MSSRConfig

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    from mssrnet_style_transfer_amd import ops as o
    o.set_compute_dtype(torch.float32)
    yield o
    o.set_compute_dtype(torch.float32)


def make(cfg, B, T, Ls, seed=0):
    from mssrnet_style_transfer_amd import style_generator_builder as SB
    torch.manual_seed(seed)
    model = SB.build_main_model(cfg).cuda()
    dec = model.decoder
    lens = torch.randint(2, Ls + 1, (B,))
    lens[0] = Ls
    lens = lens.sort(descending=True).values.cuda()
    src = torch.randint(9, cfg.vocab_size, (Ls, B, 1)).cuda()
    tgt = torch.randint(9, cfg.vocab_size, (T, B, 1)).cuda()
    tgt[T - 2:, B - 1] = cfg.pad_idx                       # target padding inside the self-attention mask
    mem = torch.randn(Ls, B, cfg.d_model, device="cuda")
    return model, dec, src, tgt, lens, mem


def run(dec, ops, src, tgt, lens, mem, w, wa, modular):
    from mssrnet_style_transfer_amd import decoder_stack as DS
    dec.zero_grad()
    m = mem.clone().requires_grad_(True)
    dec.init_state(src, m, None)
    calls = []
    orig = DS._DecoderStack.apply

    def counted(*a):
        calls.append(1)
        return orig(*a)

    DS._DecoderStack.apply = counted
    try:
        if modular:
            with ops.second_order():
                out, attns = dec(tgt, m, memory_lengths=lens)
        else:
            out, attns = dec(tgt, m, memory_lengths=lens)
    finally:
        DS._DecoderStack.apply = orig
    assert len(calls) == (0 if modular else 1), "engine taken: %d calls" % len(calls)
    (out.float() * w).sum().backward()
    grads = {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None}
    return out.detach(), attns["std"].detach(), m.grad.clone(), grads


@pytest.mark.parametrize("train", [False, True])
def test_decoder_stack_matches_modular_fp32(ops, train):
    cfg = MSSRConfig.small()
    B, T, Ls = 5, 8, 7
    model, dec, src, tgt, lens, mem = make(cfg, B, T, Ls)
    dec.train(train)                        # dropout p = 0 in the small config: both modes are deterministic
    w = torch.randn(T, B, cfg.d_model, device="cuda")
    oe, ae, ge_mem, ge = run(dec, ops, src, tgt, lens, mem, w, None, False)
    om, am, gm_mem, gm = run(dec, ops, src, tgt, lens, mem, w, None, True)
    assert_close(oe, om, 1e-4, "decoder output")
    assert_close(ae, am, 1e-4, "context attention of the last layer, head 0")
    assert_close(ge_mem, gm_mem, 2e-4, "d memory_bank")
    assert set(ge) == set(gm) and len(ge) == 2 + 26 * cfg.layers + 1          # + the embedding table
    assert_grads(ge, gm, 2e-4, "decoder")


def test_decoder_stack_soft_targets_and_no_grad(ops):
    """Soft target words (probabilities [T, B, V]: the cycle decoder pass) and a forward-only plan."""
    cfg = MSSRConfig.small()
    B, T, Ls = 4, 6, 5
    model, dec, src, _, lens, mem = make(cfg, B, T, Ls, seed=1)
    dec.eval()
    soft = torch.softmax(torch.randn(T, B, cfg.vocab_size, device="cuda") * 3, -1)
    dec.init_state(src, mem, None)
    with torch.no_grad():
        oe, ae = dec(soft, mem, memory_lengths=lens)
        with ops.second_order():
            om, am = dec(soft, mem, memory_lengths=lens)
    assert not oe.requires_grad
    assert_close(oe, om, 1e-4, "soft targets, forward-only plan")
    assert_close(ae["std"], am["std"], 1e-4, "attention")


def test_decoder_stack_dropout_backward_replays_masks(ops):
    cfg = MSSRConfig.small(dropout=0.3, attention_dropout=0.3)
    B, T, Ls = 4, 6, 5
    model, dec, src, tgt, lens, mem = make(cfg, B, T, Ls, seed=3)
    dec.train()
    dec.embeddings.make_embedding[1].dropout.p = 0.0        # embedding dropout sits outside the engine
    w = torch.randn(T, B, cfg.d_model, device="cuda")
    v = torch.randn_like(mem) * 3e-4
    dec.init_state(src, mem, None)

    def f(m):
        ops._state["seed_counter"] = 4321
        return dec(tgt, m, memory_lengths=lens)[0]

    m = mem.clone().requires_grad_(True)
    out = f(m)
    (out * w).sum().backward()
    an = float((m.grad * v).sum())
    with torch.no_grad():
        assert torch.equal(f(mem), out.detach())
        dec.eval()
        assert float((f(mem) - out.detach()).abs().max()) > 1e-5
        dec.train()
        fd = float(((f(mem + v).double() - f(mem - v).double()) * w.double()).sum() / 2)
    assert abs(an - fd) <= 6e-2 * max(abs(fd), 1e-3), (an, fd)


def test_decoder_stack_bf16_readme_shapes_and_shared_pack(ops):
    """README widths (d512, 8 x 64 heads, ff 1024; 3 layers to stay small) in bf16 against the fp32 modular path, and the
    pack: once a soft decode has claimed the decoder, the stack reads the MODEL's pack (one refresh serves both)."""
    from mssrnet_style_transfer_amd import decode_engine as DE
    from mssrnet_style_transfer_amd import decoder_stack as DS
    from mssrnet_style_transfer_amd import train_and_translate as TT
    cfg = MSSRConfig()
    cfg.layers = 3
    B, T, Ls = 6, 18, 17
    model, dec, src, tgt, lens, mem = make(cfg, B, T, Ls, seed=5)
    dec.eval()
    w = torch.randn(T, B, cfg.d_model, device="cuda")
    om, am, gm_mem, gm = run(dec, ops, src, tgt, lens, mem, w, None, True)           # fp32 modular
    ops.set_compute_dtype(torch.bfloat16)
    oe, ae, ge_mem, ge = run(dec, ops, src, tgt, lens, mem, w, None, False)
    assert_close(oe.float(), om, 2e-2, "decoder output bf16")
    assert_close(ae.float(), am, 2e-2, "attention bf16")
    assert_close(ge_mem.float(), gm_mem, 6e-2, "d memory_bank bf16")      # max-abs over three bf16 layers: 3.5-4.5e-2 by seed
    assert_grads(ge, gm, 4e-2, "decoder bf16", floor_rel=2e-2)
    assert DE.model_of(dec) is None
    private = DS._pack_for(dec, DS.stack_params(dec))
    with torch.no_grad():
        TT.decode_batch_ori(model, cfg, src, None, mem, lens, 3)
    assert DE.model_of(dec) is model
    shared = DS._pack_for(dec, DS.stack_params(dec))
    assert shared is DE.get_pack(model, DE.decoder_params(model)) and shared is not private
    oe2, _, _, _ = run(dec, ops, src, tgt, lens, mem, w, None, False)
    assert torch.equal(oe2, oe)
