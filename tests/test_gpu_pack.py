"""Weight packs (csrc/pack.hip, weight_pack.JobTable): every derived buffer of the decode / encoder engines against
the torch expression it stands for -- cast copies, packed Q|K|V rows, transposes, LayerNorm-folded weights with the
row sums of the ROUNDED values and W beta + b (the op chains the reference applies in place:
onmt/decoders/transformer.py:110-171, onmt/encoders/transformer.py:37-52, onmt/modules/position_ffn.py:35-37) --
plus the in-place refresh after an optimizer step and its guard while a saved graph still reads the buffers."""
import pytest
import torch

from helpers import assert_close
from mssrnet_style_transfer_amd.config import MSSRConfig

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    from mssrnet_style_transfer_amd import ops as o
    o.set_compute_dtype(torch.float32)
    yield o
    o.set_compute_dtype(torch.float32)


def _randomize(module, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, device="cuda", generator=g) * 0.3 + (1.0 if p.dim() == 1 else 0.0))


def _check_decoder_pack(pk, model, cd):
    dec, gen = model.decoder, model.generator[0]
    d, V, Vp = pk.d, pk.V, pk.Vp
    tol = 0.0
    for l, layer in enumerate(dec.transformer_layers):
        sa, ca, ff = layer.self_attn, layer.context_attn, layer.feed_forward
        Wqkv = torch.cat([sa.linear_query.weight, sa.linear_keys.weight, sa.linear_values.weight], 0).detach()
        bqkv = torch.cat([sa.linear_query.bias, sa.linear_keys.bias, sa.linear_values.bias], 0).detach()
        assert torch.equal(pk.w_qkv[l], Wqkv.to(cd)) and torch.equal(pk.wt_qkv[l], Wqkv.to(cd).t())
        assert torch.equal(pk.b_qkv[l], bqkv)
        for name, lin in (("so", sa.final_linear), ("cq", ca.linear_query), ("co", ca.final_linear), ("1", ff.w_1),
                          ("2", ff.w_2)):
            assert torch.equal(getattr(pk, "w_" + name)[l], lin.weight.detach().to(cd)), name
            assert torch.equal(getattr(pk, "wt_" + name)[l], lin.weight.detach().to(cd).t()), name
            assert torch.equal(getattr(pk, "b_" + name)[l], lin.bias.detach()), name
        Wckv = torch.cat([ca.linear_keys.weight, ca.linear_values.weight], 0).detach()
        assert torch.equal(pk.w_ckv[2 * l * d:(2 * l + 2) * d], Wckv.to(cd)) and torch.equal(pk.wt_ckv[l], Wckv.to(cd).t())
        assert torch.equal(pk.b_ckv[2 * l * d:(2 * l + 2) * d], torch.cat([ca.linear_keys.bias, ca.linear_values.bias]).detach())
        for name, ln in (("ln1", layer.layer_norm_1), ("ln2", layer.layer_norm_2), ("lnf", ff.layer_norm)):
            assert torch.equal(getattr(pk, name + "_g")[l], ln.weight.detach())
            assert torch.equal(getattr(pk, name + "_b")[l], ln.bias.detach())
        if pk.fuse_ln:
            for name, W, b, ln in (("qkv", Wqkv, bqkv, layer.layer_norm_1), ("cq", ca.linear_query.weight.detach(),
                                                                            ca.linear_query.bias.detach(), layer.layer_norm_2),
                                   ("1", ff.w_1.weight.detach(), ff.w_1.bias.detach(), ff.layer_norm)):
                wg = (W * ln.weight.detach()).to(cd)
                assert torch.equal(getattr(pk, "wg_" + name)[l], wg), name
                assert_close(getattr(pk, "cs_" + name)[l], wg.double().sum(-1).float(), 1e-6, "cs_" + name, floor=1e-5)
                assert_close(getattr(pk, "bf_" + name)[l], (W.double() @ ln.bias.detach().double() + b.double()).float(), 1e-6,
                             "bf_" + name, floor=1e-5)
    E, Wg = dec.embeddings.word_lut.weight.detach(), gen.weight.detach()
    assert torch.equal(pk.emb, E.to(cd)) and torch.equal(pk.emb_t[:, :V], E.to(cd).t())
    assert torch.equal(pk.w_gen, Wg.to(cd)) and torch.equal(pk.wt_gen[:, :V], Wg.to(cd).t())
    assert float(pk.emb_t[:, V:].abs().sum()) == 0.0 and float(pk.wt_gen[:, V:].abs().sum()) == 0.0
    assert torch.equal(pk.b_gen, gen.bias.detach())
    if pk.fuse_ln:
        wg = (Wg * dec.layer_norm.weight.detach()).to(cd)
        assert torch.equal(pk.wg_gen, wg)
        assert_close(pk.cs_gen, wg.double().sum(-1).float(), 1e-6, "cs_gen", floor=1e-5)
        assert_close(pk.bf_gen, (Wg.double() @ dec.layer_norm.bias.detach().double() + gen.bias.detach().double()).float(),
                     1e-6, "bf_gen", floor=1e-5)
    return tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("readme", [False, True])
def test_decoder_and_encoder_packs_match_torch(ops, dtype, readme):
    from mssrnet_style_transfer_amd import decode_engine as DE
    from mssrnet_style_transfer_amd import encoder_engine as EE
    from mssrnet_style_transfer_amd import style_generator_builder as SB
    ops.set_compute_dtype(dtype)
    cfg = MSSRConfig() if readme else MSSRConfig.small()
    if readme:
        cfg.layers = 2
    torch.manual_seed(0)
    model = SB.build_main_model(cfg).cuda()
    _randomize(model, 1)
    pk = DE.get_pack(model, DE.decoder_params(model))
    assert pk.fuse_ln == (dtype == torch.bfloat16 and cfg.d_model % 32 == 0)
    _check_decoder_pack(pk, model, dtype)
    enc = model.encoder
    ek = EE.get_pack(enc, EE.encoder_params(enc))
    for l, layer in enumerate(enc.transformer):
        sa, ff = layer.self_attn, layer.feed_forward
        Wqkv = torch.cat([sa.linear_query.weight, sa.linear_keys.weight, sa.linear_values.weight], 0).detach()
        assert torch.equal(ek.w_qkv[l], Wqkv.to(dtype)) and torch.equal(ek.wt_qkv[l], Wqkv.to(dtype).t())
        assert torch.equal(ek.w_o[l], sa.final_linear.weight.detach().to(dtype))
        assert torch.equal(ek.wt_2[l], ff.w_2.weight.detach().to(dtype).t())
        assert torch.equal(ek.b_1[l], ff.w_1.bias.detach()) and torch.equal(ek.ln1_g[l], layer.layer_norm.weight.detach())
        if ek.fuse_ln:
            wg = (ff.w_1.weight.detach() * ff.layer_norm.weight.detach()).to(dtype)
            assert torch.equal(ek.wg_1[l], wg)
            assert_close(ek.cs_1[l], wg.double().sum(-1).float(), 1e-6, "cs_1", floor=1e-5)
            assert_close(ek.bf_1[l], (ff.w_1.weight.detach().double() @ ff.layer_norm.bias.detach().double()
                                      + ff.w_1.bias.detach().double()).float(), 1e-6, "bf_1", floor=1e-5)
    assert torch.equal(ek.lnF_g, enc.layer_norm.weight.detach())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_jobs_ragged_shapes(ops, dtype):
    """Rows and columns that are not multiples of the strip (32) / vector (4, 8) sizes, unaligned transposed rows."""
    from mssrnet_style_transfer_amd.weight_pack import JobTable
    torch.manual_seed(3)
    jt = JobTable(dtype, torch.device("cuda"))
    cases = []
    for rows, cols, ld_t in ((37, 50, 41), (1, 7, 1), (64, 256, 64), (33, 515, 40), (100, 4, 104)):
        W = torch.randn(rows, cols, device="cuda")
        g, b, bias = torch.randn(cols, device="cuda"), torch.randn(cols, device="cuda"), torch.randn(rows, device="cuda")
        dst = torch.zeros(rows, cols + 3, dtype=dtype, device="cuda")
        dst_t = torch.zeros(cols, ld_t, dtype=dtype, device="cuda")
        dst_g = torch.zeros(rows, cols, dtype=dtype, device="cuda")
        cs, bf = torch.zeros(rows, device="cuda"), torch.zeros(rows, device="cuda")
        dst_f = torch.zeros(rows, cols, device="cuda")
        jt.add(W, dst=dst[:, :cols], dst_t=dst_t[:, :rows], dst_g=dst_g, gamma=g, beta=b, bias=bias, colsum=cs, bias_f=bf,
               dst_f=dst_f.view(-1))
        cases.append((W, g, b, bias, dst, dst_t, dst_g, cs, bf, dst_f))
    v = torch.randn(19, device="cuda")
    vf = torch.zeros(19, device="cuda")
    jt.add(v, dst_f=vf)
    jt.finish()
    jt.run()
    assert torch.equal(vf, v)
    for W, g, b, bias, dst, dst_t, dst_g, cs, bf, dst_f in cases:
        rows, cols = W.shape
        assert torch.equal(dst[:, :cols], W.to(dtype)) and float(dst[:, cols:].abs().sum()) == 0.0
        assert torch.equal(dst_t[:, :rows], W.to(dtype).t()) and float(dst_t[:, rows:].abs().sum()) == 0.0
        assert torch.equal(dst_g, (W * g).to(dtype)) and torch.equal(dst_f, W)
        assert_close(cs, (W * g).to(dtype).double().sum(-1).float(), 1e-6, "colsum", floor=1e-5)
        assert_close(bf, (W.double() @ b.double() + bias.double()).float(), 1e-6, "bias_f", floor=1e-5)


def test_pack_refreshes_in_place_unless_a_graph_is_pending(ops):
    from mssrnet_style_transfer_amd import decode_engine as DE
    from mssrnet_style_transfer_amd import train_and_translate as TT
    cfg = MSSRConfig.small()
    torch.manual_seed(4)
    from mssrnet_style_transfer_amd import style_generator_builder as SB
    model = SB.build_main_model(cfg).cuda()
    params = DE.decoder_params(model)
    pk = DE.get_pack(model, params)
    ptr = pk.w_qkv.data_ptr()
    with torch.no_grad():
        for p in model.decoder.parameters():
            p.add_(0.05)
    pk2 = DE.get_pack(model, params)
    assert pk2 is pk and pk2.w_qkv.data_ptr() == ptr                      # same buffers, new values
    _check_decoder_pack(pk2, model, torch.float32)
    # a decode whose backward is still pending holds the buffers: the next version gets its own pack
    B, Ls, steps = 4, 5, 6
    lens = torch.tensor([5, 4, 3, 2], device="cuda")
    src = torch.randint(9, cfg.vocab_size, (Ls, B, 1)).cuda()
    mem = torch.randn(Ls, B, cfg.d_model, device="cuda", requires_grad=True)
    model.eval()
    probs, _, _ = TT.decode_batch_ori(model, cfg, src, None, mem, lens, steps)
    w = torch.randn_like(probs)
    want = torch.autograd.grad((probs * w).sum(), mem, retain_graph=False)[0]      # reference: backward right away
    probs, _, _ = TT.decode_batch_ori(model, cfg, src, None, mem, lens, steps)
    held = DE.get_pack(model, params)
    before = held.w_qkv.clone()
    with torch.no_grad():
        for p in model.decoder.parameters():
            p.add_(0.05)
    fresh = DE.get_pack(model, params)
    assert fresh is not held and torch.equal(held.w_qkv, before)
    got = torch.autograd.grad((probs * w).sum(), mem)[0]                           # reads the held pack
    assert_close(got, want, 1e-6, "backward after the weights moved on")
    with torch.no_grad():
        for p in model.decoder.parameters():
            p.add_(0.05)
    assert DE.get_pack(model, params) is fresh                                     # no graph pending: in place again
