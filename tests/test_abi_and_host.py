"""CPU-side checks: the C-ABI library loads and exports every symbol include/mssr_hip.h declares
(no compute call is made), the binding fails loudly when the library is absent, and the host-side
logic (synthetic batch, length rule) agrees with the oracle."""
import os
import re
import ctypes

import pytest
import torch

from conftest import ROOT
from mssrnet_style_transfer_amd import _lib
from mssrnet_style_transfer_amd.config import MSSRConfig
from mssrnet_style_transfer_amd.synth import make_batch, PAD, BOS, EOS


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mssr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mssr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 18
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libmssr_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "no ctypes signature for %s" % n
    assert set(_lib.SIGNATURES) <= set(names)
    assert _lib.load().mssr_abi_version() == 2


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmssr_hip.so")
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected():
    from mssrnet_style_transfer_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.layer_norm(torch.randn(4, 8), torch.ones(8), torch.zeros(8))


def test_synthetic_batch_layout():
    b = make_batch(80, 16, 9606, seed=123)
    assert b.src.shape == (17, 80, 1) and b.tgt.shape == (18, 80, 1)
    lens = b.src_lengths - 1
    assert int(lens.max()) == 16 and int(lens.min()) >= 3
    assert torch.equal(lens, torch.sort(lens, descending=True)[0])
    assert bool((b.tgt[0] == BOS).all())
    for i in range(80):
        n = int(lens[i])
        assert int(b.tgt[n + 1, i, 0]) == EOS and bool((b.tgt[n + 2:, i, 0] == PAD).all())
        assert bool((b.src[1:n + 1, i, 0] >= 9).all()) and bool((b.src[n + 1:, i, 0] == PAD).all())
        assert int(b.src[0, i, 0]) - 4 == int(b.labels[i])
    assert b.n_tokens == int((lens + 1).sum())
    assert torch.equal(make_batch(80, 16, 9606, seed=123).src, b.src)


def test_length_rule_matches_oracle():
    from mssrnet_style_transfer_amd.train_and_translate import fetch_dec_output_length
    from oracle.phases import fetch_dec_output_length as oracle_rule
    cfg = MSSRConfig.small()
    torch.manual_seed(0)
    for trial in range(20):
        probs = torch.rand(9, 7, 20)
        probs[..., cfg.eos_idx] += (torch.rand(9, 7) > 0.7).float()
        probs[..., cfg.dot_idx] += (torch.rand(9, 7) > 0.8).float()
        for aware in (True, False):
            c = cfg.with_(sent_stop_aware=aware)
            a, ia = fetch_dec_output_length(probs, c)
            b, ib = oracle_rule(probs, c)
            assert a == b and torch.equal(ia, ib)
            assert max(a) == 9


def test_device_length_rule_matches_host_rule():
    """dec_output_length (tensor ops, no host round trip) against fetch_dec_output_length (the reference's Python
    scan, onmt/train_and_translate.py:289-326) on random and edge-case id tables."""
    import torch
    from mssrnet_style_transfer_amd.config import MSSRConfig
    from mssrnet_style_transfer_amd.train_and_translate import dec_output_length, fetch_dec_output_length
    g = torch.Generator().manual_seed(5)
    for aware in (True, False):
        cfg = MSSRConfig.small(sent_stop_aware=aware)
        V = cfg.vocab_size
        for T, B in ((1, 3), (6, 1), (9, 17), (28, 80)):
            for trial in range(6):
                ids = torch.randint(9, V, (T, B), generator=g)
                stops = torch.tensor([cfg.eos_idx, cfg.dot_idx, cfg.excl_idx, cfg.qst_idx])
                put = torch.rand(T, B, generator=g) < (0.0, 0.05, 0.2, 0.5, 1.0, 0.1)[trial]
                ids = torch.where(put, stops[torch.randint(0, 4, (T, B), generator=g)], ids)
                if trial == 5:
                    ids[0] = cfg.eos_idx                  # a stop in the very first position: 'cut <= 1' rule
                probs = torch.nn.functional.one_hot(ids, V).float()
                want, want_ids = fetch_dec_output_length(probs, cfg)
                got, got_ids = dec_output_length(probs, cfg)
                assert got.tolist() == want, (aware, T, B, trial)
                assert torch.equal(got_ids, want_ids)


def test_phase_log_reads_lazily_and_behaves_like_a_dict():
    import torch
    from mssrnet_style_transfer_amd.trainer_wgan_wh import PhaseLog
    a, b = torch.tensor(1.5, requires_grad=True), torch.tensor(0.25)
    log = PhaseLog(("x", "y"), (a * 2, b), lambda d: dict(z=d["x"] - d["y"]))
    assert log._d is None                      # nothing read yet
    assert log["x"] == 3.0 and log["z"] == 2.75
    assert dict(log) == {"x": 3.0, "y": 0.25, "z": 2.75} and len(log) == 3
    assert {**log, "k": 1}["y"] == 0.25


def test_modules_deepcopy_pickle_and_state_dict_roundtrip():
    """The reference checkpoints pickled module objects (trainer_wgan_wh.py:1093-1100), deep-copies models for its
    shadow copies (:129-131) and moves weights with load_state_dict (:1077-1078): the drop-in modules must hold no
    unpicklable handles (library pointers, packed-weight caches live outside the modules)."""
    import copy
    import io
    import pickle
    import torch
    from mssrnet_style_transfer_amd.config import MSSRConfig
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    cfg = MSSRConfig.small()
    nets = build_all(cfg, device="cpu", seed=3)
    for name, m in nets.items():
        twin = copy.deepcopy(m)
        blob = pickle.dumps(m)
        back = pickle.loads(blob)
        buf = io.BytesIO()
        torch.save(m.state_dict(), buf)
        buf.seek(0)
        sd = torch.load(buf, weights_only=True)
        for other in (twin, back):
            assert list(other.state_dict().keys()) == list(m.state_dict().keys()), name
            missing = other.load_state_dict(sd)
            assert not missing.missing_keys and not missing.unexpected_keys, name
            for (k, a), (_, b) in zip(m.state_dict().items(), other.state_dict().items()):
                assert torch.equal(a, b), (name, k)


def test_update_dropout_reaches_every_site():
    """update_dropout(dropout, attention_dropout) of the encoder / decoder (onmt/encoders/transformer.py:138-141,
    onmt/decoders/transformer.py:354-357) must rewrite the rate at every dropout site of every layer."""
    from mssrnet_style_transfer_amd.config import MSSRConfig
    from mssrnet_style_transfer_amd.style_generator_builder import build_main_model
    cfg = MSSRConfig.small(dropout=0.1, attention_dropout=0.1)
    model = build_main_model(cfg)
    model.encoder.update_dropout(0.25, 0.35)
    model.decoder.update_dropout(0.45, 0.55)
    for layer in model.encoder.transformer:
        assert layer.dropout.p == 0.25 and layer.self_attn.dropout.p == 0.35
        assert layer.feed_forward.dropout_1.p == 0.25 and layer.feed_forward.dropout_2.p == 0.25
    assert model.encoder.embeddings.make_embedding.pe.dropout.p == 0.25
    for layer in model.decoder.transformer_layers:
        assert layer.drop.p == 0.45 and layer.self_attn.dropout.p == 0.55 and layer.context_attn.dropout.p == 0.55
        assert layer.feed_forward.dropout_1.p == 0.45 and layer.feed_forward.dropout_2.p == 0.45
    assert model.decoder.embeddings.make_embedding.pe.dropout.p == 0.45


def test_module_surface_matches_the_reference_contract():
    """Attribute and method names the reference's callers reach for: checkpoint keys, the
    isinstance checks of decoders/transformer.py:153,346, StyleGenerator2 reaching into the encoder
    (style_generator.py:22,35-46), update_dropout mutating `.dropout.p`."""
    import torch
    from torch import nn
    from mssrnet_style_transfer_amd.config import MSSRConfig
    from mssrnet_style_transfer_amd.decoders.transformer import TransformerDecoder, TransformerDecoderLayer
    from mssrnet_style_transfer_amd.encoders.transformer import TransformerEncoder, TransformerEncoderLayer
    from mssrnet_style_transfer_amd.modules.embeddings import Embeddings, PositionalEncoding
    from mssrnet_style_transfer_amd.modules.multi_headed_attn import MultiHeadedAttention
    from mssrnet_style_transfer_amd.modules.position_ffn import PositionwiseFeedForward
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    cfg = MSSRConfig.small()
    nets = build_all(cfg, device="cpu", seed=1)
    enc, dec = nets["model"].encoder, nets["model"].decoder
    assert isinstance(enc, TransformerEncoder) and isinstance(dec, TransformerDecoder)
    assert isinstance(nets["style_generator"].encoder, TransformerEncoder)
    mha = enc.transformer[0].self_attn
    assert isinstance(mha, MultiHeadedAttention) and isinstance(enc.transformer[0], TransformerEncoderLayer)
    for name in ("linear_keys", "linear_values", "linear_query", "final_linear"):
        assert isinstance(getattr(mha, name), nn.Linear) and getattr(mha, name).bias is not None
    for name in ("softmax", "dropout", "dim_per_head", "model_dim", "head_count", "max_relative_positions"):
        assert hasattr(mha, name), name
    ffn = enc.transformer[0].feed_forward
    assert isinstance(ffn, PositionwiseFeedForward)
    for name in ("w_1", "w_2", "layer_norm", "dropout_1", "relu", "dropout_2"):
        assert hasattr(ffn, name), name
    assert ffn.layer_norm.eps == 1e-6 and enc.layer_norm.eps == 1e-6 and dec.layer_norm.eps == 1e-6
    for name in ("embeddings", "transformer", "layer_norm", "forward", "update_dropout", "from_opt", "_check_args"):
        assert hasattr(enc, name), name
    layer = dec.transformer_layers[0]
    assert isinstance(layer, TransformerDecoderLayer) and isinstance(layer.self_attn, MultiHeadedAttention)
    for name in ("self_attn", "context_attn", "feed_forward", "layer_norm_1", "layer_norm_2", "drop",
                 "full_context_alignment", "alignment_heads"):
        assert hasattr(layer, name), name
    for name in ("embeddings", "state", "transformer_layers", "_copy", "layer_norm", "alignment_layer", "from_opt",
                 "init_state", "map_state", "detach_state", "_init_cache", "update_dropout"):
        assert hasattr(dec, name), name
    emb = enc.embeddings
    assert isinstance(emb, Embeddings) and isinstance(emb.make_embedding.pe, PositionalEncoding)
    assert isinstance(emb.word_lut, nn.Embedding) and emb.emb_luts[0] is emb.word_lut
    for name in ("word_padding_idx", "word_vec_size", "embedding_size", "load_pretrained_vectors", "update_dropout"):
        assert hasattr(emb, name), name
    keys = set(enc.state_dict().keys())
    assert "embeddings.make_embedding.emb_luts.0.weight" in keys and "embeddings.make_embedding.pe.pe" in keys
    assert "transformer.0.self_attn.linear_keys.weight" in keys and "transformer.0.feed_forward.layer_norm.bias" in keys
    dkeys = set(dec.state_dict().keys())
    assert "transformer_layers.0.context_attn.final_linear.bias" in dkeys and "transformer_layers.0.layer_norm_2.weight" in dkeys
    assert tuple(enc.state_dict()["embeddings.make_embedding.pe.pe"].shape) == (5000, 1, cfg.d_model)


def test_transposed_weight_registry_follows_parameter_versions():
    """ops.register_transposed / ops.transposed: a registered copy is handed out only while the parameter is the same
    object, same storage, same version and same params_changed epoch, and only in the compute dtype it was made in."""
    from mssrnet_style_transfer_amd import ops
    ops.set_compute_dtype(torch.float32)
    lin = torch.nn.Linear(6, 4)
    w = lin.weight
    assert ops.transposed(w) is None
    ops.register_transposed(w, w.detach().t().contiguous())
    assert torch.equal(ops.transposed(w), w.detach().t())
    with torch.no_grad():
        w.mul_(2.0)                                        # in-place update bumps _version
    assert ops.transposed(w) is None
    ops.register_transposed(w, w.detach().t().contiguous())
    assert ops.transposed(w) is not None
    ops.params_changed(lin)                                # fused optimizers: the trainer's explicit signal
    assert ops.transposed(w) is None
    ops.register_transposed(w, w.detach().t().contiguous())
    w.data = w.data.clone()                                # new storage (e.g. module.to / load into new tensors)
    assert ops.transposed(w) is None
    ops.register_transposed(w, w.detach().t().contiguous())
    ops.set_compute_dtype(torch.bfloat16)                  # copies of another compute dtype are dropped
    try:
        assert ops.transposed(w) is None
    finally:
        ops.set_compute_dtype(torch.float32)


def test_phase_log_raises_the_deferred_label_check():
    from mssrnet_style_transfer_amd.trainer_wgan_wh import PhaseLog
    ok = PhaseLog(("x",), (torch.tensor(2.0),), check=(torch.tensor(False), "bad label"))
    assert ok["x"] == 2.0
    bad = PhaseLog(("x",), (torch.tensor(2.0),), check=(torch.tensor(True), "bad label"))
    with pytest.raises(KeyError, match="bad label"):
        bad["x"]
