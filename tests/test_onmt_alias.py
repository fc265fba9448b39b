"""SURVEY section 8b, "at the same import paths": ``onmt_alias.install()`` serves the reference's module paths from
this package, reference-pickled modules (class paths ``onmt.*``) unpickle into the drop-in classes, and the
reference's own callers accept them (fixtures written in the build container by tests/golden/make_golden.py
``pickles`` and tests/golden/make_alias_fixture.py; the reference itself is not needed here)."""
import importlib
import io
import json
import os
import sys

import pytest
import torch

from conftest import GOLDEN
from mssrnet_style_transfer_amd import onmt_alias
from mssrnet_style_transfer_amd.config import MSSRConfig


@pytest.fixture()
def alias():
    assert not any(k == "onmt" or k.startswith("onmt.") for k in sys.modules), "a real onmt is loaded in this process"
    names = onmt_alias.install(standalone=True)
    yield names
    onmt_alias.uninstall()
    assert not any(k == "onmt" or k.startswith("onmt.") for k in sys.modules)


def test_module_paths_resolve_here(alias):
    import mssrnet_style_transfer_amd as pkg
    for ref, rel in {**onmt_alias.L1_PATHS, **onmt_alias.STANDALONE_PATHS}.items():
        assert importlib.import_module(ref) is importlib.import_module(rel, pkg.__name__), ref
    import onmt.modules
    import onmt.modules.multi_headed_attn as mha
    from onmt.decoders.transformer import TransformerDecoder, TransformerDecoderLayer  # noqa: F401
    from onmt.encoders.transformer import TransformerEncoder, TransformerEncoderLayer  # noqa: F401
    from onmt.modules import Embeddings, MultiHeadedAttention, PositionalEncoding, VecEmbedding  # noqa: F401
    from onmt.modules.position_ffn import PositionwiseFeedForward  # noqa: F401
    from onmt.utils.misc import sequence_mask
    assert MultiHeadedAttention is mha.MultiHeadedAttention is onmt.modules.multi_headed_attn.MultiHeadedAttention
    assert MultiHeadedAttention.__module__ == "mssrnet_style_transfer_amd.modules.multi_headed_attn"
    assert sequence_mask(torch.tensor([1, 3]), 3).tolist() == [[True, False, False], [True, True, True]]
    from onmt.style_generator import StyleGenerator2
    from onmt.style_generator_builder import BondLayer, ContextReps2, StyleClassifier, StyleDiscriminator  # noqa: F401
    from onmt.models.model import NMTModel  # noqa: F401
    # the isinstance check of onmt/style_generator.py:22 holds for encoders built through either path
    from mssrnet_style_transfer_amd.style_generator_builder import build_style_generator
    sg = build_style_generator(MSSRConfig.small())
    assert isinstance(sg, StyleGenerator2) and isinstance(sg.encoder, TransformerEncoder)


def test_install_is_idempotent_and_uninstall_restores(alias):
    again = onmt_alias.install(standalone=True)
    assert again == alias and "onmt.encoders.transformer" in alias and "onmt" in alias


def test_checkpoint_with_reference_class_paths_round_trips(alias):
    """torch.save inside reference_class_paths() writes the class paths the reference expects
    (onmt/trainer_wgan_wh.py:1093-1100); loading maps them back onto the drop-in classes."""
    from mssrnet_style_transfer_amd.style_generator_builder import build_all
    cfg = MSSRConfig.small()
    nets = build_all(cfg, device="cpu", seed=2)
    buf = io.BytesIO()
    with onmt_alias.reference_class_paths():
        torch.save({"main_model": nets["model"], "style_generator": nets["style_generator"], "step": 3}, buf)
    blob = buf.getvalue()
    for path in (b"onmt.encoders.transformer", b"onmt.decoders.transformer", b"onmt.modules.multi_headed_attn",
                 b"onmt.modules.position_ffn", b"onmt.modules.embeddings", b"onmt.models.model",
                 b"onmt.style_generator"):
        assert path in blob, path
    assert b"mssrnet_style_transfer_amd" not in blob
    assert type(nets["model"]).__module__ == "mssrnet_style_transfer_amd.style_generator_builder"   # restored
    with pytest.raises(Exception, match="(?i)unsupported global|weights_only"):
        torch.load(io.BytesIO(blob), weights_only=True)          # not without the allow-list
    onmt_alias.uninstall()                                        # the allow-listed loader needs no aliases
    back = onmt_alias.load_checkpoint(io.BytesIO(blob))
    onmt_alias.install(standalone=True)
    assert back["step"] == 3
    for key, name in (("main_model", "model"), ("style_generator", "style_generator")):
        assert type(back[key]) is type(nets[name])
        for (k, a), (k2, b) in zip(nets[name].state_dict().items(), back[key].state_dict().items()):
            assert k == k2 and torch.equal(a, b)


def test_reference_pickled_modules_load_into_the_drop_in_classes(alias):
    """tests/golden/ref_pickles.pt was written by torch.save from the REFERENCE's own module objects (what its
    trainers checkpoint).  translator_wh.py:29-35 / cls_proxy.py:14-22 load such files with torch.load."""
    from mssrnet_style_transfer_amd import style_generator_builder as B
    from mssrnet_style_transfer_amd.decoders.transformer import TransformerDecoder
    from mssrnet_style_transfer_amd.encoders.transformer import TransformerEncoder
    from mssrnet_style_transfer_amd.modules.multi_headed_attn import MultiHeadedAttention
    ck = onmt_alias.load_checkpoint(os.path.join(GOLDEN, "ref_pickles.pt"))     # restricted unpickler, allow-list only
    cfg = MSSRConfig.small()
    mine = B.build_all(cfg, device="cpu", seed=0)
    pairs = {"main_model": "model", "style_generator": "style_generator", "style_discriminator": "style_discriminator",
             "text_discriminator": "text_discriminator", "teacher": "teacher"}
    for key, name in pairs.items():
        got = ck[key]
        assert type(got) is type(mine[name]), (key, type(got))
        assert list(got.state_dict().keys()) == list(mine[name].state_dict().keys()), key
        res = mine[name].load_state_dict(got.state_dict())
        assert not res.missing_keys and not res.unexpected_keys
    m = ck["main_model"]
    assert isinstance(m.encoder, TransformerEncoder) and isinstance(m.decoder, TransformerDecoder)
    assert isinstance(m.decoder.transformer_layers[0].context_attn, MultiHeadedAttention)
    assert isinstance(ck["style_generator"].encoder, TransformerEncoder)


def test_allow_listed_loader_refuses_foreign_globals():
    """A checkpoint naming anything outside the allow-list (here: os.system through __reduce__) is refused
    before anything runs."""
    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    buf = io.BytesIO()
    torch.save({"main_model": Evil()}, buf)
    with pytest.raises(Exception, match="(?i)unsupported global|weights_only"):
        onmt_alias.load_checkpoint(io.BytesIO(buf.getvalue()))


def test_reference_callers_built_around_the_drop_in_modules():
    """tests/golden/alias_keys.json: recorded in the build container with the reference's own StyleGenerator2 /
    StyleClassifier / StyleDiscriminator / NMTModel / TrainTranslator constructed around this package's encoder and
    decoder (aliases installed before ``import onmt``)."""
    from mssrnet_style_transfer_amd import style_generator_builder as B
    rec = json.load(open(os.path.join(GOLDEN, "alias_keys.json")))
    for name, where in rec["resolved"].items():
        assert where.startswith("mssrnet_style_transfer_amd") == name.startswith("onmt."), (name, where)
    mine = B.build_all(MSSRConfig.small(), device="cpu", seed=0)
    for name, keys in rec["keys"].items():
        assert list(mine[name].state_dict().keys()) == keys, name
    for k, shape in rec["shapes"]["model"].items():
        assert list(mine["model"].state_dict()[k].shape) == shape, k


def test_parameter_counts_at_the_readme_configuration():
    """SURVEY section 8a' (builders): 77.9 M parameters in the main model, 4.86 M in the teacher at V = 9606; the
    exact counts come from the reference's constructors (tests/golden/readme.pt, make_golden.py gen_readme)."""
    from mssrnet_style_transfer_amd import style_generator_builder as B
    want = torch.load(os.path.join(GOLDEN, "readme.pt"), weights_only=True)["param_counts"]
    assert want["model"] == 77985670 and want["teacher"] == 4862725
    cfg = MSSRConfig()
    with torch.device("meta"):
        nets = dict(model=B.build_main_model(cfg), style_generator=B.build_style_generator(cfg),
                    style_discriminator=B.build_style_discriminator(cfg),
                    text_discriminator=B.build_text_discriminator(cfg), teacher=B.build_classifier(cfg))
    for name, n in want.items():
        assert sum(p.numel() for p in nets[name].parameters()) == n, name
