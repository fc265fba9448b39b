"""Import recipe for the reference's Python modules.

Runs ONLY in the build container, from ``make_golden.py``: the reference tree
(/root/reference) never travels to the GPU box and nothing under tests/ that pytest
collects imports this file.  The recipe skips ``onmt/__init__.py`` (it imports a file
that is missing from the checkout and torchtext) and stubs the four modules that need
absent third-party packages.  No reference source is copied: modules are imported in
place.
"""
import sys
import types
import warnings

REF = "/root/reference"


def import_reference(before_imports=None):
    """``before_imports``: called after the ``onmt`` namespace exists and before any reference module is
    imported (make_alias_fixture.py registers the drop-in L1 modules there)."""
    sys.dont_write_bytecode = True
    warnings.filterwarnings("ignore", category=FutureWarning)
    if "onmt" in sys.modules and getattr(sys.modules["onmt"], "_is_ref", False):
        return sys.modules["onmt"]
    for k in [k for k in sys.modules if k == "onmt" or k.startswith("onmt.")]:
        del sys.modules[k]
    onmt = types.ModuleType("onmt")
    onmt.__path__ = [REF + "/onmt"]
    onmt._is_ref = True
    sys.modules["onmt"] = onmt
    if before_imports is not None:
        before_imports()

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _TB:
        def __init__(self, *a, **k):
            pass

        def put_scalar(self, *a, **k):
            pass

    stub("onmt.auto_occupy", auto_occupy_gpu_error_friendly2=lambda *a, **k: None)
    import onmt.utils  # noqa: F401
    stub("onmt.utils.tensor_board_wrapper", TensorBoardWrapper=_TB)
    tr = stub("onmt.translate")
    tr.__path__ = [REF + "/onmt/translate"]
    stub("onmt.translate.during_training_translator",
         build_translator_for_train=lambda *a, **k: None)
    import onmt.modules.multi_headed_attn  # noqa: F401
    import onmt.modules.position_ffn  # noqa: F401
    import onmt.modules.embeddings  # noqa: F401
    import onmt.encoders.transformer  # noqa: F401
    import onmt.decoders.transformer  # noqa: F401
    import onmt.models.model  # noqa: F401
    import onmt.style_generator  # noqa: F401
    import onmt.style_generator_builder  # noqa: F401
    import onmt.train_and_translate  # noqa: F401
    import onmt.trainer_wgan_wh  # noqa: F401
    import onmt.trainer_teacher  # noqa: F401
    import onmt.translate.greedy_search  # noqa: F401
    return onmt
