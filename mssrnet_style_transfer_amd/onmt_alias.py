"""``import onmt.modules.multi_headed_attn`` -> this package ("at the same import paths").

The reference has no plugin registry: its callers import the five L1 files by name
(``onmt/style_generator.py:9`` ``from onmt.encoders.transformer import TransformerEncoder``,
``onmt/modules/__init__.py:8``, ``onmt/decoders/transformer.py:9-10`` ...), and its checkpoints
are pickled module objects whose class paths are ``onmt.modules...`` / ``onmt.encoders...``
(``onmt/trainer_wgan_wh.py:1093-1100``, loaded by ``onmt/translate/translator_wh.py:29-35``
and ``onmt/cls_proxy.py:14-22``).  ``install()`` makes those names resolve to the drop-in
classes by registering entries in ``sys.modules``:

* **inside the reference tree** (a real ``onmt`` package is importable, or already imported):
  only the five L1 module paths are taken over, *before* ``import onmt`` runs, so that
  ``onmt/modules/__init__.py`` and every caller pick up the gfx950 classes and everything else
  of the reference (trainer, builders, translator) stays its own code;
* **stand-alone** (no reference on ``sys.path``, e.g. a serving box that only has checkpoints):
  stub parent packages are created as well, plus the caller classes that end up in checkpoints
  (``onmt.models.model.NMTModel``, ``onmt.style_generator.StyleGenerator2``,
  ``onmt.style_generator_builder.{StyleClassifier, ContextReps2, StyleDiscriminator, BondLayer}``,
  ``onmt.modules.util_class.{Elementwise, Cast}``, ``onmt.utils.misc.sequence_mask``), so a
  reference-pickled module unpickles into drop-in objects.

``load_checkpoint(path)`` is the loader to use for such files: torch's restricted unpickler
(``weights_only=True``) with an allow-list of exactly the drop-in classes under their ``onmt.*`` paths and
the torch.nn containers they are made of.  Nothing from the file is executed (no arbitrary ``find_class``,
no ``__reduce__`` callables outside the list), and it needs no ``install()``.

Nothing here touches the GPU; the classes fail loudly at ``forward`` time if the HIP library is absent.
"""
import importlib
import importlib.util
import sys
import types

_PKG = __name__.rsplit(".", 1)[0]

# the five files of SURVEY section 8b (+ the two tiny base-class files they import)
L1_PATHS = {
    "onmt.modules.multi_headed_attn": ".modules.multi_headed_attn",
    "onmt.modules.position_ffn": ".modules.position_ffn",
    "onmt.modules.embeddings": ".modules.embeddings",
    "onmt.encoders.transformer": ".encoders.transformer",
    "onmt.decoders.transformer": ".decoders.transformer",
}
# stand-alone only: containers / callers whose class paths occur in reference checkpoints
STANDALONE_PATHS = {
    "onmt.modules.util_class": ".modules.util_class",
    "onmt.encoders.encoder": ".encoders.encoder",
    "onmt.decoders.decoder": ".decoders.decoder",
    "onmt.utils.misc": ".utils.misc",
}
# name -> (module of this package, attribute names) for synthetic caller modules
STANDALONE_CALLERS = {
    "onmt.models.model": (".style_generator_builder", ("NMTModel",)),
    "onmt.style_generator": (".style_generator_builder", ("StyleGenerator2",)),
    "onmt.style_generator_builder": (".style_generator_builder",
                                     ("StyleClassifier", "ContextReps2", "StyleDiscriminator", "BondLayer")),
}
# what the parent packages re-export (onmt/modules/__init__.py:2-14, onmt/encoders/__init__.py,
# onmt/decoders/__init__.py, onmt/models/__init__.py), restricted to the path
_REEXPORTS = {
    "onmt.modules": {"MultiHeadedAttention": "onmt.modules.multi_headed_attn",
                     "Embeddings": "onmt.modules.embeddings", "PositionalEncoding": "onmt.modules.embeddings",
                     "VecEmbedding": "onmt.modules.embeddings",
                     "Elementwise": "onmt.modules.util_class", "Cast": "onmt.modules.util_class"},
    "onmt.encoders": {"TransformerEncoder": "onmt.encoders.transformer", "EncoderBase": "onmt.encoders.encoder"},
    "onmt.decoders": {"TransformerDecoder": "onmt.decoders.transformer", "DecoderBase": "onmt.decoders.decoder"},
    "onmt.models": {"NMTModel": "onmt.models.model"},
    "onmt.utils": {},
}

_installed = {}


def _mine(rel):
    return importlib.import_module(rel, _PKG)


def reference_present():
    """True when a real ``onmt`` package (a directory with the reference's files) is importable or imported."""
    mod = sys.modules.get("onmt")
    if mod is not None:
        return not getattr(mod, "_mssr_stub", False)
    try:
        return importlib.util.find_spec("onmt") is not None
    except (ImportError, ValueError):
        return False


def _stub_package(name):
    mod = sys.modules.get(name)
    if mod is None:
        mod = types.ModuleType(name)
        mod.__path__ = []                    # a package: submodule imports consult sys.modules first
        mod._mssr_stub = True
        sys.modules[name] = mod
        _installed[name] = mod
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(_stub_package(parent), leaf, mod)
    return mod


def _register(name, module):
    sys.modules[name] = module
    _installed[name] = module
    parent, _, leaf = name.rpartition(".")
    pmod = sys.modules.get(parent)
    if pmod is not None:
        setattr(pmod, leaf, module)


def install(standalone=None):
    """Registers the aliases; returns the list of ``onmt.*`` names now served by this package.
    ``standalone=None`` decides by ``reference_present()``.  Idempotent.  Inside the reference tree call it
    BEFORE ``import onmt`` (so the reference's own imports find the drop-in modules) and, if attribute-style
    access such as ``onmt.modules.position_ffn`` is wanted, once more afterwards: Python binds a submodule to
    its parent package only when the import system loads it, not when it is found in ``sys.modules``."""
    if standalone is None:
        standalone = not reference_present()
    if standalone:
        for pkg in _REEXPORTS:
            _stub_package(pkg)
    for name, rel in L1_PATHS.items():
        _register(name, _mine(rel))
    if standalone:
        for name, rel in STANDALONE_PATHS.items():
            _register(name, _mine(rel))
        for name, (rel, attrs) in STANDALONE_CALLERS.items():
            src = _mine(rel)
            mod = types.ModuleType(name)
            mod._mssr_stub = True
            for a in attrs:
                setattr(mod, a, getattr(src, a))
            _register(name, mod)
        for pkg, table in _REEXPORTS.items():
            pmod = sys.modules[pkg]
            if getattr(pmod, "_mssr_stub", False):
                for attr, owner in table.items():
                    setattr(pmod, attr, getattr(sys.modules[owner], attr))
    return sorted(_installed)


def uninstall():
    """Removes every entry ``install()`` added (tests)."""
    for name, mod in list(_installed.items()):
        if sys.modules.get(name) is mod:
            del sys.modules[name]
        parent, _, leaf = name.rpartition(".")
        pmod = sys.modules.get(parent)
        if pmod is not None and getattr(pmod, leaf, None) is mod:
            try:
                delattr(pmod, leaf)
            except AttributeError:
                pass
    _installed.clear()


def reference_class_path(cls):
    """``onmt....`` module path under which the reference pickles ``cls`` (None if it has no counterpart)."""
    here = cls.__module__
    for name, rel in list(L1_PATHS.items()) + list(STANDALONE_PATHS.items()):
        if here == _PKG + rel:
            return name
    for name, (rel, attrs) in STANDALONE_CALLERS.items():
        if here == _PKG + rel and cls.__name__ in attrs:
            return name
    return None


def checkpoint_allow_list():
    """``(class, "onmt.path.Name")`` pairs + the torch.nn classes a reference checkpoint of this path is made of:
    the argument of ``torch.serialization.safe_globals`` for ``load_checkpoint``."""
    import torch.nn as nn
    allow = []
    for name, rel in list(L1_PATHS.items()) + list(STANDALONE_PATHS.items()):
        mod = _mine(rel)
        for attr, obj in vars(mod).items():
            if isinstance(obj, type) and issubclass(obj, nn.Module) and obj.__module__ == mod.__name__:
                allow.append((obj, name + "." + attr))
    for name, (rel, attrs) in STANDALONE_CALLERS.items():
        for attr in attrs:
            allow.append((getattr(_mine(rel), attr), name + "." + attr))
    allow += [nn.Linear, nn.LayerNorm, nn.Dropout, nn.ModuleList, nn.Sequential, nn.Softmax, nn.LogSoftmax, nn.ReLU,
              nn.Sigmoid, nn.Tanh, nn.Embedding, nn.Parameter, set]
    return allow


def load_checkpoint(f, map_location=None):
    """Loads a whole-module checkpoint as the reference writes them (``torch.save({'main_model': module, ...})``,
    onmt/trainer_wgan_wh.py:1093-1100; ``torch.save(self.model)``, onmt/trainer_teacher.py:282) into the drop-in
    classes WITHOUT a code-executing unpickle: ``weights_only=True`` under ``checkpoint_allow_list()``.  A file that
    names any other global is refused by torch with the offending path in the message."""
    import torch
    with torch.serialization.safe_globals(checkpoint_allow_list()):
        return torch.load(f, map_location=map_location, weights_only=True)


class reference_class_paths:
    """Context manager: while active, the drop-in classes report the reference's module paths, so
    ``torch.save(module, f)`` inside it writes a checkpoint the REFERENCE can unpickle (and that
    ``install()`` + ``torch.load`` maps back onto this package) -- the two directions of the
    whole-module checkpoints of onmt/trainer_wgan_wh.py:1093-1100.  Needs ``install(standalone=True)``
    style entries for every class it renames (pickle verifies ``sys.modules[path].Name is cls``)."""

    def __enter__(self):
        import torch.nn as nn
        self._saved = []
        seen = set()
        rels = set(L1_PATHS.values()) | set(STANDALONE_PATHS.values()) | {r for r, _ in STANDALONE_CALLERS.values()}
        for rel in rels:
            mod = _mine(rel)
            for obj in vars(mod).values():
                if isinstance(obj, type) and issubclass(obj, nn.Module) and obj.__module__ == mod.__name__ \
                        and obj not in seen:
                    path = reference_class_path(obj)
                    if path is not None and getattr(sys.modules.get(path), obj.__name__, None) is obj:
                        seen.add(obj)
                        self._saved.append((obj, obj.__module__))
                        obj.__module__ = path
        return self

    def __exit__(self, *exc):
        for cls, module in self._saved:
            cls.__module__ = module
