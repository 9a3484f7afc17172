"""Import shim that lets the *reference* (``/root/reference``) be imported in the build container.

TEST INFRASTRUCTURE ONLY.  Used by ``tests/golden/make_golden.py`` (fixture generation) and by the
``-m "not gpu"`` tests that cross-check the oracle against the live reference when it is mounted.
Nothing on the product path imports this file, and ``/root/reference`` does not exist on the GPU box.

The reference fails to import here because ``torchvision``, ``av``, ``cv2`` and ``timm`` are absent
(``nunif/transforms/std.py:2``).  None of their attributes is touched on the hot path, so a meta-path
finder hands out inert placeholder modules.  The one external piece that *is* executed —
``torchvision.models.swin_transformer.SwinTransformerBlock`` (``waifu2x/models/swin_unet.py:9-12``) — is
provided by :mod:`oracle.tv_swin_block`, a torch restatement of torchvision 0.22's V1 block, so the
reference's own ``SwinUNetBase`` runs unmodified on top of it.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types

REFERENCE_ROOT = os.environ.get("NUNIF_REFERENCE_ROOT", "/root/reference")
_STUB_ROOTS = ("torchvision", "av", "cv2", "timm")


class _Inert:
    """Callable, attribute-chaining, subclassable placeholder."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # used as a bare decorator
        return _Inert()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Inert()

    def __mro_entries__(self, bases):
        return (object,)

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Inert()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        m.__version__ = "0.0.0+stub"
        return m

    def exec_module(self, module):
        pass


_installed = False


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "nunif"))


def install(swin_block="oracle"):
    """Make ``import nunif / waifu2x / iw3`` resolve to the reference. Idempotent.

    ``swin_block``: which class stands in for ``torchvision.models.swin_transformer.SwinTransformerBlock`` —
    ``"oracle"`` = :mod:`oracle.tv_swin_block` (the restatement), ``"hf"`` = :class:`oracle.hf_pin.HFSwinTransformerBlock`
    (HuggingFace's ``SwinLayer`` behind torchvision's constructor and key layout; the independent pin).  Must be chosen
    before the reference's ``waifu2x.models.swin_unet`` is first imported."""
    global _installed
    if _installed:
        if swin_block != _installed:
            raise RuntimeError(f"refstub already installed with swin_block={_installed!r}")
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    os.environ.setdefault("NUNIF_HOME", tempfile.mkdtemp(prefix="nunif_home_"))
    sys.meta_path.insert(0, _StubFinder())
    # real module for the one executed external symbol
    from . import tv_swin_block
    import importlib
    importlib.import_module("torchvision")
    importlib.import_module("torchvision.models")
    mod = types.ModuleType("torchvision.models.swin_transformer")
    if swin_block == "hf":
        from . import hf_pin
        mod.SwinTransformerBlock = hf_pin.HFSwinTransformerBlock
    else:
        mod.SwinTransformerBlock = tv_swin_block.SwinTransformerBlock
    sys.modules["torchvision.models.swin_transformer"] = mod
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = swin_block
