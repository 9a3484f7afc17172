"""``nunif_amd.install()`` — put the HIP engine underneath a LIVE checkout of nagadomi/nunif, so that ``waifu2x.cli`` /
``iw3.cli`` (and anything else that imports the reference) run on it unchanged; ``uninstall()`` restores every binding.

The reference's hot-path boundary is a set of Python names.  Its own modules bind those names with
``from x import y`` (``waifu2x/utils.py:6`` ``from nunif.utils.render import tiled_render``, ``iw3/utils.py:30-42``), so
replacing ``nunif.utils.render.tiled_render`` alone would leave ``waifu2x.utils.tiled_render`` pointing at the torch path.
``install()`` therefore

1. imports the reference modules that define the boundary (they must be importable: put the checkout on ``sys.path``),
2. for every entry of :data:`PATCHES` rebinds the name in its defining module AND in every loaded ``nunif`` / ``waifu2x`` /
   ``iw3`` / ``stlizer`` / ``cliqa`` module whose attribute *is* the original object,
3. overwrites the reference's model registry (``nunif/models/register.py:9`` ``_models``) for every model name both sides
   register — ``load_model`` / ``create_model`` of the reference then build HIP-engine models from the same ``.pth`` files
   (``nunif/models/utils.py:41-77`` is untouched: it only calls ``create_model``, ``load_state_dict``, ``.to``).

Nothing here computes anything; no reference source is modified.  Signatures are identical by construction and checked by
``tests/test_install.py`` against the live reference (``inspect.signature`` equality for B2-B6).
"""
import importlib
import sys

# (reference module, attribute)  ->  same module path under nunif_amd, same attribute
PATCHES = [
    # B2: tiled render + stitcher (nunif/utils/render.py:8, nunif/utils/seam_blending.py:48-174)
    ("nunif.utils.render", "tiled_render"),
    ("nunif.utils.seam_blending", "SeamBlending"),
    # B3 / B4: waifu2x contexts (waifu2x/utils.py:42-297, waifu2x/hub.py:31-175)
    ("waifu2x.utils", "Waifu2x"),
    ("waifu2x.hub", "Waifu2xImageModel"),
    # alpha / TTA helpers used by Waifu2x.convert (nunif/utils/alpha.py, nunif/transforms/tta.py)
    ("nunif.utils.alpha", "AlphaBorderPadding"),
    ("nunif.transforms.tta", "tta_split"),
    ("nunif.transforms.tta", "tta_merge"),
    # B5: depth model contract and factory (iw3/base_depth_model.py:30, iw3/depth_model_factory.py)
    ("iw3.base_depth_model", "BaseDepthModel"),
    ("iw3.depth_model_factory", "create_depth_model"),
    ("iw3.depth_scaler", "EMAMinMaxScaler"),
    ("iw3.depth_anything_model", "batch_preprocess"),
    ("iw3.mapper", "get_mapper"),
    # B6: stereo synthesis (iw3/forward_warp.py:246, iw3/backward_warp.py:96,124,343)
    ("iw3.forward_warp", "apply_divergence_forward_warp"),
    ("iw3.forward_warp", "nonwarp_mask"),
    ("iw3.backward_warp", "apply_divergence_grid_sample"),
    ("iw3.backward_warp", "apply_divergence_nn_LR"),
    ("iw3.backward_warp", "apply_divergence_nn_symmetric"),
    ("iw3.backward_warp", "backward_warp"),
    ("iw3.backward_warp", "postprocess_hole_mask"),
    ("iw3.stereo_model_factory", "create_stereo_model"),
    # depth post-processing (iw3/dilation.py:41-153)
    ("iw3.dilation", "dilate_edge"),
    ("iw3.dilation", "dilate"),
    ("iw3.dilation", "erode"),
    ("iw3.dilation", "closing"),
    ("iw3.dilation", "mask_closing"),
    ("iw3.dilation", "dilate_outer"),
    ("iw3.dilation", "dilate_inner"),
    # per-frame glue of iw3.utils (iw3/utils.py:247-487)
    ("iw3.utils", "apply_divergence"),
    ("iw3.utils", "postprocess_image"),
    ("iw3.utils", "preprocess_image"),
    ("iw3.equirectangular", "equirectangular_projection"),
]

# reference packages whose modules may hold ``from ... import`` copies of a patched name
_CONSUMER_ROOTS = ("nunif", "waifu2x", "iw3", "stlizer", "cliqa")

# modules that register models (imported on both sides so that the registries are complete before they are merged)
_MODEL_MODULES = [("waifu2x.models", ["nunif_amd.waifu2x.models.swin_unet", "nunif_amd.waifu2x.models.cunet",
                                      "nunif_amd.waifu2x.models.vgg_7", "nunif_amd.waifu2x.models.upconv_7",
                                      "nunif_amd.waifu2x.models.swin_unet_v2"]),
                  ("iw3.models", ["nunif_amd.iw3.models"])]

_state = None      # {"bindings": [(module, attr, original)], "registry": {name: original factory or _MISSING}}
_MISSING = object()


def is_installed():
    return _state is not None


def _consumers():
    for name, mod in list(sys.modules.items()):
        if mod is not None and name.split(".")[0] in _CONSUMER_ROOTS:
            yield mod


def install(registry=True, strict=True):
    """Rebind the reference's hot-path names to the HIP engine.  Returns a report
    ``{"patched": {"module.attr": n_bindings}, "models": [names], "skipped": [(module.attr, reason)]}``.

    ``strict=False`` skips a reference module that fails to import (e.g. ``iw3.utils`` without PyAV) instead of raising."""
    global _state
    if _state is not None:
        raise RuntimeError("nunif_amd is already installed; call uninstall() first")
    report = {"patched": {}, "models": [], "skipped": []}
    bindings = []
    reg_saved = {}
    try:
        jobs = []
        for ref_name, attr in PATCHES:
            try:
                ref_mod = importlib.import_module(ref_name)
                ours = getattr(importlib.import_module("nunif_amd." + ref_name), attr)
                orig = getattr(ref_mod, attr)
            except Exception as e:
                if strict:
                    raise ImportError(f"nunif_amd.install: cannot resolve {ref_name}.{attr}: {e!r} "
                                      "(is the nunif checkout on sys.path?)") from e
                report["skipped"].append((f"{ref_name}.{attr}", repr(e)))
                continue
            jobs.append((ref_name, attr, orig, ours))
        for ref_name, attr, orig, ours in jobs:
            n = 0
            for mod in _consumers():
                for key, val in list(vars(mod).items()):
                    if val is orig:
                        bindings.append((mod, key, orig))
                        setattr(mod, key, ours)
                        n += 1
            report["patched"][f"{ref_name}.{attr}"] = n
        if registry:
            for ref_models, our_modules in _MODEL_MODULES:
                try:
                    importlib.import_module(ref_models)
                except Exception as e:
                    if strict:
                        raise
                    report["skipped"].append((ref_models, repr(e)))
                for m in our_modules:
                    importlib.import_module(m)
            ref_reg = importlib.import_module("nunif.models.register")._models
            our_reg = importlib.import_module("nunif_amd.nunif.models.register")._models
            new_entries = {}
            for name, factory in our_reg.items():
                if name not in ref_reg:                  # only names the reference itself knows: a drop-in, not an extension
                    continue
                if getattr(factory, "_nunif_amd_unsupported", False):
                    # a name the engine only knows in order to refuse it: the reference's own torch factory stays reachable
                    report["skipped"].append((name, "engine factory is an 'unsupported' stub; reference factory kept"))
                    continue
                new_entries[name] = factory
            # built first, applied in one step: nothing between here and `_state` can fail half-way through the registry
            reg_saved = {name: ref_reg[name] for name in new_entries}
            ref_reg.update(new_entries)
            report["models"].extend(new_entries)
        _state = {"bindings": bindings, "registry": reg_saved}
    except Exception:
        for mod, key, orig in reversed(bindings):
            setattr(mod, key, orig)
        if reg_saved:
            importlib.import_module("nunif.models.register")._models.update(reg_saved)
        raise
    return report


def uninstall():
    """Undo :func:`install`: every rebound name and registry entry gets its original object back."""
    global _state
    if _state is None:
        return
    for mod, key, orig in reversed(_state["bindings"]):
        setattr(mod, key, orig)
    if _state["registry"]:
        ref_reg = importlib.import_module("nunif.models.register")._models
        for name, factory in _state["registry"].items():
            ref_reg[name] = factory
    _state = None
