"""The C-ABI library: loads, exports every symbol include/nunif_hip.h declares, and its host-only integer
routines agree with the reference fixtures.  No device compute here."""
import ctypes
import json
import os
import re
import struct

import pytest

from conftest import GOLDEN, ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nunif_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nunif_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hiplib):
    from nunif_amd import _hip
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(hiplib, s), f"{s} declared in nunif_hip.h but not exported"
        assert s in _hip.SIGNATURES, f"{s} has no ctypes signature in nunif_amd/_hip.py"
    assert sorted(_hip.SIGNATURES) == syms
    assert hiplib.nunif_hip_abi_version() == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nunif_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _hip.lib()


def test_tile_grid_bit_exact_with_reference_configs(hiplib):
    from nunif_amd import _hip
    for case in json.load(open(os.path.join(GOLDEN, "seam_configs.json"))):
        h, w, s, o, t, b = case["args"]
        cfg = _hip.tile_grid(h, w, s, o, t, b).as_config()
        ref = dict(case["config"])
        ref["pad"] = tuple(ref["pad"])
        assert cfg == ref, case["args"]
        if b > 0:
            ramp = _hip.blend_ramp(b)
            bits = [struct.unpack("<i", struct.pack("<f", v))[0] for v in ramp]
            assert bits == case["ramp_bits"][:b], case["args"]


def test_tile_grid_rejects_bad_arguments(hiplib):
    from nunif_amd import _hip
    with pytest.raises(_hip.NunifHipError) as e:
        _hip.tile_grid(100, 100, 2, 16, 16, 8)     # step <= 0
    assert e.value.status == -1 and "too small" in str(e.value)
    with pytest.raises(_hip.NunifHipError):
        _hip.tile_grid(0, 100, 2, 16, 64, 8)


def test_mirror_create_config_and_filter(hiplib):
    import torch
    from nunif_amd.nunif.utils.seam_blending import SeamBlending
    from oracle import seam_blending as OS
    assert SeamBlending.create_config((1080, 1920), 2, 16, 256, 8) == OS.create_config(1080, 1920, 2, 16, 256, 8)
    assert torch.equal(SeamBlending.create_blend_filter(2, 16, 256, 8, 3), OS.blend_filter(2, 16, 256, 8, 3))
    assert torch.equal(SeamBlending.create_blend_filter(4, 32, 64, 16, 1), OS.blend_filter(4, 32, 64, 16, 1))
