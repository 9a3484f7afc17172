"""Red-cyan anaglyph composition on the HIP engine.  Mirrors ``iw3/anaglyph.py``: ``apply_anaglyph_redcyan`` :96-110 and
the seven methods it dispatches to (``color`` :9, ``half_color`` :14, ``gray`` :21, ``wimmer`` :29, ``wimmer2`` :38,
``dubois`` / ``dubois2`` :51-93) — one pointwise kernel, ``nunif_hip_anaglyph``."""
from . import _ops

ANAGLYPH_MODES = {"color": 0, "gray": 1, "half-color": 2, "wimmer": 3, "wimmer2": 4, "dubois": 5, "dubois2": 6}


def apply_anaglyph_redcyan(left_eye, right_eye, anaglyph_type):
    """CHW, CHW float in [0,1] -> CHW."""
    if anaglyph_type not in ANAGLYPH_MODES:
        raise ValueError(f"Unknown anaglyph_type {anaglyph_type}")
    return _ops.anaglyph(left_eye, right_eye, ANAGLYPH_MODES[anaglyph_type])
