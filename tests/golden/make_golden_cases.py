"""Argument sets shared by make_golden.py (``formats`` group) and tests/test_formats.py."""
FORMAT_CASES = {
    # name: postprocess_image arguments (reference argparse names)
    "sbs_ipd": dict(ipd_offset=3.0), "sbs_ipd_neg": dict(ipd_offset=-5.0),
    "pad_tblr": dict(pad=0.1, pad_mode="tblr"), "pad_tb": dict(pad=0.07, pad_mode="tb"), "pad_lr": dict(pad=0.13, pad_mode="lr"),
    "pad_top": dict(pad=0.2, pad_mode="top", tb=True), "pad_169": dict(pad_mode="16:9", cross_eyed=True),
    "vr180": dict(vr180=True), "half_tb": dict(half_tb=True),
    "max_out": dict(half_sbs=True, max_output_width=100, max_output_height=40, keep_aspect_ratio=True),
    "ana_color": dict(anaglyph="color"), "ana_gray": dict(anaglyph="gray"), "ana_half": dict(anaglyph="half-color"),
    "ana_wimmer": dict(anaglyph="wimmer"), "ana_wimmer2": dict(anaglyph="wimmer2"), "ana_dubois": dict(anaglyph="dubois"),
    "ana_dubois2": dict(anaglyph="dubois2", ipd_offset=2.0),
}
