"""ctypes binding of ``libnunif_hip.so`` (the C ABI declared in ``include/nunif_hip.h``).

There is no CPU fallback: if the library is missing or a call fails this module raises.  The library is built
in-tree by ``python -m nunif_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NUNIF_HIP_LIB: another build of the same ABI (same-box A/B runs of two kernel versions, tools/ab_lib.sh); never a fallback
LIB_PATH = os.environ.get("NUNIF_HIP_LIB") or os.path.join(_HERE, "libnunif_hip.so")

c_void_p, c_int32, c_int64, c_float, c_char_p, c_double = (
    ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_char_p, ctypes.c_double)


class NunifHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libnunif_hip error {status}: {message}")
        self.status = status


class TileGrid(ctypes.Structure):
    """``nunif_tile_grid`` — the integer outputs of SeamBlending.create_config (seam_blending.py:109-143)."""
    _fields_ = [(n, c_int32) for n in (
        "x_h", "x_w", "scale", "offset", "tile_size", "blend_size",
        "y_h", "y_w", "h_blocks", "w_blocks",
        "pad_l", "pad_r", "pad_t", "pad_b", "y_buffer_h", "y_buffer_w",
        "input_tile_step", "output_tile_step", "out_tile_size")]

    def as_config(self):
        return {"y_h": self.y_h, "y_w": self.y_w, "h_blocks": self.h_blocks, "w_blocks": self.w_blocks,
                "pad": (self.pad_l, self.pad_r, self.pad_t, self.pad_b),
                "y_buffer_h": self.y_buffer_h, "y_buffer_w": self.y_buffer_w,
                "input_tile_step": self.input_tile_step, "output_tile_step": self.output_tile_step}


class TensorDesc(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("data", c_void_p), ("ndim", c_int32), ("shape", c_int64 * 4)]


class ForwardWarpParams(ctypes.Structure):
    _fields_ = [("B", c_int32), ("H", c_int32), ("W", c_int32), ("divergence", c_double), ("convergence", c_double),
                ("fill", c_int32), ("synthetic_view", c_int32), ("width_base", c_int32)]


class ProfRecord(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("total_ms", c_double), ("launches", c_int64),
                ("flops", c_double), ("bytes", c_double)]


# name -> (restype, argtypes); every symbol include/nunif_hip.h declares
SIGNATURES = {
    "nunif_hip_abi_version": (c_int32, []),
    "nunif_hip_last_error": (c_char_p, []),
    "nunif_hip_tile_grid_init": (c_int32, [c_int32] * 6 + [ctypes.POINTER(TileGrid)]),
    "nunif_hip_blend_ramp": (c_int32, [c_int32, ctypes.POINTER(c_float)]),
    "nunif_hip_gather_tiles": (c_int32, [c_void_p, c_void_p, ctypes.POINTER(TileGrid), c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_stitch_tiles": (c_int32, [c_void_p, c_void_p, ctypes.POINTER(TileGrid), c_int32, c_void_p]),
    "nunif_hip_swin_unet_create": (c_int32, [ctypes.POINTER(TensorDesc), c_int32, c_int32, ctypes.POINTER(c_void_p)]),
    "nunif_hip_swin_unet_destroy": (None, [c_void_p]),
    "nunif_hip_swin_unet_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "nunif_hip_swin_unet_render": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_swin_unet_render_tile_rows": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_swin_unet_tile_row_band": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "nunif_hip_swin_unet_stitch_rows": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_swin_unet_v2_create": (c_int32, [ctypes.POINTER(TensorDesc), c_int32, c_int32, ctypes.POINTER(c_void_p)]),
    "nunif_hip_swin_unet_v2_destroy": (None, [c_void_p]),
    "nunif_hip_swin_unet_v2_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_cunet_create": (c_int32, [ctypes.POINTER(TensorDesc), c_int32, c_int32, ctypes.POINTER(c_void_p)]),
    "nunif_hip_cunet_destroy": (None, [c_void_p]),
    "nunif_hip_cunet_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "nunif_hip_cunet_render": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_forward_warp": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         ctypes.POINTER(ForwardWarpParams), c_void_p]),
    "nunif_hip_backward_warp": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 6 +
                                [c_double, c_double, c_int32, c_void_p]),
    "nunif_hip_row_flow_create": (c_int32, [ctypes.POINTER(TensorDesc), c_int32, ctypes.POINTER(c_void_p)]),
    "nunif_hip_row_flow_destroy": (None, [c_void_p]),
    "nunif_hip_row_flow_delta": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_delta_warp": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 6 + [c_double, c_int32, c_void_p]),
    "nunif_hip_mlbw_create": (c_int32, [ctypes.POINTER(TensorDesc), c_int32, ctypes.POINTER(c_void_p)]),
    "nunif_hip_mlbw_destroy": (None, [c_void_p]),
    "nunif_hip_mlbw_num_layers": (c_int32, [c_void_p]),
    "nunif_hip_mlbw_delta": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_light_inpaint_create": (c_int32, [ctypes.POINTER(TensorDesc), c_int32, ctypes.POINTER(c_void_p)]),
    "nunif_hip_light_inpaint_destroy": (None, [c_void_p]),
    "nunif_hip_light_inpaint_infer": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p]),
    "nunif_hip_light_inpaint_infer_ex": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p]),
    "nunif_hip_anaglyph": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_equirectangular": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_mlbw_has_hole_mask": (c_int32, [c_void_p]),
    "nunif_hip_mlbw_delta_mask": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                            c_void_p]),
    "nunif_hip_hole_mask_postprocess": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 5 + [c_float, c_int32, c_int32,
                                                  c_void_p, c_int32, c_void_p]),
    "nunif_hip_delta_weight_warp": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 7 +
                                    [c_double, c_int32, c_void_p]),
    "nunif_hip_depth_aa_create": (c_int32, [ctypes.POINTER(TensorDesc), c_int32, ctypes.POINTER(c_void_p)]),
    "nunif_hip_depth_aa_destroy": (None, [c_void_p]),
    "nunif_hip_depth_aa_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_depth_anything_create": (c_int32, [ctypes.POINTER(TensorDesc), c_int32, ctypes.POINTER(c_void_p)]),
    "nunif_hip_depth_anything_create_ex": (c_int32, [ctypes.POINTER(TensorDesc), c_int32, ctypes.POINTER(c_int32), ctypes.c_float,
                                                     ctypes.POINTER(c_void_p)]),
    "nunif_hip_depth_anything_destroy": (None, [c_void_p]),
    "nunif_hip_depth_anything_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_depth_anything_reset_state": (c_int32, [c_void_p]),
    "nunif_hip_depth_anything_is_temporal": (c_int32, [c_void_p]),
    "nunif_hip_tta_view": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_tta_merge": (c_int32, [ctypes.POINTER(c_void_p), c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_alpha_border_padding": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_resize_aa": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64] + [c_int32] * 7 +
                            [ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p]),
    "nunif_hip_dilate_edge": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 5 + [c_void_p]),
    "nunif_hip_dilate_edge_work_floats": (ctypes.c_int64, [c_int32, c_int32, c_int32]),
    "nunif_hip_minmax_normalize": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "nunif_hip_mask_morphology": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 6 + [c_void_p]),
    "nunif_hip_reflection_pad2d": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                             c_void_p]),
    "nunif_hip_depth_postprocess": (c_int32, [c_void_p, c_void_p, c_int64, c_float, c_int32, c_float, c_int32, c_void_p]),
    "nunif_hip_frame_to_tensor": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_stereo_to_frame": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_stereo_compose": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "nunif_hip_map_depth": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_double, c_double, c_void_p]),
    "nunif_hip_swin_unet_debug_taps": (c_int32, [c_void_p, c_int32]),
    "nunif_hip_swin_unet_get_tap": (c_int32, [c_void_p, c_int32, c_char_p, c_int32, c_void_p, c_int64,
                                              ctypes.POINTER(c_int64)]),
    "nunif_hip_minmax": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "nunif_hip_ema_scaler_push": (c_int32, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_double, c_void_p]),
    "nunif_hip_ema_scaler_ring_minmax": (c_int32, [c_void_p, c_int32, c_void_p]),
    "nunif_hip_range_normalize": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "nunif_hip_make_input_planes": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_double, c_double, c_int32, c_void_p]),
    "nunif_hip_stack": (c_int32, [ctypes.POINTER(c_void_p), c_int32, c_int64, c_void_p, c_void_p]),
    "nunif_hip_profile_enable": (c_int32, [c_int32]),
    "nunif_hip_profile_read": (c_int32, [ctypes.POINTER(ProfRecord), c_int32, c_int32]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  It must be in the process BEFORE this
        # library is dlopen'ed so that both bind to ONE HIP runtime (streams and pointers are shared); loading
        # /opt/rocm's copy first gives two runtimes and hipMalloc fails.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP engine is not built. Run `python -m nunif_amd.build` "
                "(there is no CPU fallback).")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == ABI mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(status):
    if status != 0:
        raise NunifHipError(status, lib().nunif_hip_last_error().decode("utf-8", "replace"))


def current_stream_ptr(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def tile_grid(x_h, x_w, scale, offset, tile_size, blend_size):
    g = TileGrid()
    check(lib().nunif_hip_tile_grid_init(x_h, x_w, scale, offset, tile_size, blend_size or 0, ctypes.byref(g)))
    return g


def blend_ramp(blend_size):
    buf = (c_float * max(1, blend_size))()
    check(lib().nunif_hip_blend_ramp(blend_size, buf))
    return list(buf)[:blend_size]


def profile_enable(on=True):
    check(lib().nunif_hip_profile_enable(1 if on else 0))


def profile_read(reset=True):
    recs = (ProfRecord * 64)()
    n = lib().nunif_hip_profile_read(recs, 64, 1 if reset else 0)
    return [{"name": recs[i].name.decode(), "total_ms": recs[i].total_ms, "launches": recs[i].launches,
             "flops": recs[i].flops, "bytes": recs[i].bytes} for i in range(n)]
