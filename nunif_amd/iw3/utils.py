"""Per-frame stereo glue on the HIP engine.  Mirrors the hot-path functions of ``iw3/utils.py`` (reference):
``apply_divergence`` :292-391 (mapper + method dispatch; forward / forward_fill / grid_sample / backward / NULL),
``postprocess_image`` :430-487 (IPD pad, ``postprocess_padding`` :394-427, VR180 projection, half-SBS / half-TB / half-RGBD
bicubic-antialias resize, anaglyph / SBS / TB / cross-eyed compose, max-output resize), ``apply_rgbd`` :74-88 and ``nunif/utils/video.py`` ``to_tensor`` / ``to_frame`` :218-269 (``to_frame_tensor`` here returns
the quantised HWC tensor; wrapping it into an ``av.VideoFrame`` is the caller's codec business).
``row_flow_v3`` (the default method), ``mlbw_l2/l4/l2s/l4s`` and the inpaint side models (``forward_inpaint`` ->
``nunif_amd.iw3.forward_inpaint.ForwardInpaint``, ``mlbw_l2_inpaint`` -> ``nunif_amd.iw3.mlbw_inpaint.MLBWInpaint``) run on the
engine."""
import torch
import torch.nn.functional as F

from . import _ops
from .anaglyph import apply_anaglyph_redcyan
from .backward_warp import apply_divergence_grid_sample, apply_divergence_nn_LR
from .equirectangular import equirectangular_projection
from .forward_warp import apply_divergence_forward_warp
from .mapper import get_mapper


# iw3/utils.py:48-51
ROW_FLOW_V2_MAX_DIVERGENCE = 2.5
ROW_FLOW_V3_MAX_DIVERGENCE = 5.0
ROW_FLOW_V2_AUTO_STEP_DIVERGENCE = 2.0
ROW_FLOW_V3_AUTO_STEP_DIVERGENCE = 4.0


def calc_auto_warp_steps(method, divergence, synthetic_view):
    """``iw3/utils.py:2179-2186``: the warp_steps the reference's CLI fills in when ``--warp-steps`` is not given
    (``set_state_args`` :2314-2316): beyond the divergence a flow net was trained for, the warp is split into steps."""
    import math
    divergence = divergence if synthetic_view == "both" else divergence * 2
    if method == "row_flow_v2" and divergence > ROW_FLOW_V2_MAX_DIVERGENCE:
        return math.ceil(divergence / ROW_FLOW_V2_AUTO_STEP_DIVERGENCE)
    if method in {"row_flow", "row_flow_v3"} and divergence > ROW_FLOW_V3_MAX_DIVERGENCE:
        return math.ceil(divergence / ROW_FLOW_V3_AUTO_STEP_DIVERGENCE)
    return None


def apply_divergence(depth, im, args, side_model, reset_pts=None):
    batch = depth.ndim == 4
    if not batch:
        depth, im = depth.unsqueeze(0), im.unsqueeze(0)
    state = getattr(args, "state", None)
    if isinstance(state, dict) and state.get("convergence_model") is not None:
        # iw3/utils.py:303-307: --convergence auto feeds a per-frame convergence TENSOR (ConvergenceEstimator on a U2NETP
        # saliency net) through the mapper into every warp; the warp kernels here take a scalar.  Refuse instead of
        # silently using args.convergence (same policy as --autocrop).
        raise NotImplementedError("--convergence auto (args.state['convergence_model']) is not supported by the HIP engine")
    depth = get_mapper(args.mapper)(depth)
    convergence = args.convergence
    if args.method == "NULL":
        left, right = im.clone(), im.clone()
    elif args.method in {"grid_sample", "backward"}:
        left, right = apply_divergence_grid_sample(im, depth, args.divergence, convergence=convergence,
                                                   synthetic_view=args.synthetic_view)
    elif args.method in {"forward", "forward_fill"}:
        left, right = apply_divergence_forward_warp(im, depth, args.divergence, convergence=convergence,
                                                    method=args.method, synthetic_view=args.synthetic_view,
                                                    width_base=False)
    elif args.method in {"row_flow_v3", "row_flow", "row_flow_v3_sym", "row_flow_sym", "mlbw_l2", "mlbw_l4", "mlbw_l2s",
                         "mlbw_l4s"}:
        # iw3/utils.py:369-387: optional --stereo-width resize of the depth, then the NN backward warp
        if side_model is None:
            raise ValueError(f"method={args.method} needs a side model (nunif_amd.iw3.models.row_flow_v3.RowFlowV3)")
        stereo_width = getattr(args, "stereo_width", None)
        if stereo_width is not None:
            H, W = im.shape[2:]
            stereo_width = min(W, stereo_width)
            if depth.shape[3] != stereo_width:
                new_h = int(H * (stereo_width / W))
                depth = _ops.resize_aa(depth, (new_h, stereo_width), mode="bilinear", align_corners=True, clamp01=True)
        left, right = apply_divergence_nn_LR(side_model, im, depth, args.divergence, convergence,
                                             getattr(args, "warp_steps", None), synthetic_view=args.synthetic_view,
                                             preserve_screen_border=getattr(args, "preserve_screen_border", False),
                                             enable_amp=not getattr(args, "disable_amp", False))
    elif args.method in {"forward_inpaint", "mlbw_l2_inpaint"}:
        # iw3/utils.py:333-365: one side_model.infer per frame (the video mode keeps a 12-frame queue and answers None until
        # it is full), a flush where a scene ends; whatever came out is concatenated
        if side_model is None:
            raise ValueError(f"method={args.method} needs a side model (nunif_amd.iw3.forward_inpaint.ForwardInpaint / "
                             "nunif_amd.iw3.mlbw_inpaint.MLBWInpaint)")
        g = lambda k, d=None: getattr(args, k, d)     # noqa: E731
        lefts, rights = [], []
        reset_pts = reset_pts if reset_pts is not None else [False] * depth.shape[0]
        for i in range(depth.shape[0]):
            conv_i = convergence[i:i + 1] if torch.is_tensor(convergence) else convergence
            le, ri = side_model.infer(im[i:i + 1], depth[i:i + 1], divergence=args.divergence, convergence=conv_i,
                                      preserve_screen_border=g("preserve_screen_border", False),
                                      synthetic_view=args.synthetic_view, inner_dilation=g("mask_inner_dilation", 0),
                                      outer_dilation=g("mask_outer_dilation", 0), max_width=g("inpaint_max_width"),
                                      enable_amp=not g("disable_amp", False))
            if le is not None:
                lefts.append(le)
                rights.append(ri)
            if reset_pts[i]:
                le, ri = side_model.flush(enable_amp=not g("disable_amp", False))
                if le is not None:
                    lefts.append(le)
                    rights.append(ri)
        if not lefts:
            return None, None
        return (lefts[0], rights[0]) if len(lefts) == 1 else (torch.cat(lefts, dim=0), torch.cat(rights, dim=0))
    else:
        raise NotImplementedError(f"method={args.method}: this side model is not on the HIP engine yet")
    if not batch:
        left, right = left.squeeze(0), right.squeeze(0)
    return left, right


def preprocess_image(x, args):
    """iw3/utils.py:247-271: optional 90-degree rotation, then the --max-output-height cap (bicubic antialias,
    align_corners=True, even sizes) on CHW or BCHW."""
    g = lambda k, d=None: getattr(args, k, d)     # noqa: E731
    if g("rotate_left"):
        x = torch.rot90(x, 1, (-2, -1)).contiguous()
    elif g("rotate_right"):
        x = torch.rot90(x, 3, (-2, -1)).contiguous()
    h, w = x.shape[-2:]
    new_w, new_h = w, h
    if g("max_output_height") is not None and new_h > args.max_output_height:
        new_w = int(args.max_output_height / new_h * new_w)
        new_h = args.max_output_height
    if new_w != w or new_h != h:
        new_h -= new_h % 2
        new_w -= new_w % 2
        single = x.ndim == 3
        y = _ops.resize_aa(x.unsqueeze(0) if single else x, (new_h, new_w), mode="bicubic", align_corners=True, clamp01=True)
        x = y[0] if single else y
    return x


def _zero_pad(x, left, top, right, bottom):
    return F.pad(x, (left, right, top, bottom), mode="constant", value=0.0)


def apply_rgbd(im, depth, mapper):
    """RGBD output (iw3/utils.py:74-88): left = image, right = the (mapped) depth resized bicubic-antialias, 3 channels."""
    height, width = im.shape[-2:]
    if mapper is not None:
        depth = get_mapper(mapper)(depth)
    batch = depth.ndim == 4
    right = _ops.resize_aa(depth if batch else depth.unsqueeze(0), (height, width), mode="bicubic")
    if not batch:
        right = right.squeeze(0)
    return im, right.expand_as(im)


def postprocess_padding(left_eye, right_eye, pad, pad_mode):
    """Zero padding of both eyes (iw3/utils.py:394-427): fractions of the frame (tblr / tb / lr / top) or fit to 16:9."""
    assert pad_mode in {"tblr", "tb", "lr", "16:9", "top"}
    pad_l = pad_t = pad_r = pad_b = 0
    if pad_mode in {"tblr", "tb", "lr"}:
        if "tb" in pad_mode:
            pad_t = pad_b = round(left_eye.shape[1] * pad) // 2
        if "lr" in pad_mode:
            pad_l = pad_r = round(left_eye.shape[2] * pad) // 2
    elif pad_mode == "top":
        pad_t = round(left_eye.shape[1] * pad)
    else:
        target_ratio = 16 / 9
        height, width = left_eye.shape[1:]
        current_ratio = width / height
        if abs(target_ratio - current_ratio) > 1e-3:
            if current_ratio > target_ratio:
                pad_t = pad_b = (round(width / target_ratio) - height) // 2
            else:
                pad_l = pad_r = (round(height * target_ratio) - width) // 2
        else:
            return left_eye, right_eye
    return _zero_pad(left_eye, pad_l, pad_t, pad_r, pad_b), _zero_pad(right_eye, pad_l, pad_t, pad_r, pad_b)


def postprocess_image(left_eye, right_eye, args):
    """CHW, CHW -> CHW float in [0,1]."""
    g = lambda k, d=None: getattr(args, k, d)     # noqa: E731
    ipd_pad = int(abs(g("ipd_offset", 0)) * 0.01 * max(left_eye.shape[-2:]))
    ipd_pad -= ipd_pad % 2
    if ipd_pad > 0 and not (g("rgbd") or g("half_rgbd")):
        pad_o, pad_i = (ipd_pad * 2, ipd_pad) if g("ipd_offset", 0) > 0 else (ipd_pad, ipd_pad * 2)
        left_eye = _zero_pad(left_eye, pad_o, 0, pad_i, 0)
        right_eye = _zero_pad(right_eye, pad_i, 0, pad_o, 0)
    if g("pad") is not None or g("pad_mode") == "16:9":
        left_eye, right_eye = postprocess_padding(left_eye, right_eye, pad=g("pad"), pad_mode=g("pad_mode"))
    if g("vr180"):
        left_eye, right_eye = equirectangular_projection(left_eye), equirectangular_projection(right_eye)
    elif g("half_sbs") or g("half_rgbd"):
        size = (left_eye.shape[1], left_eye.shape[2] // 2)
        left_eye, right_eye = (_ops.resize_aa(e.unsqueeze(0), size, mode="bicubic")[0] for e in (left_eye, right_eye))
    elif g("half_tb"):
        size = (left_eye.shape[1] // 2, left_eye.shape[2])
        left_eye, right_eye = (_ops.resize_aa(e.unsqueeze(0), size, mode="bicubic")[0] for e in (left_eye, right_eye))
    if g("anaglyph") is not None:
        sbs = apply_anaglyph_redcyan(left_eye, right_eye, g("anaglyph"))
        return _max_output_resize(sbs, args)
    layout = "tb" if (g("tb") or g("half_tb")) else ("cross_eyed" if g("cross_eyed") else "sbs")
    sbs = _ops.stereo_compose(left_eye.contiguous(), right_eye.contiguous(), layout)
    return _max_output_resize(sbs, args)


def postprocess_to_frame(left_eye, right_eye, args, use_16bit=False):
    """``to_frame_tensor(postprocess_image(left, right, args))`` — the frame's way out of the scheduler.  In the plain SBS / TB /
    cross-eyed case (no IPD / aspect padding, no VR180 projection, no half-size or anaglyph format, no output-size cap that
    bites) compose + clamp + quantise is ONE kernel (``stereo_to_frame``) and the fp32 side-by-side image is never written:
    62 MB instead of 162 MB of traffic per 1080p frame; the bytes that come out are the same."""
    g = lambda k, d=None: getattr(args, k, d)     # noqa: E731
    h, w = left_eye.shape[-2:]
    ipd_pad = int(abs(g("ipd_offset", 0) or 0) * 0.01 * max(h, w))
    ipd_pad -= ipd_pad % 2
    tb = bool(g("tb"))
    out_h, out_w = (2 * h, w) if tb else (h, 2 * w)
    plain = (left_eye.ndim == 3 and left_eye.shape == right_eye.shape and ipd_pad == 0
             and g("pad") is None and g("pad_mode") != "16:9" and g("anaglyph") is None
             and not (g("vr180") or g("half_sbs") or g("half_rgbd") or g("half_tb"))
             and (g("max_output_height") is None or out_h <= g("max_output_height"))
             and (g("max_output_width") is None or out_w <= g("max_output_width")))
    if plain:
        layout = "tb" if tb else ("cross_eyed" if g("cross_eyed") else "sbs")
        return _ops.stereo_to_frame(left_eye.contiguous(), right_eye.contiguous(), layout, 16 if use_16bit else 8)
    return to_frame_tensor(postprocess_image(left_eye, right_eye, args), use_16bit=use_16bit)


def _max_output_resize(sbs, args):
    g = lambda k, d=None: getattr(args, k, d)     # noqa: E731
    h, w = sbs.shape[1:]
    new_w, new_h = w, h
    if g("max_output_height") is not None and new_h > args.max_output_height:
        if g("keep_aspect_ratio"):
            new_w = int(args.max_output_height / new_h * new_w)
        new_h = args.max_output_height
    if g("max_output_width") is not None and new_w > args.max_output_width:
        if g("keep_aspect_ratio"):
            new_h = int(args.max_output_width / new_w * new_h)
        new_w = args.max_output_width
    if new_w != w or new_h != h:
        new_h -= new_h % 2
        new_w -= new_w % 2
        sbs = _ops.resize_aa(sbs.unsqueeze(0), (new_h, new_w), mode="bicubic", clamp01=True)[0]
    return sbs


def debug_depth_image(depth, args):
    """iw3/utils.py:489-502 (``--debug-depth``): raw | mapped depth side by side as a grey 3-channel image."""
    depth = depth.float()
    out = torch.cat([depth, get_mapper(args.mapper)(depth.unsqueeze(0))[0]], dim=2)
    return out.repeat((3, 1, 1))


def process_image(x, args, depth_model, side_model=None, skip_autocrop=None, autocrop_uncrop=False):
    """The image-mode entry (iw3/utils.py:505-548): ``preprocess_image`` -> depth -> per-image min-max -> stereo method ->
    ``postprocess_image``; CHW float in, CHW float SBS (or the chosen format) out.  ``--autocrop`` (border detection,
    ``nunif/utils/autocrop.py``) is not on the engine and is refused rather than ignored."""
    assert depth_model.get_ema_buffer_size() == 1
    g = lambda k, d=None: getattr(args, k, d)     # noqa: E731
    if g("autocrop") is not None and not skip_autocrop:
        raise NotImplementedError("--autocrop is outside the HIP engine's scope; crop the frame before process_image")
    with torch.inference_mode():
        x = preprocess_image(x, args)
        depth = depth_model.infer(x, tta=g("tta", False), low_vram=g("low_vram", False), enable_amp=not g("disable_amp", False),
                                  edge_dilation=g("edge_dilation", 0), depth_aa=g("depth_aa", False))
        depth = depth_model.minmax_normalize_chw(depth)
        if g("debug_depth"):
            return debug_depth_image(depth, args)
        if g("rgbd") or g("half_rgbd"):
            left_eye, right_eye = apply_rgbd(x, depth, mapper=args.mapper)
            return postprocess_image(left_eye, right_eye, args)
        while True:                                   # a video inpaint side model answers None until its queue is full
            left_eye, right_eye = apply_divergence(depth, x, args, side_model)
            if left_eye is not None:
                break
        if left_eye.ndim == 4:
            left_eye, right_eye = left_eye[0], right_eye[0]
        return postprocess_image(left_eye, right_eye, args)


def to_tensor(frame_hwc, device=None):
    """uint8/uint16 HWC (numpy array or tensor) -> CHW float on the device (video.py:218-223)."""
    if not torch.is_tensor(frame_hwc):
        import numpy as np
        arr = np.ascontiguousarray(frame_hwc)
        frame_hwc = torch.from_numpy(arr.view(np.int16) if arr.dtype == np.uint16 else arr)
    if device is not None:
        frame_hwc = frame_hwc.to(device)
    return _ops.frame_to_tensor(frame_hwc)


def to_frame_tensor(x, use_16bit=False):
    """CHW float -> HWC uint8 (or uint16 bit pattern in int16), (x*max).round() like video.py:236-245."""
    # (stereo frames normally leave through stereo_to_frame directly, which fuses the SBS compose as well)
    return _ops.to_frame(x, 16 if use_16bit else 8)
