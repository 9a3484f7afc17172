#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself (``/root/reference``) on CPU.

Run in the build container only (the reference is not present on the GPU box):

    python tests/golden/make_golden.py [group ...]

Each group writes ``tests/golden/<group>.npz`` (+ ``seam_configs.json``).  Inputs and weights are seeded and
re-creatable (``oracle.*.random_state_dict(seed)``), so only reference *outputs* (and small inputs) are stored.
The fixtures pin the oracle (``tests/test_oracle_golden.py``) and, through it, the HIP path.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import refstub  # noqa: E402

refstub.install()
torch.set_grad_enabled(False)


def sd_checksum(sd):
    return float(sum(v.double().sum().item() for v in sd.values() if v.is_floating_point()))


def synth_image(seed, c, h, w):
    """Smooth + noise image in [0,1] (SURVEY.md §8d synthetic input recipe)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, c, max(2, h // 16 + 1), max(2, w // 16 + 1), generator=g)
    up = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)[0]
    return torch.clamp(up * 0.8 + 0.2 * torch.rand(c, h, w, generator=g), 0, 1)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_seam():
    from nunif.utils.seam_blending import SeamBlending
    cases = []
    for (h, w, s, o, t, b) in [(512, 512, 1, 28, 256, 0), (1080, 1920, 2, 16, 256, 8), (2160, 3840, 4, 32, 256, 16),
                               (1080, 1920, 2, 16, 640, 8), (100, 130, 2, 16, 64, 8), (1, 1, 2, 16, 64, 8),
                               (17, 300, 1, 8, 64, 4), (720, 1280, 4, 32, 112, 16), (333, 777, 2, 36, 256, 0),
                               (64, 64, 2, 16, 64, 8), (49, 48, 1, 8, 64, 4), (1080, 1920, 2, 8, 256, 8)]:
        cfg = SeamBlending.create_config((h, w), s, o, t, b)
        case = {"args": [h, w, s, o, t, b], "config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}}
        if b > 0:
            f = SeamBlending.create_blend_filter(s, o, t, b, 1)[0]
            mid = f.shape[0] // 2
            case["ramp_bits"] = [int(v) for v in f[mid, :b + 1].view(torch.int32)]  # fp32 bit patterns of the edge ramp
            case["filter_sum"] = float(f.double().sum())
            case["corner"] = [float(v) for v in f[:b + 1, :b + 1].reshape(-1)]
        cases.append(case)
    with open(os.path.join(HERE, "seam_configs.json"), "w") as fh:
        json.dump(cases, fh, indent=1)
    print("wrote seam_configs.json", len(cases), "cases")


def gen_swin():
    from waifu2x.models.swin_unet import SwinUNet, SwinUNet2x, SwinUNet4x
    from nunif.utils.render import tiled_render
    from oracle import swin_unet as O
    out = {}
    x = synth_image(11, 3, 64, 64).unsqueeze(0)
    out["x"] = x
    for cls, sf, tag in ((SwinUNet, 1, "1x"), (SwinUNet2x, 2, "2x"), (SwinUNet4x, 4, "4x")):
        sd = O.random_state_dict(100 + sf, sf)
        m = cls().eval()
        m.load_state_dict(sd, strict=True)
        out["y_" + tag] = m(x)
        out["sdsum_" + tag] = sd_checksum(sd)
        if sf == 4:
            out["y_4x_to2x"] = m.to_2x().eval()(x)
            out["y_4x_to1x"] = m.to_1x().eval()(x)
    # shift-mask edge case: a 28x28 tile gives a 12x12 level-1 map, 6x6 at level 2 (shift disabled), 3x3 would
    # break the %6 assumption -> the smallest legal tile is 64; use a second, different image instead
    x2 = synth_image(12, 3, 112, 112).unsqueeze(0)
    sd = O.random_state_dict(102, 2)
    m = SwinUNet2x().eval()
    m.load_state_dict(sd, strict=True)
    out["x_112"] = x2
    out["y_2x_112"] = m(x2)
    # tiled render through the reference stitcher (ragged size; 2x3 tiles of 64)
    img = synth_image(13, 3, 100, 130)
    out["img"] = img
    out["render_2x_t64_b4"] = tiled_render(img, m, tile_size=64, batch_size=4)
    save("swin_unet", **out)


def gen_swin8x():
    """waifu2x.swin_unet_8x (two-layer ToImage head, pixel_shuffle 8) on the reference; stored as fp16."""
    from waifu2x.models.swin_unet import SwinUNet8x
    from oracle import swin_unet as O
    x = synth_image(14, 3, 64, 64).unsqueeze(0)
    sd = O.random_state_dict(108, 8)
    m = SwinUNet8x().eval()
    m.load_state_dict(sd, strict=True)
    y = m(x)
    print(tuple(y.shape), float(y.mean()), float(y.std()), float((y <= 0).float().mean()), float((y >= 1).float().mean()))
    save("swin_unet_8x", x=x, y=y.half(), sdsum=sd_checksum(sd))


def gen_swin4xl():
    """waifu2x.swin_unet_4xl (base_dim 192, 12 heads, LayerNormNoBias; swin_unet.py:390-394) and the LayerNorm variant of the
    2x net at base_dim 96 on the reference; outputs stored as fp16."""
    from waifu2x.models.swin_unet import swin_unet_4xl, SwinUNet2x
    from oracle import swin_unet as O
    x = synth_image(15, 3, 64, 64).unsqueeze(0)
    sd = O.random_state_dict(204, 4, base_dim=192, layer_norm=True)
    m = swin_unet_4xl().eval()
    m.load_state_dict(sd, strict=True)
    y = m(x)
    print(tuple(y.shape), float(y.mean()), float(y.std()), float((y <= 0).float().mean()), float((y >= 1).float().mean()))
    sd2 = O.random_state_dict(205, 2, base_dim=96, layer_norm=True)
    m2 = SwinUNet2x(layer_norm=True).eval()
    m2.load_state_dict(sd2, strict=True)
    y2 = m2(x)
    print(tuple(y2.shape), float(y2.mean()), float(y2.std()), float((y2 <= 0).float().mean()), float((y2 >= 1).float().mean()))
    save("swin_unet_4xl", x=x, y=y.half(), sdsum=sd_checksum(sd), y2_ln=y2.half(), sdsum2=sd_checksum(sd2))


def fake_vda_net(frame):
    """Deterministic stand-in for the external streaming net: a smooth positive 'metric depth' of the frame (0.3 .. 12)."""
    g = frame.mean(dim=0, keepdim=True)
    g = torch.nn.functional.avg_pool2d(g.unsqueeze(0), 5, stride=1, padding=2)[0]
    return 0.3 + 6.0 * torch.sigmoid(g) ** 2 + 0.002 * torch.arange(g.shape[-1]).view(1, 1, -1)


class FakeOnlineVDA:
    """Stand-in for the external ``VideoDepthAnythingOnline``: buffers frames and emits them 4 at a time (so outputs lag the inputs
    and a drain with ``infer(None)`` pads with copies of the last frame, as the windowed model does)."""
    metric_depth = False

    def __init__(self, net):
        self.net, self.buf, self.last, self.prep_lower_bound = net, [], None, None

    def reset_state(self):
        self.buf, self.last = [], None

    def infer(self, frame, use_amp=True):
        if frame is None:
            frame = self.last
        self.last = frame
        self.buf.append(frame)
        if len(self.buf) < 4:
            return None
        out, self.buf = [self.net(f)[0] for f in self.buf], []
        return out


def gen_vda_online():
    """The non-streaming ``VideoDepthAnythingModel`` wrapper (frame counting, drain / unpad at a scene cut, EMA normalisation)
    on the reference, around ``FakeOnlineVDA``: 11 frames in batches of 3, a scene cut after frame 5, final flush."""
    from iw3 import video_depth_anything_model as RV
    x = torch.stack([synth_image(51 + i, 3, 54, 96) for i in range(11)]).half().float()
    m = RV.VideoDepthAnythingModel("VDA_S")
    m.model, m.device = FakeOnlineVDA(fake_vda_net), torch.device("cpu")
    m.model.prep_lower_bound = 56
    m.enable_ema(0.75, buffer_size=2)
    outs, counts = [], []
    for i in range(0, 11, 3):
        got = m.infer_with_normalize(x[i:i + 3], list(range(i, min(i + 3, 11))), reset_pts={5}, edge_dilation=2)
        counts.append(len(got))
        outs += got
    tail = m.flush_with_normalize(edge_dilation=2)
    counts.append(len(tail))
    outs += tail
    assert len(outs) == 11, (len(outs), counts)
    print("vda_online counts", counts)
    save("video_depth_anything_online", x=x.half(), out=torch.stack(outs), counts=np.asarray(counts))


def gen_vda():
    """VideoDepthAnything pre/post-processing on the reference (video_depth_anything_model.py batch_preprocess / postprocess,
    the streaming wrapper's per-frame loop) around a deterministic fake network; relative + metric, DepthAA on / off."""
    from iw3 import video_depth_anything_model as RV
    from iw3.models.depth_aa import DepthAA
    from oracle import depth_aa as ODA
    out = {}
    x = torch.stack([synth_image(31 + i, 3, 54, 96) for i in range(5)]).half().float()     # stored as fp16, exactly
    out["x"] = x
    sd = ODA.random_state_dict(501)
    aa = DepthAA().eval()
    aa.load_state_dict(sd, strict=True)
    for tag, metric, lb in (("rel", False, 56), ("met", True, 84)):
        pre = RV.batch_preprocess(x.clone(), lb, metric_depth=metric)
        out["pre_" + tag] = pre
        raw = torch.stack([fake_vda_net(f) for f in pre]).squeeze(1)
        raw[0, 3, 5] = float("nan")
        if metric:              # (+inf in a relative map becomes FLT_MAX and overflows the reference's own dilation to NaN)
            raw[1, 7, 9] = float("inf")
        out["raw_" + tag] = raw
        out["post_" + tag] = RV.postprocess(raw.clone(), edge_dilation=2, depth_aa=None, metric_depth=metric, force_disparity=True)
        out["post_aa_" + tag] = RV.postprocess(raw.clone(), edge_dilation=[2, 1], depth_aa=aa, metric_depth=metric,
                                                force_disparity=True, enable_amp=False)
        print(tag, tuple(pre.shape), tuple(out["post_" + tag].shape), float(out["post_" + tag].min()), float(out["post_" + tag].max()))
    out["post_met_nodisp"] = RV.postprocess(out["raw_met"].clone(), edge_dilation=1, depth_aa=None, metric_depth=True,
                                            force_disparity=False, max_dist=5.0)
    save("video_depth_anything", **{k: (v.half() if k.startswith("x") else v) for k, v in out.items()})


def gen_morph():
    """iw3/dilation.py mask morphology + iw3/mapper.py named mappers on the reference (tiny, exact)."""
    from iw3 import dilation as RD
    from iw3 import mapper as RM
    g = torch.Generator().manual_seed(77)
    m = (torch.rand(3, 1, 23, 37, generator=g) > 0.8)
    mf = m.float()
    out = {"mask": m.numpy()}
    out["dilate"], out["erode"] = RD.dilate(mf), RD.erode(mf)
    out["closing2"], out["closing1"] = RD.closing(m), RD.closing(m, n_iter=1)
    out["mask_closing2"] = RD.mask_closing(m)
    out["outer3"], out["inner2"] = RD.dilate_outer(m, 3), RD.dilate_inner(m, 2)
    out["outer_bw"] = RD.dilate_outer(m, 4, base_width=74)         # round(37 / 74 * 4) = 2
    x = torch.rand(2, 1, 9, 11, generator=g)
    out["x"] = x
    out["softplus01"], out["inv_softplus01"] = RM.softplus01(x, 0.343, 12), RM.inv_softplus01(x, -0.002102, 7.8788)
    out["softplus01_legacy"] = RM.softplus01_legacy(x, 6)
    out["distance_to_disparity"], out["shift_relative_depth"] = RM.distance_to_disparity(x, 0.6), RM.shift_relative_depth(x, 1.4)
    # forward_warp.nonwarp_mask (:259-295) and backward_warp / make_grid / pad_delta_y (:67-93, :239-243)
    from iw3 import forward_warp as RF
    from iw3 import backward_warp as RB
    from oracle.forward_warp import synth_depth
    c = synth_image(81, 3, 40, 64).unsqueeze(0)
    depth = synth_depth(13, 1, 40, 64, "smooth_edges")
    out["fw_c"], out["fw_depth"] = c, depth
    for view in ("right", "left"):
        cc, mask = RF.nonwarp_mask(c.clone(), depth.clone(), 16.0, 0.5, view=view)
        out["fw_mask_" + view] = mask
        assert torch.equal(cc, c)
    delta_x = (torch.rand(1, 1, 20, 32, generator=g) - 0.5) * 6.0
    grid = RB.make_grid(1, 32, 20, delta_x.device)
    out["bw_delta_x"] = delta_x
    out["bw_out"] = RB.backward_warp(c, grid, RB.pad_delta_y(delta_x), 1.0 / (64 // 2 - 1))
    save("morph", **out)


def gen_light_inpaint():
    """inpaint.light_inpaint_v1 on the reference (infer / forward) + the MLBWInpaintImage flow (mask MLBW warp, hole mask,
    inpaint, left eye processed flipped) assembled from the reference's own functions."""
    from iw3.models.light_inpaint_v1 import LightInpaintV1
    from iw3.models.mlbw import MLBW
    from iw3 import mlbw_inpaint as RI
    from oracle import light_inpaint as OL, mlbw as OM
    from oracle.forward_warp import synth_depth
    out = {}
    sd = OL.random_state_dict(701)
    m = LightInpaintV1().eval()
    m.load_state_dict(sd, strict=True)
    out["sdsum"] = sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    x = torch.stack([synth_image(111, 3, 70, 100), synth_image(112, 3, 70, 100)])
    g = torch.Generator().manual_seed(113)
    mask = torch.rand(2, 1, 70, 100, generator=g) > 0.93
    mask[0, :, 20:42, 30:55] = True
    mask[1, :, 5:12, 60:100] = True
    out["x"], out["mask"] = x, mask
    out["infer"] = m.infer(x, mask)
    out["infer_close"] = m.infer(x, mask, closing=True, inner_dilation=1, outer_dilation=2, base_width=50)
    out["forward_off"] = m(x, mask.float())                      # with the i2i offset crop (training-style call)
    print("infer", float(out["infer"].std()), float((out["infer"] - x).abs().mean()),
          float(((out["infer"] <= 0) | (out["infer"] >= 1)).float().mean()))
    # MLBWInpaintImage.forward :118-157 without the model downloads
    sdm = OM.random_state_dict(431, 2, False, hole_mask=True)
    mm = MLBW(num_layers=2, base_dim=32, hole_mask=True).eval()
    mm.load_state_dict(sdm, strict=True)
    mm.delta_output = True
    depth = synth_depth(9, 2, 58, 104, "smooth_edges")
    c = torch.stack([synth_image(84, 3, 116, 208), synth_image(85, 3, 116, 208)])
    out["depth"], out["c"] = depth, c
    le, re, lm, rm = RI.apply_divergence(mm, c, depth, 2.0, 0.5, False, "both", False)
    kw = dict(inner_dilation=1, outer_dilation=1, base_width=depth.shape[-1])
    out["mi_left"] = RI.forward_left(m, le, lm, **kw).half()
    out["mi_right"] = RI.forward_right(m, re, rm, **kw).half()
    le, re, lm, rm = RI.apply_divergence(mm, c[:1], depth[:1], 2.0, 0.5, False, "right", False)
    out["mi_right_only"] = RI.forward_right(m, re, rm, inner_dilation=0, outer_dilation=0, base_width=104).half()
    save("light_inpaint", **out)


def gen_forward_inpaint():
    """iw3/forward_inpaint.py on the reference (``--method forward_inpaint``): un-filled forward warp + hole masks, mask closing /
    directional dilation, light_inpaint_v1 per eye (left eye mirrored) — image mode (both eyes, right only, max_width) — and
    the 12-frame ForwardInpaintVideo queue (batches of 3, flush) around seeded models."""
    from iw3.models.light_inpaint_v1 import LightInpaintV1
    from iw3.models.light_video_inpaint_v1 import LightVideoInpaintV1
    from iw3 import forward_inpaint as RF
    from oracle import light_inpaint as OL
    from oracle.forward_warp import synth_depth
    out = {}
    m = LightInpaintV1().eval()
    m.load_state_dict(OL.random_state_dict(701), strict=True)
    img = object.__new__(RF.ForwardInpaintImage)
    torch.nn.Module.__init__(img)
    img.model = m
    depth = synth_depth(19, 2, 58, 104, "smooth_edges")
    c = torch.stack([synth_image(184, 3, 116, 208), synth_image(185, 3, 116, 208)])
    out["depth"], out["c"] = depth, c
    le, re = img.infer(c, depth, divergence=2.5, convergence=0.5, synthetic_view="both", inner_dilation=1, outer_dilation=2)
    out["fi_left"], out["fi_right"] = le.half(), re.half()
    le, re = img.infer(c[:1], depth[:1], divergence=2.0, convergence=0.3, synthetic_view="right")
    assert torch.equal(le, c[:1])
    out["fi_right_only"] = re.half()
    le, re = img.infer(c[1:], depth[1:], divergence=2.0, convergence=0.5, synthetic_view="left", max_width=150)
    out["fi_left_mw"], out["fi_right_mw"] = le.half(), re.half()
    print("image", float((out["fi_left"].float() - c).abs().mean()), tuple(out["fi_left_mw"].shape))
    # video queue
    mv = LightVideoInpaintV1().eval()
    mv.load_state_dict(OL.video_random_state_dict(801), strict=True)
    vid = object.__new__(RF.ForwardInpaintVideo)
    torch.nn.Module.__init__(vid)
    vid.model, vid.model_seq, vid.pre_padding, vid.post_padding = mv, 12, 3, 3
    vid.frame_queue = vid.synthetic_view = vid.inner_dilation = vid.outer_dilation = vid.base_width = None
    n_frames = 15
    wide = synth_image(231, 3, 44, 80 + n_frames)
    frames = torch.stack([wide[:, :, i:i + 80] for i in range(n_frames)])
    vdepth = synth_depth(25, 1, 22, 40, "smooth_edges").expand(n_frames, 1, 22, 40).clone()
    vdepth = (vdepth + torch.linspace(0, 0.2, n_frames).view(-1, 1, 1, 1)).clamp(0, 1)
    out["v_frames"], out["v_depth"] = frames, vdepth
    lefts, rights, sizes = [], [], []
    for i in range(0, n_frames, 3):
        le, ri = vid.infer(frames[i:i + 3], vdepth[i:i + 3], divergence=2.0, convergence=0.5, synthetic_view="both",
                           inner_dilation=1, outer_dilation=1)
        sizes.append(0 if le is None else le.shape[0])
        if le is not None:
            lefts.append(le.clone()); rights.append(ri.clone())
    le, ri = vid.flush()
    sizes.append(0 if le is None else le.shape[0])
    if le is not None:
        lefts.append(le.clone()); rights.append(ri.clone())
    out["v_sizes"] = np.asarray(sizes)
    out["v_left"], out["v_right"] = torch.cat(lefts).half(), torch.cat(rights).half()
    print("video queue sizes", sizes, tuple(out["v_left"].shape))
    save("forward_inpaint", **out)


def gen_swin_v2():
    """waifu2x.swin_unet_v2_{1x,2x,4x} on the reference (waifu2x/models/swin_unet_v2.py): constructor weights under a seed with
    every bias / norm weight re-drawn (oracle.swin_unet_v2.randomize), one 64 x 64 tile batch each; stored: state-dict checksum,
    input, output.  The HIP engine does not carry this family yet — the fixture pins the oracle the kernels will be held to."""
    from waifu2x.models import swin_unet_v2 as RV
    from oracle import swin_unet_v2 as OV
    out = {}
    x = torch.stack([synth_image(301, 3, 64, 64), synth_image(302, 3, 64, 64)])
    out["x"] = x
    for tag, cls, seed in (("1x", RV.SwinUNet1xV2, 11), ("2x", RV.SwinUNet2xV2, 12), ("4x", RV.SwinUNet4xV2, 13)):
        torch.manual_seed(seed)
        m = cls().eval()
        sd = OV.randomize(m.state_dict(), seed + 100)
        m.load_state_dict(sd, strict=True)
        y = m(x)
        print(tag, tuple(y.shape), float(y.std()), float(((y <= 0) | (y >= 1)).float().mean()), m.i2i_offset, m.i2i_scale)
        out["y_" + tag], out["raw_" + tag] = y, m.unet(x)          # clamped model output, un-clamped network output
        out["sdsum_" + tag] = sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    save("swin_unet_v2", **out)


def gen_light_video_inpaint_ml():
    """inpaint.light_video_inpaint_v1_medium / _large (base_dim 128 / 192, lv2_mlp_ratio 2; light_video_inpaint_v1.py:230-246)
    on the reference: one 12-frame infer each; outputs stored as fp16."""
    from iw3.models.light_video_inpaint_v1 import LightVideoInpaintV1Medium, LightVideoInpaintV1Large
    from oracle import light_inpaint as OL
    base = synth_image(141, 3, 40, 84)
    x = torch.stack([base[:, :, i:i + 72] for i in range(12)]).half().float()
    g = torch.Generator().manual_seed(125)
    mask = torch.rand(12, 1, 40, 72, generator=g) > 0.94
    for i in range(12):
        mask[i, :, 8:24, 22 + i:46 + i] = True
    out = {"x": x.half(), "mask": mask.numpy()}
    for tag, cls, seed, dim in (("medium", LightVideoInpaintV1Medium, 811, 128), ("large", LightVideoInpaintV1Large, 812, 192)):
        sd = OL.video_random_state_dict(seed, base_dim=dim, lv2_mlp_ratio=2)
        m = cls().eval()
        m.load_state_dict(sd, strict=True)
        y = m.infer(x, mask, inner_dilation=1)
        print(tag, float(y.std()), float((y - x).abs().mean()), float(((y <= 0) | (y >= 1)).float().mean()))
        out["y_" + tag] = y.half()
        out["sdsum_" + tag] = sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    save("light_video_inpaint_ml", **out)


def gen_light_video_inpaint():
    """inpaint.light_video_inpaint_v1 on the reference (12-frame infer, a 7-frame batch padded by infer itself) and the
    MLBWInpaintVideo queue (pre / post padding 3, batches of 3 frames, flush) assembled around seeded models."""
    from iw3.models.light_video_inpaint_v1 import LightVideoInpaintV1
    from iw3.models.mlbw import MLBW
    from iw3 import mlbw_inpaint as RI
    from oracle import light_inpaint as OL, mlbw as OM
    from oracle.forward_warp import synth_depth
    out = {}
    sd = OL.video_random_state_dict(801)
    m = LightVideoInpaintV1().eval()
    m.load_state_dict(sd, strict=True)
    out["sdsum"] = sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    base = synth_image(121, 3, 40, 84)
    x = torch.stack([base[:, :, i:i + 72] for i in range(12)])              # a panning shot
    g = torch.Generator().manual_seed(123)
    mask = torch.rand(12, 1, 40, 72, generator=g) > 0.94
    for i in range(12):
        mask[i, :, 10:26, 20 + i:44 + i] = True
    out["x"], out["mask"] = x, mask
    out["infer12"] = m.infer(x, mask)
    out["infer7"] = m.infer(x[:7], mask[:7], closing=True, inner_dilation=1, outer_dilation=1, base_width=36)
    y = out["infer12"]
    print("infer12", float(y.std()), float((y - x).abs().mean()), float(((y <= 0) | (y >= 1)).float().mean()))
    # MLBWInpaintVideo :160-293 without the downloads
    sdm = OM.random_state_dict(431, 2, False, hole_mask=True)
    mm = MLBW(num_layers=2, base_dim=32, hole_mask=True).eval()
    mm.load_state_dict(sdm, strict=True)
    mm.delta_output = True
    vid = object.__new__(RI.MLBWInpaintVideo)
    torch.nn.Module.__init__(vid)
    vid.model, vid.mask_mlbw = m, mm
    vid.model_seq, vid.pre_padding, vid.post_padding = 12, 3, 3
    vid.frame_queue = vid.synthetic_view = vid.inner_dilation = vid.outer_dilation = vid.base_width = None
    vid.device = torch.device("cpu")
    n_frames = 18
    wide = synth_image(131, 3, 44, 80 + n_frames)
    frames = torch.stack([wide[:, :, i:i + 80] for i in range(n_frames)])
    depth = synth_depth(15, 1, 22, 40, "smooth_edges").expand(n_frames, 1, 22, 40).clone()
    depth = (depth + torch.linspace(0, 0.2, n_frames).view(-1, 1, 1, 1)).clamp(0, 1)
    out["v_frames"], out["v_depth"] = frames, depth
    lefts, rights, sizes = [], [], []
    for i in range(0, n_frames, 3):          # the queue only lands exactly on 12 with batches of 1 or 3 (3 + 3k, then 6 + 3k)
        le, ri = vid.infer(frames[i:i + 3], depth[i:i + 3], divergence=2.0, convergence=0.5, synthetic_view="both",
                           inner_dilation=1, outer_dilation=1, enable_amp=False)
        sizes.append(0 if le is None else le.shape[0])
        if le is not None:
            lefts.append(le.clone()); rights.append(ri.clone())
    le, ri = vid.flush(enable_amp=False)
    sizes.append(0 if le is None else le.shape[0])
    if le is not None:
        lefts.append(le.clone()); rights.append(ri.clone())
    out["v_sizes"] = np.asarray(sizes)
    out["v_left"], out["v_right"] = torch.cat(lefts).half(), torch.cat(rights).half()
    print("video queue sizes", sizes, tuple(out["v_left"].shape))
    save("light_video_inpaint", **out)


def gen_iw3():
    from iw3.forward_warp import apply_divergence_forward_warp
    from iw3.backward_warp import apply_divergence_grid_sample
    from iw3.dilation import dilate_edge
    from iw3.depth_anything_model import batch_preprocess
    from oracle.forward_warp import synth_depth
    out = {}
    c = synth_image(31, 3, 64, 160).unsqueeze(0)
    d = synth_depth(32, 1, 64, 160, "edges")
    d_small = synth_depth(33, 1, 32, 80, "smooth_edges")
    out["c"], out["depth"], out["depth_small"] = c, d, d_small
    for tag, depth in (("full", d), ("small", d_small)):
        le, ri, lm, rm = apply_divergence_forward_warp(c.clone(), depth.clone(), 40.0, 0.5, method="forward_fill",
                                                       synthetic_view="both", return_mask=True, width_base=False)
        out[f"fw_{tag}_left"], out[f"fw_{tag}_right"], out[f"fw_{tag}_lmask"], out[f"fw_{tag}_rmask"] = le, ri, lm, rm
    le, ri = apply_divergence_forward_warp(c.clone(), d.clone(), 12.0, 0.2, method="forward", synthetic_view="both")
    out["fw_nofill_left"], out["fw_nofill_right"] = le, ri
    _, ri = apply_divergence_forward_warp(c.clone(), d.clone(), 8.0, 0.5, method="forward_fill", synthetic_view="right")
    out["fw_right_only"] = ri
    le, ri = apply_divergence_grid_sample(c, d, 2.5, 0.3, "both")
    out["gs_left"], out["gs_right"] = le, ri
    le, ri = apply_divergence_grid_sample(c, d_small, 2.5, 0.3, "both")
    out["gs_small_left"], out["gs_small_right"] = le, ri
    raw = synth_depth(34, 2, 56, 98, "smooth_edges") * 7.0 + 0.5      # un-normalised network-like output
    out["raw_depth"] = raw
    out["dilate_2_1"] = dilate_edge(raw.clone(), [2, 1])
    out["dilate_1_3"] = dilate_edge(raw.clone(), [1, 3])
    out["dilate_2"] = dilate_edge(raw.clone(), 2)
    img = torch.stack([synth_image(35, 3, 90, 160), synth_image(36, 3, 90, 160)])
    out["pre_in"] = img
    out["pre_out"] = batch_preprocess(img.clone(), lower_bound=56)
    save("iw3", **out)


def gen_cunet():
    from waifu2x.models.cunet import CUNet
    from nunif.utils.render import tiled_render
    from oracle import cunet as OC
    out = {}
    sd = OC.random_state_dict(201, up=False)
    m = CUNet().eval()
    m.load_state_dict(sd, strict=True)
    x = torch.stack([synth_image(51, 3, 96, 96), synth_image(52, 3, 96, 96)])
    out["x"], out["y"], out["sdsum"] = x, m(x), sd_checksum(sd)
    m2 = CUNet(no_clip=True).eval()
    m2.load_state_dict(sd, strict=True)
    out["y_no_clip"] = m2(x[:1])
    img = synth_image(53, 3, 150, 170)          # 3x3 tiles of 96 (step 40), plain-overwrite stitch
    out["img"] = img
    out["render_t96_b4"] = tiled_render(img, m, tile_size=96, batch_size=4)
    # UpCUNet (scale 2, offset 36): the released models are trained with no_clip=True (cunet.py:127-134)
    from waifu2x.models.cunet import UpCUNet
    sdu = OC.random_state_dict(203, up=True)
    mu = UpCUNet(no_clip=True).eval()
    mu.load_state_dict(sdu, strict=True)
    out["up_y"], out["up_sdsum"] = mu(x), sd_checksum(sdu)
    muc = UpCUNet().eval()
    muc.load_state_dict(sdu, strict=True)
    out["up_y_clip"] = muc(x[:1])
    imgu = synth_image(54, 3, 100, 130)         # 3x4 tiles of 64 (step 28), plain-overwrite stitch at scale 2
    out["up_img"] = imgu
    out["up_render_t64_b5"] = tiled_render(imgu, mu, tile_size=64, batch_size=5)
    save("cunet", **out)


def gen_row_flow():
    """sbs.row_flow_v3 (iw3/models/row_flow_v3.py) + apply_divergence_nn_LR (iw3/backward_warp.py) on the reference."""
    from iw3.models.row_flow_v3 import RowFlowV3
    from iw3 import backward_warp as RB
    from oracle import row_flow_v3 as ORF
    from oracle.forward_warp import synth_depth
    out = {}
    sd = ORF.random_state_dict(301)
    m = RowFlowV3().eval()
    m.load_state_dict(sd, strict=True)
    m.delta_output = True
    depth = synth_depth(3, 2, 58, 104, "smooth_edges")          # pads to 60 x 192 -> 60 x 24 tokens
    x = ORF.make_input(depth, 2.0, 0.5, 104)
    out["depth"], out["sdsum"] = depth, sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    out["delta"] = m(x)[:, :1]
    c = torch.stack([synth_image(71, 3, 116, 208), synth_image(72, 3, 116, 208)])
    out["c"] = c
    out["left"], out["right"] = RB.apply_divergence_nn_LR(m, c, depth, 2.0, 0.5, steps=1, synthetic_view="both",
                                                          enable_amp=False)
    _, out["right_only"] = RB.apply_divergence_nn_LR(m, c[:1], depth[:1], 2.0, 0.5, steps=1, synthetic_view="right",
                                                     enable_amp=False)
    lb, rb = RB.apply_divergence_nn_LR(m, c[:1], depth[:1], 2.5, 0.4, steps=1, synthetic_view="both",
                                       preserve_screen_border=True, enable_amp=False)
    out["left_border"], out["right_border"] = lb, rb
    # same resolution for image and depth (no grid resize)
    out["left_same"], out["right_same"] = RB.apply_divergence_nn_LR(m, c[:1, :, :58, :104].contiguous(), depth[:1], 2.0, 0.5,
                                                                    steps=1, synthetic_view="both", enable_amp=False)
    save("row_flow", **out)


def gen_row_flow_steps():
    """apply_divergence_nn_LR with warp_steps 2 and 3 (iw3/backward_warp.py:190-231): the net runs on the depth warped by
    the previous steps' flows, the image is warped by the flows one after the other.  divergence 6 / 9 are what
    calc_auto_warp_steps (iw3/utils.py:2179-2186) maps to 2 / 3 steps for row_flow_v3."""
    from iw3.models.row_flow_v3 import RowFlowV3
    from iw3 import backward_warp as RB
    import av
    av.__version__ = "14.2.0"                      # the inert stub's version string does not parse (nunif/utils/video.py)
    from iw3.utils import calc_auto_warp_steps
    from oracle import row_flow_v3 as ORF
    from oracle.forward_warp import synth_depth
    out = {}
    sd = ORF.random_state_dict(301)
    m = RowFlowV3().eval()
    m.load_state_dict(sd, strict=True)
    m.delta_output = True
    depth = synth_depth(3, 2, 58, 104, "smooth_edges")
    c = torch.stack([synth_image(71, 3, 116, 208), synth_image(72, 3, 116, 208)])
    out["depth"], out["c"] = depth, c
    out["sdsum"] = sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    for steps, div in ((2, 6.0), (3, 9.0)):
        assert calc_auto_warp_steps("row_flow_v3", div, "both") == steps
        le, ri = RB.apply_divergence_nn_LR(m, c, depth, div, 0.5, steps=steps, synthetic_view="both", enable_amp=False)
        out[f"left_s{steps}"], out[f"right_s{steps}"] = le, ri
    _, out["right_only_s2"] = RB.apply_divergence_nn_LR(m, c[:1], depth[:1], 3.0, 0.4, steps=2, synthetic_view="right",
                                                        preserve_screen_border=True, enable_amp=False)
    out["auto_steps"] = torch.tensor([[d, calc_auto_warp_steps("row_flow_v3", d, v) or 0]
                                      for d in (1.0, 2.5, 5.0, 5.1, 6.0, 8.0, 8.1, 12.0) for v in ("both",)] +
                                     [[d, calc_auto_warp_steps("row_flow_v3", d, "right") or 0] for d in (2.5, 2.6, 4.0, 4.1)])
    save("row_flow_steps", **out)


def gen_row_flow_sym():
    """The symmetric use of sbs.row_flow_v3 (``row_flow_v3_sym``: model.symmetric = True, apply_divergence_nn_symmetric)."""
    from iw3.models.row_flow_v3 import RowFlowV3
    from iw3 import backward_warp as RB
    from oracle import row_flow_v3 as ORF
    from oracle.forward_warp import synth_depth
    out = {}
    sd = ORF.random_state_dict(311)
    m = RowFlowV3().eval()
    m.load_state_dict(sd, strict=True)
    m.delta_output, m.symmetric = True, True
    depth = synth_depth(5, 1, 58, 104, "smooth_edges")
    c = synth_image(77, 3, 116, 208)[None]
    out["depth"], out["c"] = depth, c
    out["left"], out["right"] = RB.apply_divergence_nn_LR(m, c, depth, 2.0, 0.5, steps=1, synthetic_view="both", enable_amp=False)
    _, out["right_only"] = RB.apply_divergence_nn_LR(m, c, depth, 2.0, 0.5, steps=1, synthetic_view="right", enable_amp=False)
    out["left_only"], _ = RB.apply_divergence_nn_LR(m, c, depth, 2.0, 0.5, steps=1, synthetic_view="left", enable_amp=False)
    save("row_flow_sym", **out)


def gen_mlbw():
    """sbs.mlbw (iw3/models/mlbw.py) + apply_divergence_nn_delta_weight on the reference: l2, l4 and the small l2s."""
    from iw3.models.mlbw import MLBW
    from iw3 import backward_warp as RB
    from oracle import mlbw as OM, row_flow_v3 as ORF
    from oracle.forward_warp import synth_depth
    out = {}
    depth = synth_depth(4, 2, 58, 104, "smooth_edges")          # pads to 60 x 128 -> 60 x 16 tokens
    c = torch.stack([synth_image(74, 3, 116, 208), synth_image(75, 3, 116, 208)])
    out["depth"], out["c"] = depth, c
    for tag, L, small in (("l2", 2, False), ("l4", 4, False), ("l2s", 2, True)):
        sd = OM.random_state_dict(400 + L + (10 if small else 0), L, small)
        m = MLBW(num_layers=L, base_dim=32, small=small).eval()
        m.load_state_dict(sd, strict=True)
        m.delta_output = True
        d, w = m(ORF.make_input(depth[:1], 2.0, 0.5, 104))
        out[tag + "_delta"], out[tag + "_weight"] = d, w
        out[tag + "_sdsum"] = sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})
        nb = 2 if tag == "l2" else 1
        out[tag + "_left"], out[tag + "_right"] = RB.apply_divergence_nn_LR(m, c[:nb], depth[:nb], 2.0, 0.5, steps=1,
                                                                            synthetic_view="both", enable_amp=False)
    save("mlbw", **out)


def gen_convstack():
    """waifu2x.vgg_7 / waifu2x.upconv_7 on the reference: one tile forward + a small tiled_render each."""
    from waifu2x.models.vgg_7 import VGG7
    from waifu2x.models.upconv_7 import UpConv7
    from nunif.utils.render import tiled_render
    from oracle import cunet as OC
    out = {}
    x = torch.stack([synth_image(101, 3, 64, 64), synth_image(102, 3, 64, 64)])
    frame = synth_image(103, 3, 90, 130)
    out["x"], out["frame"] = x, frame
    for tag, cls, seed in (("vgg_7", VGG7, 601), ("upconv_7", UpConv7, 602)):
        sd = OC.conv_stack_state_dict(seed, tag)
        m = cls().eval()
        m.load_state_dict(sd, strict=True)
        out[tag + "_sdsum"] = sd_checksum(sd)
        out[tag + "_z"] = m(x)
        out[tag + "_render"] = tiled_render(frame, m, tile_size=64, batch_size=4)
        print(tag, tuple(out[tag + "_z"].shape), tuple(out[tag + "_render"].shape), float(out[tag + "_z"].mean()),
              float(out[tag + "_z"].std()), float((out[tag + "_z"] <= 0).float().mean()), float((out[tag + "_z"] >= 1).float().mean()))
    save("convstack", **out)


def gen_hole_mask():
    """sbs.mask_mlbw_l2 (MLBW(hole_mask=True)) + return_mask / hole fill / postprocess_hole_mask / nonwarp_mask on the reference."""
    from iw3.models.mlbw import MLBW
    from iw3 import backward_warp as RB
    from oracle import mlbw as OM, row_flow_v3 as ORF
    from oracle.forward_warp import synth_depth
    out = {}
    depth = synth_depth(9, 2, 58, 104, "smooth_edges")
    c = torch.stack([synth_image(84, 3, 116, 208), synth_image(85, 3, 116, 208)])
    out["depth"], out["c"] = depth, c
    sd = OM.random_state_dict(431, 2, False, hole_mask=True)
    m = MLBW(num_layers=2, base_dim=32, hole_mask=True).eval()
    m.load_state_dict(sd, strict=True)
    m.delta_output = True
    out["sdsum"] = sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    d, w, lg = m(ORF.make_input(depth[:1], 2.0, 0.5, 104))
    out["delta"], out["weight"], out["logits"] = d, w, lg
    for tag, shift in (("l", -1), ("r", 1)):
        z, lgs = RB.apply_divergence_nn_delta_weight(m, c, depth, 2.0, 0.5, steps=1, shift=shift, enable_amp=False,
                                                     return_mask=True)
        out["z_" + tag], out["logits_" + tag] = z, lgs
        out["fill_" + tag] = RB.apply_divergence_nn_delta_weight(m, c, depth, 2.0, 0.5, steps=1, shift=shift, enable_amp=False)
    lg_r = out["logits_r"]
    out["mask_same"] = RB.postprocess_hole_mask(lg_r, (58, 104), 0.15)
    out["mask_up"] = RB.postprocess_hole_mask(lg_r, (116, 208), 0.15)
    out["mask_dil"] = RB.postprocess_hole_mask(lg_r, (116, 208), 0.15, inner_dilation=1, outer_dilation=2)
    out["mask_odd"] = RB.postprocess_hole_mask(lg_r, (131, 259), 0.3, inner_dilation=0, outer_dilation=1)
    _, out["nonwarp"] = RB.nonwarp_mask(m, c, depth, 4.0, 0.5, threshold=0.15, inner_dilation=1, outer_dilation=1)
    for k in ("mask_same", "mask_up", "mask_dil", "mask_odd", "nonwarp"):
        print(k, float(out[k].float().mean()))
        out[k] = np.packbits(out[k].numpy().astype(np.uint8), axis=None)
    for k in ("z_l", "z_r", "fill_l", "fill_r"):
        out[k] = out[k].half()
    save("hole_mask", **out)


from make_golden_cases import FORMAT_CASES  # noqa: E402


def format_args(**kw):
    import argparse
    base = dict(ipd_offset=0, rgbd=False, half_rgbd=False, pad=None, pad_mode="tblr", vr180=False, half_sbs=False, half_tb=False,
                anaglyph=None, tb=False, cross_eyed=False, max_output_height=None, max_output_width=None,
                keep_aspect_ratio=False)
    base.update(kw)
    return argparse.Namespace(**base)


def gen_formats():
    """iw3 output formats on the reference: postprocess_image with padding modes, VR180, anaglyph, RGBD (iw3/utils.py)."""
    import av
    av.__version__ = "14.2.0"          # the import stub's version string does not parse; iw3.utils only compares it
    import iw3.utils as U
    from oracle.forward_warp import synth_depth

    class TFShim:
        """torchvision.transforms.functional is not installed: its two tensor functions postprocess_image uses, restated
        (pad: F.pad with (left, top, right, bottom) order; resize: F.interpolate(bicubic, antialias) — what torchvision
        0.22 does for tensors)."""
        @staticmethod
        def pad(img, padding, padding_mode="constant"):
            l, t, r, b = padding
            return torch.nn.functional.pad(img, (l, r, t, b), mode=padding_mode)

        @staticmethod
        def resize(img, size, interpolation=None, antialias=True):
            return torch.nn.functional.interpolate(img[None], size=tuple(size), mode="bicubic", align_corners=False,
                                                   antialias=antialias)[0]
    U.TF = TFShim
    out = {}
    left, right = synth_image(91, 3, 54, 100), synth_image(92, 3, 54, 100)
    out["left"], out["right"] = left, right
    for name, kw in FORMAT_CASES.items():
        out[name] = U.postprocess_image(left.clone(), right.clone(), format_args(**kw))
        print(name, tuple(out[name].shape))
    depth = synth_depth(12, 1, 30, 52, "smooth_edges")[0]
    out["depth"] = depth
    for name, kw in (("rgbd", dict(rgbd=True)), ("half_rgbd", dict(half_rgbd=True, ipd_offset=3.0))):
        le, re = U.apply_rgbd(left.clone(), depth.clone(), mapper="pow2")
        out[name] = U.postprocess_image(le, re, format_args(**kw))
    save("formats", **out)


def frame_pool_args(**kw):
    """Reference argparse names read by bind_batch_frame_callback / apply_divergence / postprocess_image."""
    import argparse
    base = dict(batch_size=2, tta=False, low_vram=False, disable_amp=True, edge_dilation=0, depth_aa=False, rgbd=False,
                half_rgbd=False, method="grid_sample", mapper="none", divergence=2.0, convergence=0.5, synthetic_view="both",
                cuda_stream=False, pix_fmt="yuv420p", rotate_left=False, rotate_right=False, debug_depth=False,
                state={"device": torch.device("cpu"), "convergence_model": None})
    base.update(vars(format_args()))
    base.update(kw)
    return argparse.Namespace(**base)


from make_golden_cases import (FRAME_CALLBACK_CASES, FRAME_POOL_CASES, FakeSide, FakeWindowDepth, fake_depth_net,  # noqa: E402
                                frame_pool_frames)


def gen_frame_pool():
    """The reference's frame scheduler on CPU: VU.FrameCallbackPool driving iw3.utils.bind_batch_frame_callback (batching,
    EMA min-max look-ahead, flush at scene cuts, ordered output) around a fake depth net and the real grid_sample warp."""
    import av
    av.__version__ = "14.2.0"
    import iw3.utils as U
    import nunif.utils.video as VU
    from iw3.base_depth_model import BaseDepthModel

    class Frame:
        def __init__(self, x, pts):
            self.x, self.pts = x, pts

    VU.to_tensor = lambda frame, device=None: frame.x.to(device) if device is not None else frame.x
    VU.to_frame = lambda x, use_16bit=False: x

    class FakeDepth(BaseDepthModel):
        def load_model(self, model_type, resolution, device):
            return None

        def is_metric(self):
            return False

        @classmethod
        def supported(cls, name):
            return True

        @classmethod
        def get_name(cls):
            return "fake"

        @classmethod
        def multi_gpu_supported(cls, name):
            return True

        @classmethod
        def force_update(cls):
            pass

        @classmethod
        def get_model_path(cls, name):
            return None

        @classmethod
        def has_checkpoint_file(cls, name):
            return True

        def infer(self, x, **kw):
            return fake_depth_net(x) if x.ndim == 4 else fake_depth_net(x[None])[0]

    out = {}
    # the single-frame route (:618-709) and the windowed-depth route (:834-926); frames arrive as HWC uint8 ndarrays
    VU.to_ndarray = lambda frame: (frame.x * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().numpy()
    for name, (n, bs, cuts, ema) in FRAME_CALLBACK_CASES.items():
        args = frame_pool_args(batch_size=bs)
        if name == "single":
            dm = FakeDepth("fake")
            dm.enable_ema(ema[0], buffer_size=ema[1])
            cb = U.bind_single_frame_callback(dm, FakeSide(), set(cuts), args)
        else:
            cb = U.bind_vda_frame_callback(FakeWindowDepth(), FakeSide(), set(cuts), args)
        counts, frames = [], []
        for i, x in enumerate(frame_pool_frames(n)):
            r = cb(Frame(x, i)) or []
            counts.append(len(r))
            frames += r
        r = cb(None)
        counts.append(len(r))
        frames += r
        out[name + "_counts"] = torch.tensor(counts)
        out[name + "_frames"] = torch.stack(frames)
        print(name, counts, len(frames))
    for name, (n, bs, cuts, ema, workers) in FRAME_POOL_CASES.items():
        dm = FakeDepth("fake")
        if ema is not None:
            dm.enable_ema(ema[0], buffer_size=ema[1])
        args = frame_pool_args(batch_size=bs)
        cb, pre = U.bind_batch_frame_callback(dm, None, set(cuts), args)
        pool = VU.FrameCallbackPool(frame_callback=cb, preprocess_callback=pre, batch_size=bs, device=[torch.device("cpu")],
                                    max_workers=workers, max_batch_queue=workers + 1, require_pts=True, require_flush=True)
        counts, frames = [], []
        for i, x in enumerate(frame_pool_frames(n)):
            r = pool(Frame(x, i)) or []
            counts.append(len(r))
            frames += r
        r = pool(None)
        counts.append(len(r))
        frames += r
        pool.shutdown()
        assert len(frames) == n, (name, len(frames))
        out[name + "_counts"] = torch.tensor(counts)
        if name == "ema_threads":          # worker threads change WHEN frames are handed back, never the frames
            assert torch.equal(torch.stack(frames), out["ema_frames"])
            continue
        out[name + "_frames"] = torch.stack(frames)
        print(name, counts)
    save("frame_pool", **out)


def gen_depth_aa():
    """iw3.depth_aa (iw3/models/depth_aa.py) on the reference: forward (clamped / unclamped) and infer()."""
    from iw3.models.depth_aa import DepthAA
    from oracle import depth_aa as ODA
    from oracle.forward_warp import synth_depth
    out = {}
    sd = ODA.random_state_dict(501)
    m = DepthAA().eval()
    m.load_state_dict(sd, strict=True)
    x = synth_depth(7, 2, 70, 100, "smooth_edges")             # pads to 80 x 112 -> 40 x 56 tokens
    out["x"], out["sdsum"] = x, sd_checksum({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    out["y"] = m(x)
    out["y_noclamp"] = m(x, clamp=False)
    xi = x[:1] * 7.0 + 3.0                                      # un-normalised (metric-like) depth
    out["xi"], out["y_infer"] = xi, m.infer(xi)
    save("depth_aa", **out)


def hot_image(seed, h, w):
    """synth_image with SATURATED FLATS: blocks of exact 0 / 1 (clipped highlights, crushed blacks of real pictures).
    Must stay identical to tests/conftest.py::hot_image."""
    x = synth_image(seed, 3, h, w)
    x[:, : h // 3, : w // 2] = 1.0
    x[:, h // 2:, w // 3: 2 * w // 3] = 0.0
    x[1, h // 4: h // 2, w // 2:] = 1.0
    return x


# (tag, class name, scale factor, seed): seeds picked so that the emulated-fp16 reference sits between 45 and 50 dB (a hard but
# meaningful regime) plus one chaotic case (2x, seed 422: 29 dB, 37 % of the picture clamped)
HOT_SWIN = (("2x", "SwinUNet2x", 2, 432), ("2x_chaos", "SwinUNet2x", 2, 422), ("4x", "SwinUNet4x", 4, 404), ("1x", "SwinUNet", 1, 411))


def gen_hot():
    """Trained-regime stress: ``regime="hot"`` weights (nunif_amd/synthetic.py), inputs with saturated flats.
    Per case the fixture holds the REFERENCE's fp32 output (``y_ref``; for the external depth net: the HuggingFace-pinned
    restatement's) and the output of the same arithmetic under ``oracle.fp16_emulation`` (``y_emu``: every op result rounded to
    fp16 = the reference's own CUDA autocast mode, emulated).  tests/test_gpu_hot_regime.py holds the HIP engine to
    PSNR(hip, y_ref) >= PSNR(y_emu, y_ref) - 1 dB."""
    import waifu2x.models.swin_unet as RS
    from waifu2x.models.cunet import CUNet
    from oracle import swin_unet as O, cunet as OC, depth_anything_v2 as OD
    from oracle.fp16_emulation import fp16_autocast_emulation, half_weights
    out = {}
    x = torch.stack([hot_image(21, 64, 64), hot_image(22, 64, 64)])
    out["swin_x"] = x
    for tag, cls, sf, seed in HOT_SWIN:
        sd = O.random_state_dict(seed, sf, regime="hot")
        m = getattr(RS, cls)().eval()
        m.load_state_dict(sd, strict=True)
        y_ref = m(x)
        y_or = torch.clamp(O.unet_forward(sd, x, sf), 0, 1)
        # the oracle IS the reference here too (fp32 op order differs; the chaotic case amplifies that to 7e-3 max-abs)
        assert torch.mean((y_ref - y_or) ** 2).item() < 1e-7 and (y_ref - y_or).abs().max() < 2e-2, (tag, (y_ref - y_or).abs().max())
        with fp16_autocast_emulation():
            y_emu = torch.clamp(O.unet_forward(half_weights(sd), x, sf), 0, 1)
        out[f"swin_{tag}_ref"], out[f"swin_{tag}_emu"], out[f"swin_{tag}_sdsum"] = y_ref, y_emu, sd_checksum(sd)
        mse = torch.mean((y_emu.double() - y_ref.double()) ** 2).item()
        print(f"swin {tag}: emulated fp16 vs fp32 {10 * np.log10(1 / (mse + 1e-6)):.2f} dB, clamped {float(((y_ref == 0) | (y_ref == 1)).float().mean()):.3f}")
    xc = torch.stack([hot_image(51, 96, 96), hot_image(52, 96, 96)])
    sd = OC.random_state_dict(601, up=False, regime="hot")
    m = CUNet().eval()
    m.load_state_dict(sd, strict=True)
    with fp16_autocast_emulation():
        y_emu = OC.model_forward(half_weights(sd), xc)
    out["cunet_x"], out["cunet_ref"], out["cunet_emu"], out["cunet_sdsum"] = xc, m(xc), y_emu, sd_checksum(sd)
    g = torch.Generator().manual_seed(5)
    xd = torch.randn(2, 3, 56, 70, generator=g)
    sd = OD.random_state_dict(301, grid=37, regime="hot")
    with fp16_autocast_emulation():
        y_emu = OD.model_forward(half_weights(sd), xd)
    out["depth_x"], out["depth_ref"], out["depth_emu"] = xd, OD.model_forward(sd, xd), y_emu
    out["depth_sdsum"] = sd_checksum(sd)
    save("hot_regime", **out)


GROUPS = {"hot": gen_hot, "seam": gen_seam, "swin": gen_swin, "iw3": gen_iw3, "cunet": gen_cunet, "row_flow": gen_row_flow,
          "mlbw": gen_mlbw, "depth_aa": gen_depth_aa, "hole_mask": gen_hole_mask, "formats": gen_formats, "convstack": gen_convstack, "row_flow_sym": gen_row_flow_sym, "row_flow_steps": gen_row_flow_steps, "swin8x": gen_swin8x, "swin4xl": gen_swin4xl, "morph": gen_morph, "vda": gen_vda, "vda_online": gen_vda_online, "light_inpaint": gen_light_inpaint, "light_video_inpaint": gen_light_video_inpaint, "light_video_inpaint_ml": gen_light_video_inpaint_ml,
          "frame_pool": gen_frame_pool, "forward_inpaint": gen_forward_inpaint, "swin_v2": gen_swin_v2}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GROUPS)
    for n in names:
        GROUPS[n]()
