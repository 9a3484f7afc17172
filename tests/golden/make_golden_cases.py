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


# ---- frame scheduler (gen_frame_pool / tests/test_frame_pipeline.py) -------------------------------------------------
import torch  # noqa: E402


def _synth_image(seed, c, h, w):
    """Same generator as make_golden.synth_image / conftest.synth_image."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, c, max(2, h // 16 + 1), max(2, w // 16 + 1), generator=g)
    up = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)[0]
    return torch.clamp(up * 0.8 + 0.2 * torch.rand(c, h, w, generator=g), 0, 1)


def fake_depth_net(x):
    """Deterministic stand-in for a depth net: BCHW in [0,1] -> B1HW whose range differs from frame to frame."""
    w = torch.tensor([0.5, 0.3, 0.2]).view(1, 3, 1, 1)
    d = (x * w).sum(dim=1, keepdim=True)
    gain = 0.5 + x.mean(dim=(1, 2, 3), keepdim=True)
    return d * gain + 0.1 * gain


FRAME_POOL_CASES = {
    # name: (n_frames, batch_size, scene-cut pts, (ema decay, buffer) or None, max_workers)
    "plain": (11, 2, (5,), None, 0),
    "ema": (13, 3, (6,), (0.75, 4), 0),
    "ema_threads": (13, 3, (6,), (0.75, 4), 2),
    "cut_last_of_batch": (8, 2, (3, 7), (0.5, 2), 0),
}


def frame_pool_frames(n, h=24, w=40):
    """uint8-derived frames (what a decoder hands over): CHW float = u8 / 255."""
    return [(_synth_image(700 + i, 3, h, w) * (0.6 + 0.04 * i) * 255).round().clamp(0, 255) / 255 for i in range(n)]




class FakeSide:
    """Side model with a temporal queue as far as the schedulers care: ``flush`` hands back one more stereo pair."""

    def flush(self, enable_amp=True):
        return torch.full((1, 3, 24, 40), 0.25), torch.full((1, 3, 24, 40), 0.75)


class FakeWindowDepth:
    """Duck-typed stand-in for ``VideoDepthAnythingModel`` on the ``bind_vda_frame_callback`` route: outputs lag the inputs by
    two frames, every frame is normalised by its own range, a scene cut / the final flush drains the window."""

    def __init__(self):
        self.buf = []

    def reset(self):
        self.buf = []

    @staticmethod
    def _emit(d):
        return (d - d.amin()) / (d.amax() - d.amin())

    def infer_with_normalize(self, x, pts, reset_pts, **kw):
        out = []
        for i in range(x.shape[0]):
            self.buf.append(fake_depth_net(x[i:i + 1].cpu())[0].to(x.device))
            if len(self.buf) > 2:
                out.append(self._emit(self.buf.pop(0)))
            if pts[i] in reset_pts:
                out += [self._emit(d) for d in self.buf]
                self.buf = []
        return out

    def flush_with_normalize(self, **kw):
        out = [self._emit(d) for d in self.buf]
        self.buf = []
        return out


# route -> (n_frames, batch_size, scene-cut pts, (ema decay, buffer) or None)
FRAME_CALLBACK_CASES = {"single": (7, 1, (3,), (0.75, 3)), "vda": (11, 3, (4,), None)}
