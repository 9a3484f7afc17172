"""The Video-Depth-Anything streaming network (config 5's depth net): ``oracle/video_depth_anything_net.py`` (published architecture,
PARITY UNPINNED — no implementation of it exists offline) and the HIP engine's temporal modules against it.

CPU: the oracle's own invariants, and the ENGINE'S ARRANGEMENT of the temporal attention — K0 = Wk h / V0 = Wv h ring caches, the
position code through three [32][C] tables, exp2 softmax with the scale folded into Wq (nunif_amd/csrc/depth_temporal.hip) — restated
in torch and held against the published form (hidden-state cache, K = Wk (h + pe) for the whole window on every frame).
GPU: the engine, frame by frame over more than one window, against the oracle.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import psnr
from oracle import video_depth_anything_net as VN


def clip(seed, n, h, w):
    """n frames of a drifting synthetic scene, ImageNet-normalised range: neighbours are similar, far frames are not."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(3, h + 2 * n, w + 2 * n, generator=g)
    base = F.avg_pool2d(base.unsqueeze(0), 5, stride=1, padding=2)[0] * 3.0
    return torch.stack([base[:, i:i + h, 2 * i:2 * i + w] + 0.05 * torch.randn(3, h, w, generator=g) for i in range(n)])


def test_positional_encoding_is_the_sinusoidal_table():
    pe = VN.positional_encoding(64)
    assert pe.shape == (32, 64) and torch.equal(pe[0, 0::2], torch.zeros(32)) and torch.equal(pe[0, 1::2], torch.ones(32))
    assert pe[5, 6].item() == pytest.approx(math.sin(5 * 10000 ** (-6 / 64)), abs=1e-6)
    assert pe[5, 7].item() == pytest.approx(math.cos(5 * 10000 ** (-6 / 64)), abs=1e-6)


def test_oracle_shapes_and_first_frame():
    sd = VN.random_state_dict(5)
    assert abs(sum(v.numel() for v in sd.values()) / 1e6 - 29.03) < 0.01            # 24.78 M (Depth-Anything-V2-S) + 4.25 M temporal
    frames = clip(1, 4, 56, 84)
    st = VN.new_state()
    outs = torch.cat([VN.infer_video_depth_one(sd, f, st) for f in frames])
    assert outs.shape == (4, 56, 84) and float(outs.min()) >= 0 and float(outs.std()) > 0.1 and st["frames"] == 4
    assert all(c.shape[0] == 4 for mod in st["caches"] for c in mod)
    # an empty cache and a one-frame clip are the same computation
    assert torch.equal(VN.forward_window(sd, frames[:1]), outs[:1])
    # the temporal path is live: frame 3 with its history differs from frame 3 alone, and from the offline window's frame 3
    alone = VN.infer_video_depth_one(sd, frames[3], VN.new_state())
    off = VN.forward_window(sd, frames)
    assert float((alone[0] - outs[3]).abs().max()) > 0.05 * float(outs[3].std())
    assert off.shape == outs.shape and float((off[3] - outs[3]).abs().max()) > 1e-3
    # causal: what frame 1 produced does not depend on frames 2, 3
    st2 = VN.new_state()
    two = torch.cat([VN.infer_video_depth_one(sd, f, st2) for f in frames[:2]])
    assert torch.equal(two, outs[:2])


class EngineArrangement:
    """depth_temporal.hip / depth_anything.hip run_tmod for ONE temporal attention block, in torch fp32: what is cached, how the
    window slides, where the position code enters."""

    def __init__(self, sd, ab, C):
        self.C, self.hd = C, C // 8
        qs = self.hd ** -0.5 * 1.4426950408889634
        self.w = torch.cat([sd[ab + "to_q.weight"] * qs, sd[ab + "to_k.weight"], sd[ab + "to_v.weight"]])
        pe = VN.positional_encoding(C).double()
        self.pq = (pe @ sd[ab + "to_q.weight"].double().t() * qs).float()
        self.pk = (pe @ sd[ab + "to_k.weight"].double().t()).float()
        self.pv = (pe @ sd[ab + "to_v.weight"].double().t()).float()
        self.kc = self.vc = None
        self.start, self.len = 0, 0

    def step(self, n):                                  # n: [P, C] LayerNorm'ed hidden states of the new frame -> att [P, C]
        P, C, hd = n.shape[0], self.C, self.hd
        if self.kc is None:
            self.kc, self.vc = torch.full((32, P, C), float("nan")), torch.full((32, P, C), float("nan"))
        qkv = n @ self.w.t()
        idx, start = self.len, self.start
        cur = (start + idx) & 31
        self.kc[cur], self.vc[cur] = qkv[:, C:2 * C], qkv[:, 2 * C:]
        slots = [(start + j) & 31 for j in range(idx + 1)]
        q = (qkv[:, :C] + self.pq[idx]).reshape(P, 8, 1, hd)
        k = (self.kc[slots] + self.pk[:idx + 1, None, :]).reshape(idx + 1, P, 8, hd).permute(1, 2, 0, 3)
        v = (self.vc[slots] + self.pv[:idx + 1, None, :]).reshape(idx + 1, P, 8, hd).permute(1, 2, 0, 3)
        s = (q @ k.transpose(-2, -1)).squeeze(2)                          # [P, 8, idx + 1], log2 units
        p = torch.exp2(s - s.max(dim=-1, keepdim=True).values)
        att = (p.unsqueeze(2) @ v).squeeze(2) / p.sum(dim=-1, keepdim=True)
        if self.len + 1 > 31:
            self.start = (self.start + 1) & 31
        else:
            self.len += 1
        return att.reshape(P, C)


def test_engine_arrangement_equals_the_published_form_across_window_slides():
    """40 frames through one attention block: the published form (cache of hidden states, the position code added to the whole window
    and projected on every frame) against the engine's (K0 / V0 ring, position tables)."""
    C, P = 64, 7
    sd = VN.random_state_dict(11)
    b = "head.motion_modules.2.temporal_transformer.transformer_blocks.0."
    ab = b + "attention_blocks.0."
    eng = EngineArrangement(sd, ab, C)
    g = torch.Generator().manual_seed(3)
    cache = None
    pe = VN.positional_encoding(C)
    worst = 0.0
    for t in range(40):
        n = torch.randn(P, C, generator=g)
        seq = n[None] if cache is None else torch.cat([cache, n[None]])
        cache = seq[-31:]
        L = seq.shape[0]
        assert L == min(t + 1, 32)
        kv = seq + pe[:L, None, :]
        q = F.linear(kv[-1:], sd[ab + "to_q.weight"]).reshape(1, P, 8, C // 8).permute(1, 2, 0, 3)
        k = F.linear(kv, sd[ab + "to_k.weight"]).reshape(L, P, 8, C // 8).permute(1, 2, 0, 3)
        v = F.linear(kv, sd[ab + "to_v.weight"]).reshape(L, P, 8, C // 8).permute(1, 2, 0, 3)
        ref = (torch.softmax(q * (C // 8) ** -0.5 @ k.transpose(-2, -1), dim=-1) @ v).permute(2, 0, 1, 3).reshape(P, C)
        got = eng.step(n)
        worst = max(worst, float((got - ref).abs().max()))
    assert worst < 2e-5, worst
    assert eng.len == 31 and eng.start == 9 and not torch.isnan(eng.kc).any()


def test_streaming_model_names_resolve_to_the_published_checkpoint_files(tmp_path):
    """iw3/video_depth_anything_streaming_model.py:20-27 ``MODEL_FILES``: the same file names, under ``<model_dir>/checkpoints``."""
    from nunif_amd.iw3.video_depth_anything_streaming_model import MODEL_FILE_NAMES, VideoDepthAnythingStreamingModel as M
    assert MODEL_FILE_NAMES["VDA_Stream_S"] == "video_depth_anything_vits.pth"
    assert MODEL_FILE_NAMES["VDA_Stream_Metric_L"] == "metric_video_depth_anything_vitl.pth" and set(MODEL_FILE_NAMES) == {
        f"VDA_Stream_{m}{s}" for m in ("", "Metric_") for s in "SBL"}
    assert M.get_model_path("VDA_Stream_B").endswith("checkpoints/video_depth_anything_vitb.pth")
    assert M._path("VDA_Stream_S", str(tmp_path)) == str(tmp_path / "checkpoints" / "video_depth_anything_vits.pth")
    assert not M.has_checkpoint_file("VDA_Stream_S") or True           # (a developer box may hold one)


@pytest.fixture(scope="module")
def vda_net(hiplib):
    from nunif_amd.iw3.video_depth_anything_net import HipVideoDepthAnythingStreaming
    sd = VN.random_state_dict(5)
    return sd, HipVideoDepthAnythingStreaming(sd)


@pytest.mark.gpu
def test_hip_streaming_network_vs_oracle_over_two_windows(vda_net):
    """36 frames: the window fills (frames 0-31) and slides (32-35).  Per frame: PSNR >= 50 dB on the frame's own range and relative
    rms < 1e-2 against the fp32 oracle — the criterion of the per-frame Depth-Anything engine (tests/test_depth_anything.py)."""
    sd, net = vda_net
    frames = clip(2, 36, 126, 154)
    net.reset_state()
    st = VN.new_state()
    outs, worst_p, worst_r = [], 1e9, 0.0
    for i, f in enumerate(frames):
        ref = VN.infer_video_depth_one(sd, f, st)
        y = net.infer_video_depth_one(f.to("cuda:0")).cpu()
        assert y.shape == ref.shape == (1, 126, 154)
        span = float(ref.max() - ref.min())
        p, rel = psnr(y / span, ref / span), float(((y - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
        worst_p, worst_r = min(worst_p, p), max(worst_r, rel)
        assert float(ref.std()) > 1e-3 and p >= 50.0 and rel < 1e-2, (i, p, rel)
        outs.append(y)
    print(f"\nVDA streaming, 36 frames: worst PSNR {worst_p:.2f} dB, worst rel. rms {worst_r:.2e}")
    # reset_state starts the same stream again: the same bits
    net.reset_state()
    again = [net.infer_video_depth_one(f.to("cuda:0")).cpu() for f in frames[:3]]
    assert all(torch.equal(a, b) for a, b in zip(again, outs))
    # ... and the history matters: frame 2 alone is not frame 2 after frames 0, 1
    net.reset_state()
    alone = net.infer_video_depth_one(frames[2].to("cuda:0")).cpu()
    assert float((alone - outs[2]).abs().max()) > 0.05 * float(outs[2].std())


@pytest.mark.gpu
def test_the_two_temporal_attention_kernels_agree(vda_net, monkeypatch):
    """``vda_tattn2_kernel`` (a thread per pixel and 8-channel chunk, position tables in LDS) against ``vda_tattn_kernel`` (a thread
    per pixel and head; NUNIF_VDA_TATTN=1): the same sums in another order."""
    sd, net = vda_net
    frames = clip(9, 34, 70, 98)
    outs = {}
    for form in ("0", "1"):
        monkeypatch.setenv("NUNIF_VDA_TATTN", form)
        net.reset_state()
        outs[form] = torch.stack([net.infer_video_depth_one(f.to("cuda:0")).cpu() for f in frames])
    span = float(outs["0"].max() - outs["0"].min())
    assert float(outs["0"].std()) > 1e-3 and float((outs["0"] - outs["1"]).abs().max()) < 2e-3 * span, float((outs["0"] - outs["1"]).abs().max()) / span


@pytest.mark.gpu
def test_hip_streaming_network_resolution_change(vda_net):
    sd, net = vda_net
    net.reset_state()
    for f in clip(4, 3, 126, 154):
        net.infer_video_depth_one(f.to("cuda:0"))
    # another resolution starts a new window (the caches are laid out per pixel)
    f2 = clip(6, 1, 70, 98)[0]
    y = net.infer_video_depth_one(f2.to("cuda:0")).cpu()
    ref = VN.infer_video_depth_one(sd, f2, VN.new_state())
    span = float(ref.max() - ref.min())
    assert psnr(y / span, ref / span) >= 50.0


@pytest.mark.gpu
def test_batches_of_consecutive_frames_equal_the_per_frame_stream(vda_net):
    """``infer_video_depth_batch``: 35 frames as batches of 3 / 1 / 4 / ... (the window fills and slides INSIDE batches) against the
    same frames one per call — the same arithmetic per token (a GEMM may tile another way at another M: a tolerance, not bits) — and
    against the oracle."""
    sd, net = vda_net
    frames = clip(12, 35, 70, 98)
    net.reset_state()
    one = torch.cat([net.infer_video_depth_one(f.to("cuda:0")).cpu() for f in frames])
    net.reset_state()
    outs, i = [], 0
    for n in (3, 1, 4, 3, 3, 5, 3, 3, 2, 3, 3, 2):
        outs.append(net.infer_video_depth_batch(frames[i:i + n].to("cuda:0")).cpu())
        i += n
    assert i == 35
    bat = torch.cat(outs)
    span = float(one.max() - one.min())
    assert bat.shape == one.shape and float((bat - one).abs().max()) < 2e-3 * span, float((bat - one).abs().max()) / span
    st = VN.new_state()
    for k, f in enumerate(frames):
        ref = VN.infer_video_depth_one(sd, f, st)[0]
        sp = float(ref.max() - ref.min())
        assert psnr(bat[k] / sp, ref / sp) >= 50.0, k


@pytest.mark.gpu
def test_streaming_model_wrapper_with_the_temporal_network(vda_net, monkeypatch):
    """``VideoDepthAnythingStreamingModel.infer`` (iw3/video_depth_anything_streaming_model.py:77-103) around the temporal network:
    pre-processing, the per-frame loop, post-processing — against the oracle's loop around the oracle's network."""
    from nunif_amd.iw3.video_depth_anything_streaming_model import VideoDepthAnythingStreamingModel
    from oracle import video_depth_anything as OV
    sd, net = vda_net
    model = VideoDepthAnythingStreamingModel("VDA_Stream_S", backbone=net).load(gpu=0, resolution=126)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(3, 3, 90, 160, generator=g)
    x = F.avg_pool2d(x, 3, stride=1, padding=1)
    model.reset_state()
    y = model.infer(x.to("cuda:0"), edge_dilation=0).cpu()
    st = VN.new_state()
    ref = OV.streaming_infer(lambda f: VN.infer_video_depth_one(sd, f, st), x, 126, False, edge_dilation=0)
    assert y.shape == ref.shape
    span = float(ref.max() - ref.min())
    assert psnr(y / span, ref / span) >= 45.0, psnr(y / span, ref / span)
    # NUNIF_VDA_BATCH=0: the reference's per-frame loop around infer_video_depth_one
    monkeypatch.setenv("NUNIF_VDA_BATCH", "0")
    model.reset_state()
    y1 = model.infer(x.to("cuda:0"), edge_dilation=0).cpu()
    assert psnr(y1 / span, y / span) >= 55.0


@pytest.mark.gpu
def test_streaming_model_loads_the_network_from_a_checkpoint_file(vda_net, tmp_path):
    """``VideoDepthAnythingStreamingModel("VDA_Stream_S").load()`` without a backbone: the published checkpoint file from
    ``<model_dir>/checkpoints`` on the engine's streaming network (here: the seeded weights saved under that name)."""
    from nunif_amd.iw3.video_depth_anything_net import HipVideoDepthAnythingStreaming
    from nunif_amd.iw3.video_depth_anything_streaming_model import VideoDepthAnythingStreamingModel
    sd, net = vda_net
    (tmp_path / "checkpoints").mkdir()
    torch.save(sd, tmp_path / "checkpoints" / "video_depth_anything_vits.pth")
    import nunif_amd.iw3.video_depth_anything_streaming_model as SM
    SM._WARNED_UNPINNED = False
    with pytest.warns(RuntimeWarning, match="parity unpinned"):             # a real checkpoint in the unpinned network is announced
        model = VideoDepthAnythingStreamingModel("VDA_Stream_S", model_dir=str(tmp_path)).load(gpu=0, resolution=126)
    assert isinstance(model.model, HipVideoDepthAnythingStreaming) and model.model.prep_lower_bound == 126
    g = torch.Generator().manual_seed(21)
    x = F.avg_pool2d(torch.rand(2, 3, 90, 160, generator=g), 3, stride=1, padding=1)
    y = model.infer(x.to("cuda:0")).cpu()
    net.reset_state()
    from nunif_amd.iw3.video_depth_anything_streaming_model import VideoDepthAnythingStreamingModel as M
    ref = M("VDA_Stream_S", backbone=net).load(gpu=0, resolution=126).infer(x.to("cuda:0")).cpu()
    assert torch.equal(y, ref)                                               # the same weights, the same engine: the same bits
