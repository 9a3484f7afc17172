"""Hot-regime parity statistics: for each swin_unet hot case (the golden fixture's weights) the gap

    gap = PSNR(HIP, fp32 oracle) - PSNR(emulated-fp16 oracle, fp32 oracle)      [dB; negative = HIP noisier than the emulation]

over N input images of 64 x 64 (seeded ``hot_image``) plus one 256 x 256 tile, mean / standard deviation / worst.  The oracle runs
on the host at test time (0.05 s per 64 x 64 image; it IS the reference on these weights: tests/test_hot_regime.py pins that
against the committed reference outputs).  Run on the GPU box:  python tools/hot_regime_stats.py [n_inputs]  -> one JSON line.
``tests/test_hot_regime.py::test_hip_swin_unet_hot_regime_gap_statistics`` asserts on the same function."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

HOT_SWIN = (("2x", 2, 432), ("2x_chaos", 2, 422), ("4x", 4, 404), ("1x", 1, 411))


def _psnr(a, b):
    import math
    return 10 * math.log10(1.0 / (torch.mean((a.double() - b.double()) ** 2).item() + 1e-12))


def case_gaps(tag, sf, seed, n_inputs=8, big=True, device="cuda:0"):
    """-> {"gaps": [...], "emu": [...], "hip": [...], "big": {...} | None} for one hot case."""
    from conftest import hot_image
    from oracle import swin_unet as O
    from oracle.fp16_emulation import fp16_autocast_emulation, half_weights
    from nunif_amd.waifu2x.models import swin_unet as M
    sd = O.random_state_dict(seed, sf, regime="hot")
    sdh = half_weights(sd)
    m = {1: M.SwinUNet, 2: M.SwinUNet2x, 4: M.SwinUNet4x}[sf]().eval()
    m.load_state_dict(sd, strict=True)
    m = m.to(device)

    def one(x):
        ref = torch.clamp(O.unet_forward(sd, x, sf), 0, 1)
        with fp16_autocast_emulation():
            emu = torch.clamp(O.unet_forward(sdh, x, sf), 0, 1)
        y = m(x.to(device)).float().cpu()
        assert y.shape == ref.shape and torch.isfinite(y).all(), tag
        return _psnr(y, ref), _psnr(emu, ref)

    hip, emu = [], []
    for k in range(n_inputs):
        h, e = one(hot_image(21 + 7 * k, 64, 64)[None])
        hip.append(h)
        emu.append(e)
    rec = {"hip": hip, "emu": emu, "gaps": [h - e for h, e in zip(hip, emu)], "big": None}
    if big:
        h, e = one(hot_image(91, 256, 256)[None])
        rec["big"] = {"hip": h, "emu": e, "gap": h - e}
    return rec


def summarize(rec):
    g = torch.tensor(rec["gaps"], dtype=torch.float64)
    out = {"n": len(rec["gaps"]), "mean_gap_db": round(g.mean().item(), 3), "std_gap_db": round(g.std(unbiased=True).item(), 3),
           "worst_gap_db": round(g.min().item(), 3), "best_gap_db": round(g.max().item(), 3),
           "mean_emu_db": round(sum(rec["emu"]) / len(rec["emu"]), 2), "mean_hip_db": round(sum(rec["hip"]) / len(rec["hip"]), 2)}
    if rec["big"]:
        out["tile256"] = {k: round(v, 2) for k, v in rec["big"].items()}
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.set_grad_enabled(False)
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    res = {"lib": os.environ.get("NUNIF_HIP_LIB", "default")}
    for tag, sf, seed in HOT_SWIN:
        res[tag] = summarize(case_gaps(tag, sf, seed, n))
        print(tag, res[tag], file=sys.stderr, flush=True)
    print(json.dumps(res))
