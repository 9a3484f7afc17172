#!/usr/bin/env python3
"""Where does the hot ``2x_chaos`` case lose its 2.7 dB against the emulated fp16 reference?    For the hot
weights of each swin case and N inputs: every stage tap of (a) the HIP engine (``nunif_hip_swin_unet_debug_taps``) and (b) the
fp16-autocast emulation of the oracle, both as relative rms error against the fp32 oracle's tap, plus their ratio.  The first stage
where HIP / emulation exceeds 1.3 is flagged.  Run on the GPU box:
    python tools/hot_taps.py [case ...] > profiles/r06_hot_taps.txt"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
HOT_SWIN = {"2x": (2, 432), "2x_chaos": (2, 422), "4x": (4, 404), "1x": (1, 411)}


def read_taps(engine):
    from nunif_amd import _hip
    lib = _hip.lib()
    taps, i = {}, 0
    while True:
        name = ctypes.create_string_buffer(64)
        nbytes = ctypes.c_int64(0)
        rc = lib.nunif_hip_swin_unet_get_tap(engine.handle, i, name, 64, None, 0, ctypes.byref(nbytes))
        if rc == 1:
            break
        _hip.check(rc)
        buf = np.empty(nbytes.value // 2, dtype=np.float16)
        _hip.check(lib.nunif_hip_swin_unet_get_tap(engine.handle, i, name, 64, buf.ctypes.data_as(ctypes.c_void_p), nbytes.value,
                                                   ctypes.byref(nbytes)))
        taps[name.value.decode()] = torch.from_numpy(buf.astype(np.float32))
        i += 1
    return taps


def rel(a, ref):
    return ((a.double() - ref.double()).pow(2).mean().sqrt() / (ref.double().pow(2).mean().sqrt() + 1e-12)).item()


def main():
    from conftest import hot_image
    from oracle import swin_unet as O
    from oracle.fp16_emulation import fp16_autocast_emulation, half_weights
    from nunif_amd import _hip
    from nunif_amd.waifu2x.models import swin_unet as M
    torch.set_grad_enabled(False)
    lib = _hip.lib()
    n_inputs = int(os.environ.get("HOT_TAPS_INPUTS", "4"))
    for tag in (sys.argv[1:] or ["2x_chaos", "2x"]):
        sf, seed = HOT_SWIN[tag]
        sd = O.random_state_dict(seed, sf, regime="hot")
        sdh = half_weights(sd)
        m = {1: M.SwinUNet, 2: M.SwinUNet2x, 4: M.SwinUNet4x}[sf]().eval()
        m.load_state_dict(sd, strict=True)
        m = m.to("cuda:0")
        acc = {}
        order = []
        for k in range(n_inputs):
            x = hot_image(21 + 7 * k, 64, 64)[None]
            ref_t, emu_t = {}, {}
            O.unet_forward(sd, x, sf, taps=ref_t)
            with fp16_autocast_emulation():
                O.unet_forward(sdh, x, sf, taps=emu_t)
            eng = m.engine()
            _hip.check(lib.nunif_hip_swin_unet_debug_taps(eng.handle, 1))
            m(x.to("cuda:0"))
            hip_t = read_taps(eng)
            _hip.check(lib.nunif_hip_swin_unet_debug_taps(eng.handle, 0))
            for name, r in ref_t.items():
                if name not in hip_t:
                    continue
                if name not in acc:
                    acc[name] = [[], [], []]
                    order.append(name)
                acc[name][0].append(rel(hip_t[name].reshape(r.shape), r))
                acc[name][1].append(rel(emu_t[name].float(), r))
                acc[name][2].append(r.pow(2).mean().sqrt().item())
        print(f"== hot case {tag} (seed {seed}), {n_inputs} inputs of 64 x 64: rel rms error of each stage tap against the fp32 oracle")
        print(f"{'stage':18s} {'HIP':>10s} {'emulated fp16':>14s} {'HIP / emu':>10s} {'|ref| rms':>10s}")
        first = None
        for name in order:
            h, e, r = (float(np.mean(v)) for v in acc[name])
            ratio = h / max(e, 1e-12)
            flag = ""
            if ratio > 1.3 and first is None:
                first, flag = name, "   <-- first stage with HIP > 1.3 x emulation"
            print(f"{name:18s} {h:10.3e} {e:14.3e} {ratio:10.2f} {r:10.3f}{flag}")
        print(f"first stage above 1.3x: {first}\n")


if __name__ == "__main__":
    main()
