"""Oracle: tile grid, blend filter and tiled render (torch CPU / pure ints).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Follows ``nunif/utils/seam_blending.py`` (reference) — line numbers cited per function.
"""
import math
import torch
import torch.nn.functional as F


def create_config(x_h, x_w, scale, offset, tile_size, blend_size):
    """Integer tile grid.  Reference: ``SeamBlending.create_config`` seam_blending.py:109-143.

    The block counters are the reference's while-loops in closed form: the smallest ``n >= 1`` with
    ``(n-1)*step + tile >= size + 2*input_offset``.
    """
    io = math.ceil(offset / scale)
    ib = math.ceil(blend_size / scale)
    step = tile_size - (io * 2 + ib)
    assert step > 0

    def blocks(size):
        need = size + io * 2
        n = max(1, -(-(need - tile_size) // step) + 1)
        return n, (n - 1) * step + tile_size

    hb, in_h = blocks(x_h)
    wb, in_w = blocks(x_w)
    return {
        "y_h": math.floor(x_h * scale), "y_w": math.floor(x_w * scale),
        "h_blocks": hb, "w_blocks": wb,
        "pad": (io, in_w - (x_w + io), io, in_h - (x_h + io)),
        "y_buffer_h": in_h * scale, "y_buffer_w": in_w * scale,
        "input_tile_step": step, "output_tile_step": step * scale,
    }


def blend_ramp(n_out, blend_size):
    """1-D ramp r with F[y, x] = min(r[y], r[x]).

    Reference: ``create_blend_filter`` seam_blending.py:146-153 pads a block of ones ``blend_size`` times with
    the constant ``1 - (1/(blend+1))*(i+1)`` (a Python double, stored to fp32) — the i-th pad ring ends up
    ``blend_size-1-i`` pixels from the border, hence the closed form below (SURVEY.md §7 "4K 4x buffers").
    """
    r = torch.ones(n_out, dtype=torch.float32)
    for d in range(blend_size):          # d = distance from the tile border
        i = blend_size - 1 - d
        value = 1 - (1 / (blend_size + 1)) * (i + 1)
        r[d] = value
        r[n_out - 1 - d] = value
    return r


def blend_filter(scale, offset, tile_size, blend_size, channels):
    n_out = tile_size * scale - offset * 2
    r = blend_ramp(n_out, blend_size)
    f = torch.minimum(r[:, None], r[None, :])
    return f.unsqueeze(0).expand(channels, n_out, n_out).contiguous()


def tiled_render(x, model_fn, scale, offset, blend_size, tile_size, batch_size=4):
    """Reference: ``SeamBlending.tiled_render`` seam_blending.py:48-106 and ``update`` :156-174.

    ``model_fn``: [B,C,T,T] -> [B,C,T*s-2*off,T*s-2*off].  Cumulative running-mean update, exactly in the
    reference's tile order (row-major) and arithmetic order.
    """
    blend_size = blend_size or 0
    c, h, w = x.shape
    cfg = create_config(h, w, scale, offset, tile_size, blend_size)
    pixels = torch.zeros(c, cfg["y_buffer_h"], cfg["y_buffer_w"], dtype=torch.float32)
    weights = torch.zeros_like(pixels) if blend_size > 0 else None
    filt = blend_filter(scale, offset, tile_size, blend_size, c) if blend_size > 0 else None
    xp = F.pad(x.unsqueeze(0), cfg["pad"], mode="replicate")[0]
    coords = [(i, j) for i in range(cfg["h_blocks"]) for j in range(cfg["w_blocks"])]
    istep, ostep = cfg["input_tile_step"], cfg["output_tile_step"]
    for b0 in range(0, len(coords), batch_size):
        chunk = coords[b0:b0 + batch_size]
        mb = torch.stack([xp[:, i * istep:i * istep + tile_size, j * istep:j * istep + tile_size]
                          for i, j in chunk])
        z = model_fn(mb)
        for k, (i, j) in enumerate(chunk):
            t = z[k]
            th, tw = t.shape[1:]
            ys, xs = slice(ostep * i, ostep * i + th), slice(ostep * j, ostep * j + tw)
            if blend_size > 0:
                w_old = weights[:, ys, xs]
                w_new = w_old + filt
                a = w_old / w_new
                pixels[:, ys, xs] = pixels[:, ys, xs] * a + t * (1 - a)
                weights[:, ys, xs] = w_new
            else:
                pixels[:, ys, xs] = t
    return torch.clamp(pixels[:, :cfg["y_h"], :cfg["y_w"]], 0.0, 1.0).contiguous()


def tiled_render_closed_form(x, model_fn, scale, offset, blend_size, tile_size, batch_size=4):
    """Single-pass form the HIP stitcher implements: out = sum_k F*tile_k / sum_k F over covering tiles,
    accumulated in the same row-major tile order (equals the cumulative form to ~1e-7)."""
    blend_size = blend_size or 0
    c, h, w = x.shape
    cfg = create_config(h, w, scale, offset, tile_size, blend_size)
    if blend_size == 0:
        return tiled_render(x, model_fn, scale, offset, blend_size, tile_size, batch_size)
    num = torch.zeros(c, cfg["y_buffer_h"], cfg["y_buffer_w"], dtype=torch.float32)
    den = torch.zeros_like(num)
    filt = blend_filter(scale, offset, tile_size, blend_size, c)
    xp = F.pad(x.unsqueeze(0), cfg["pad"], mode="replicate")[0]
    coords = [(i, j) for i in range(cfg["h_blocks"]) for j in range(cfg["w_blocks"])]
    istep, ostep = cfg["input_tile_step"], cfg["output_tile_step"]
    for b0 in range(0, len(coords), batch_size):
        chunk = coords[b0:b0 + batch_size]
        mb = torch.stack([xp[:, i * istep:i * istep + tile_size, j * istep:j * istep + tile_size]
                          for i, j in chunk])
        z = model_fn(mb)
        for k, (i, j) in enumerate(chunk):
            t = z[k]
            th, tw = t.shape[1:]
            ys, xs = slice(ostep * i, ostep * i + th), slice(ostep * j, ostep * j + tw)
            num[:, ys, xs] += filt * t
            den[:, ys, xs] += filt
    out = num[:, :cfg["y_h"], :cfg["y_w"]] / den[:, :cfg["y_h"], :cfg["y_w"]]
    return torch.clamp(out, 0.0, 1.0).contiguous()
