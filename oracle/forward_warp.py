"""Oracle: iw3 depth-ordered bilinear forward warp (torch CPU fp32), restated as row-local closed forms.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ``iw3/forward_warp.py`` (reference): ``make_bilinear_data`` :75-85, ``ordered_index_copy`` :88-110,
``warp`` :113-132, ``shift_fill`` :18-30, ``shift_fill_pack`` :33-42, ``fix_layered_holes`` :45-59, ``gen_mask2``
:135-137, ``depth_order_bilinear_forward_warp`` :140-243, ``apply_divergence_forward_warp`` :246-256.

The reference sorts ALL pixels by depth and overwrites destinations in that order (``index_copy_``); because every
shift is horizontal this equals, per destination pixel, "the source with the largest depth wins" evaluated
separately for the floor and the ceil target.  The iterative hole loops are restated as
capped row scans.  ``tests/test_oracle_vs_reference.py`` checks bit-exact equality with the live reference.
"""
import torch
import torch.nn.functional as F

UNDEFINED = -1.0
LAYERED = -2.0
MAX_TRIES = 100


def _zwinner(depth_row_key, dest, width):
    """For each destination column the source column whose (depth, src) key is largest; -1 when none.
    depth_row_key: [N, Wp] int64 monotone key of depth (larger = nearer), dest: [N, Wp] int64 destination column."""
    n, wp = dest.shape
    src = torch.arange(wp).expand(n, wp)
    key = depth_row_key * (wp + 1) + src + 1                      # > 0, unique per source in a row
    best = torch.zeros(n, wp, dtype=torch.int64)
    best.scatter_reduce_(1, dest, key, reduce="amax", include_self=True)
    win = best % (wp + 1) - 1                                     # source column or -1
    return win


def _depth_key(depth):
    """Order-preserving int64 key of fp32 values (handles negatives)."""
    bits = depth.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    neg = bits >= 0x80000000
    return torch.where(neg, 0xFFFFFFFF - bits, bits + 0x80000000)


def warp_one_eye(c4, depth, index_shift):
    """``warp`` :113-132 for one eye.  c4: [B,C,H,Wp] (last channel = x index), depth/index_shift: [B,H,Wp]."""
    b, ch, h, wp = c4.shape
    x = torch.arange(wp, dtype=torch.float32).view(1, 1, wp)
    fidx = torch.clamp(x + index_shift, 0, wp - 1)
    floor_i = torch.clamp(fidx.floor(), 0, wp - 1)
    ceil_i = torch.clamp(fidx.ceil(), 0, wp - 1)
    cw = torch.clamp(fidx - floor_i, min=1e-5, max=1.0 - 1e-5)    # source-indexed weights
    fw = 1.0 - cw
    key = _depth_key(depth).view(b * h, wp)
    wf = _zwinner(key, floor_i.long().view(b * h, wp), wp).view(b, h, wp)
    wc = _zwinner(key, ceil_i.long().view(b * h, wp), wp).view(b, h, wp)

    def gather(winner, weight):
        has = winner >= 0
        idx = winner.clamp(min=0)
        wgt = torch.where(has, torch.gather(weight, 2, idx), torch.zeros(()))
        val = torch.gather(c4, 3, idx.unsqueeze(1).expand(b, ch, h, wp))
        val = torch.where(has.unsqueeze(1), val, torch.full((), UNDEFINED))
        return wgt.unsqueeze(1), val

    fwt, fval = gather(wf, fw)
    cwt, cval = gather(wc, cw)
    out = (fval * fwt + cval * cwt) / (fwt + cwt)
    return torch.nan_to_num(out, UNDEFINED)


def shift_fill(x, sign, max_tries=MAX_TRIES):
    """``shift_fill`` :18-30 (flip_sign=False) in closed form.  sign<0: take from the left, sign>0: from the right;
    zero inflow at the border; after ``max_tries`` steps an unfilled element holds the (negative) value found
    ``max_tries`` positions away."""
    if sign > 0:
        return shift_fill(x.flip(-1), -1, max_tries).flip(-1)
    w = x.shape[-1]
    xp = F.pad(x, (max_tries, 0))                                  # virtual zeros left of the row (>= 0: "defined")
    out = x.clone()
    todo = x < 0
    for k in range(1, max_tries + 1):
        cand = xp[..., max_tries - k:max_tries - k + w]
        take = todo & ((cand >= 0) | (k == max_tries))
        out = torch.where(take, cand, out)
        todo = todo & ~take
    return out


def fix_layered_holes(side, idx, sign, max_tries=MAX_TRIES):
    """``fix_layered_holes`` :45-59 in closed form: windowed running min (sign>0) / max (sign<0) of the index row;
    pixels whose index changed are marked LAYERED on every channel.  Returns (side, idx)."""
    w = idx.shape[-1]
    if sign > 0:
        pad = F.pad(idx, (0, max_tries), value=float("inf"))
        new = torch.stack([pad[..., k:k + w] for k in range(max_tries + 1)], 0).amin(0)
    else:
        pad = F.pad(idx, (max_tries, 0), value=float("-inf"))
        new = torch.stack([pad[..., max_tries - k:max_tries - k + w] for k in range(max_tries + 1)], 0).amax(0)
    changed = new != idx
    side = torch.where(changed.expand_as(side), torch.full((), LAYERED), side)
    return side, new


def gen_mask(eye):
    m = eye[:, 0:1]
    return torch.clamp((m == UNDEFINED).float() + (m == LAYERED).float() * 0.5, 0, 1)


def forward_warp(c, depth, divergence, convergence, fill=True, synthetic_view="both", return_mask=False,
                 width_base=True):
    """``depth_order_bilinear_forward_warp`` :140-243 (inconsistent_shift=False)."""
    assert synthetic_view in ("both", "left", "right")
    src = c
    if c.shape[2:] != depth.shape[2:]:
        depth = F.interpolate(depth, size=c.shape[-2:], mode="bilinear", align_corners=True, antialias=True)
    if synthetic_view != "both":
        divergence = divergence * 2
    base = c.shape[-1] if width_base else max(c.shape[-2:])
    pad = int(base * divergence * 0.01 + 2)
    cp = F.pad(c, (pad, pad, 0, 0), mode="replicate")
    dp = F.pad(depth, (pad, pad, 0, 0), mode="replicate")
    b, _, h, wp = dp.shape
    shift_size = divergence * 0.01 * base * 0.5
    index_shift = (dp * shift_size - (shift_size * convergence)).view(b, h, wp)
    xi = torch.arange(wp, dtype=c.dtype).view(1, 1, 1, wp).expand(b, 1, h, wp)
    c4 = torch.cat([cp, xi], 1)
    d = dp.view(b, h, wp)

    def eye(sign):
        e = warp_one_eye(c4, d, index_shift * sign)[..., pad:wp - pad]
        img, idx = e[:, :-1], e[:, -1:]
        # left eye (sign>0): holes take from the left; the right eye is processed flipped (shift_fill_pack :38-41)
        idx = shift_fill(idx, -1 if sign > 0 else 1)
        img, idx = fix_layered_holes(img, idx, 1 if sign > 0 else -1)
        mask = gen_mask(img) if return_mask else None
        img = shift_fill(img, -1 if sign > 0 else 1) if fill else torch.clamp(img, 0, 1)
        return img.contiguous(), mask

    left, lmask = eye(1) if synthetic_view in ("both", "left") else (src, None)
    right, rmask = eye(-1) if synthetic_view in ("both", "right") else (src, None)
    if return_mask:
        return left, right, lmask, rmask
    return left, right


def synth_depth(*args, **kwargs):
    """Alias of ``nunif_amd.synthetic.synth_depth``."""
    from nunif_amd.synthetic import synth_depth as f
    return f(*args, **kwargs)


def nonwarp_mask(c, depth, divergence, convergence, view="right"):
    """``iw3/forward_warp.py`` ``nonwarp_mask`` :259-295 on top of this module's ``forward_warp`` restatement."""
    divergence = divergence * 0.5
    if c.shape[-2:] != depth.shape[-2:]:
        depth = F.interpolate(depth, size=c.shape[-2:], mode="bilinear", align_corners=True, antialias=True)
    depth3 = depth.repeat(1, 3, 1, 1)
    if view == "right":
        wd, _ = forward_warp(depth3, depth, divergence, convergence, fill=True, synthetic_view="left")
        wd = wd.mean(dim=1, keepdim=True)
        _, _, _, mask = forward_warp(torch.zeros_like(c), wd, divergence, convergence, fill=False, synthetic_view="right",
                                     return_mask=True)
    else:
        c, depth, depth3 = c.flip(-1), depth.flip(-1), depth3.flip(-1)
        _, wd = forward_warp(depth3, depth, divergence, convergence, fill=True, synthetic_view="right")
        wd = wd.mean(dim=1, keepdim=True)
        _, _, mask, _ = forward_warp(torch.zeros_like(c), wd, divergence, convergence, fill=False, synthetic_view="left",
                                     return_mask=True)
        c, mask = c.flip(-1), mask.flip(-1)
    return c, mask
