"""Oracle: waifu2x swin_unet forward as pure functions over a reference-format ``state_dict`` (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ``waifu2x/models/swin_unet.py`` (reference): ``SwinUNetBase.forward`` :180-199, ``PatchDown`` :45-62,
``PatchUp`` :65-82, ``ToImage`` :85-116, wrappers ``SwinUNet``/``SwinUNet2x``/``SwinUNet4x`` :208-296 and
``SwinUNetDownscaled`` :339-379.  The transformer block follows torchvision 0.22 ``SwinTransformerBlock``
(external, see oracle/tv_swin_block.py for the pinning status).

Keys are the reference's (prefix ``unet.``): ``patch.{0,2}``, ``swin{1..5}.block.{i}.attn.{qkv,proj}``,
``...attn.relative_position_bias_table``, ``...mlp.{0,3}``, ``down{1,2}.conv``, ``up{1,2}.proj``, ``proj2``,
``to_image.proj``.
"""
import math
import torch
import torch.nn.functional as F

from .tv_swin_block import relative_position_index, shift_region_ids, window_partition, window_merge

WINDOW = (6, 6)


def swin_block(sd, p, x, heads, shifted, taps=None):
    """One V1 block, Identity norm (``NO_NORM_LAYER`` swin_unet.py:16-17) or LayerNormNoBias when present.
    ``taps`` (dict) optionally receives ``qkv`` and ``attn`` (pre-projection) as un-rolled NHWC maps."""
    def norm(t, key):
        w = sd.get(p + key + ".weight")
        return t if w is None else F.layer_norm(t, (t.shape[-1],), w, None, 1e-5)

    b, h, w, c = x.shape
    assert h % 6 == 0 and w % 6 == 0
    hd = c // heads
    shift = (3 if (shifted and h > 6) else 0, 3 if (shifted and w > 6) else 0)
    t = norm(x, "norm1")
    if sum(shift):
        t = torch.roll(t, (-shift[0], -shift[1]), (1, 2))
    t = window_partition(t, WINDOW)
    n = 36
    qkv = F.linear(t, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    if taps is not None:
        m = window_merge(qkv, WINDOW, b, h, w)
        taps["qkv"] = torch.roll(m, shift, (1, 2)) if sum(shift) else m
    qkv = qkv.reshape(-1, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    a = q @ k.transpose(-2, -1)
    idx = relative_position_index(*WINDOW)
    a = a + sd[p + "attn.relative_position_bias_table"][idx].view(n, n, heads).permute(2, 0, 1)[None]
    if sum(shift):
        nw = (h // 6) * (w // 6)
        ids = shift_region_ids(h, w, WINDOW, shift).view(h // 6, 6, w // 6, 6).permute(0, 2, 1, 3).reshape(nw, n)
        m = (ids[:, None, :] != ids[:, :, None]).float() * -100.0
        a = (a.view(b, nw, heads, n, n) + m[None, :, None]).view(-1, heads, n, n)
    a = a.softmax(-1)
    t = (a @ v).transpose(1, 2).reshape(-1, n, c)
    if taps is not None:
        m = window_merge(t, WINDOW, b, h, w)
        taps["attn"] = torch.roll(m, shift, (1, 2)) if sum(shift) else m
    t = F.linear(t, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    t = window_merge(t, WINDOW, b, h, w)
    if sum(shift):
        t = torch.roll(t, shift, (1, 2))
    x = x + t
    t = norm(x, "norm2")
    t = F.gelu(F.linear(t, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
    t = F.linear(t, sd[p + "mlp.3.weight"], sd[p + "mlp.3.bias"])
    return x + t


def swin_stage(sd, p, x, heads, layers, taps=None, tap_prefix=""):
    """``SwinTransformerBlocks`` swin_unet.py:20-42: shift = window//2 on odd layers."""
    for i in range(layers):
        bt = {} if taps is not None else None
        x = swin_block(sd, f"{p}.block.{i}.", x, heads, shifted=(i % 2 == 1), taps=bt)
        if taps is not None:
            taps[f"{tap_prefix}.b{i}.qkv"] = bt["qkv"]
            taps[f"{tap_prefix}.b{i}.attn"] = bt["attn"]
            taps[f"{tap_prefix}.b{i}.out"] = x
    return x


def patch_down(sd, p, x):
    """NHWC -> conv 2x2 stride 2 -> NHWC.  swin_unet.py:55-62."""
    y = F.conv2d(x.permute(0, 3, 1, 2), sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=2)
    return y.permute(0, 2, 3, 1).contiguous()


def patch_up(sd, p, x):
    """Linear C->4C' then pixel_shuffle(2) in NHWC.  swin_unet.py:76-82."""
    y = F.linear(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"]).permute(0, 3, 1, 2)
    return F.pixel_shuffle(y, 2).permute(0, 2, 3, 1).contiguous()


def unet_forward(sd, x, scale_factor, prefix="unet.", taps=None):
    """``SwinUNetBase.forward`` swin_unet.py:180-199.  x: [B,3,T,T] fp32 -> [B,3,(T-16)*s,(T-16)*s].
    ``taps`` (dict) optionally receives named NHWC intermediates (same names as the HIP engine's debug taps)."""
    def tap(name, v):
        if taps is not None:
            taps[name] = v
    P = prefix
    c = sd[P + "patch.2.weight"].shape[0]
    heads = c // 16
    t = F.leaky_relu(F.conv2d(x, sd[P + "patch.0.weight"], sd[P + "patch.0.bias"]), 0.1)
    t = F.leaky_relu(F.conv2d(t, sd[P + "patch.2.weight"], sd[P + "patch.2.bias"]), 0.1)
    t = t[:, :, 6:-6, 6:-6]
    assert t.shape[2] % 12 == 0 and t.shape[2] % 16 == 0
    x2 = t.permute(0, 2, 3, 1).contiguous()
    tap("stem", x2)
    x3 = swin_stage(sd, P + "swin1", x2, heads, 2, taps, "swin1")
    d1 = patch_down(sd, P + "down1", x3)
    tap("down1", d1)
    x4 = swin_stage(sd, P + "swin2", d1, heads, 2, taps, "swin2")
    d2 = patch_down(sd, P + "down2", x4)
    tap("down2", d2)
    x5 = swin_stage(sd, P + "swin3", d2, heads, 6, taps, "swin3")
    y = patch_up(sd, P + "up2", x5) + x4
    tap("up2", y)
    y = swin_stage(sd, P + "swin4", y, heads, 2, taps, "swin4")
    y = patch_up(sd, P + "up1", y)
    if P + "proj2.weight" in sd:        # 4x / 8x nets: Linear(C, 2C) on the skip (swin_unet.py:166)
        y = y + F.linear(x3, sd[P + "proj2.weight"], sd[P + "proj2.bias"])
    else:
        y = y + x3
    tap("up1", y)
    y = swin_stage(sd, P + "swin5", y, heads, 2, taps, "swin5")
    if P + "to_image.proj.0.weight" in sd:  # 8x head: Linear, LeakyReLU(0.2), Linear (swin_unet.py:99-104)
        y = F.linear(y, sd[P + "to_image.proj.0.weight"], sd[P + "to_image.proj.0.bias"])
        y = F.linear(F.leaky_relu(y, 0.2), sd[P + "to_image.proj.2.weight"], sd[P + "to_image.proj.2.bias"])
    else:
        y = F.linear(y, sd[P + "to_image.proj.weight"], sd[P + "to_image.proj.bias"])
    y = y.permute(0, 3, 1, 2).contiguous()
    if scale_factor > 1:
        y = F.pixel_shuffle(y, scale_factor)
    return y


# name -> (i2i_scale, i2i_offset, i2i_blend_size, unet scale_factor)  swin_unet.py:213,234,267,311
GEOMETRY = {
    "waifu2x.swin_unet_1x": (1, 8, 4, 1),
    "waifu2x.swin_unet_2x": (2, 16, 8, 2),
    "waifu2x.swin_unet_4x": (4, 32, 16, 4),
    "waifu2x.swin_unet_8x": (4, 64, 32, 8),
}


def model_forward(sd, x, name="waifu2x.swin_unet_2x", downscale_factor=None):
    """Eval-mode wrapper: clamp(unet(x), 0, 1) (swin_unet.py:221-226,244-249); ``SwinUNetDownscaled`` adds
    bicubic-antialias /f and a second clamp (:366-379)."""
    if downscale_factor:
        z = torch.clamp(unet_forward(sd, x, 4), 0.0, 1.0)
        z = F.interpolate(z, size=(z.shape[-2] // downscale_factor, z.shape[-1] // downscale_factor),
                          mode="bicubic", align_corners=False, antialias=True)
        return torch.clamp(z, 0.0, 1.0)
    return torch.clamp(unet_forward(sd, x, GEOMETRY[name][3]), 0.0, 1.0)


def valid_tile_size(size):
    """swin_unet.py:202-205."""
    return size > 16 and (size - 16) % 12 == 0 and (size - 16) % 16 == 0


def find_valid_tile_size(base):
    """``_find_valid_tile_size`` nunif/models/model.py:51-62."""
    t = int(base)
    while t > 0:
        if valid_tile_size(t):
            return t
        t -= 1
    raise ValueError(f"Could not find valid tile size: tile_size={base}")


# Conditioning of the seeded test weights.  With plain xavier/kaiming init the residual stream of this 14-block
# net doubles every two blocks (rms 0.1 -> 13), attention logits reach +-60 and the softmax becomes a hard argmax:
# a regime no trained net is in, where fp16 storage (the reference's own GPU mode) flips winners and PSNR measures
# chaos instead of kernel correctness (measured: rel. error 1e-2 in the last attention).  So the two residual
# branch outputs (attn.proj, mlp.3) are scaled by BRANCH_GAIN (stream rms ends ~1.3, logits O(1)), and the head is
# scaled so the un-clamped image sits around 0.5 +- 0.2 like a real picture instead of saturating the clamp.
BRANCH_GAIN = 0.8
HEAD_GAIN = 0.12


def random_state_dict(*args, **kwargs):
    """Seeded test weights: alias of ``nunif_amd.synthetic.swin_unet_state_dict`` (moved there so that bench.py and the tools do
    not import the oracle for their inputs)."""
    from nunif_amd.synthetic import swin_unet_state_dict
    return swin_unet_state_dict(*args, **kwargs)
