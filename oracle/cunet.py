"""Oracle: waifu2x CUNet / UpCUNet forward as pure functions over a reference-format ``state_dict`` (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ``waifu2x/models/cunet.py`` (reference): ``UNetConv`` :10-28, ``UNet1`` :31-67, ``UNet2`` :70-121, ``CUNet``
:172-203, ``UpCUNet`` :139-169 and ``nunif/modules/attention.py`` ``SEBlock`` :29-44.
Keys: ``unet{1,2}.conv{N}.conv.{0,2}``, ``...seblock.conv{1,2}``, ``conv{N}_down``, ``conv{N}_up``, ``conv_bottom``.
"""
import math

import torch
import torch.nn.functional as F

SLOPE = 0.1


def _conv(sd, key, x, stride=1):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride)


def _deconv(sd, key, x, stride=2, padding=0):
    return F.conv_transpose2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=padding)


def unet_conv(sd, key, x):
    """Two VALID 3x3 convs + LeakyReLU(0.1), optional squeeze-excitation."""
    x = F.leaky_relu(_conv(sd, key + ".conv.0", x), SLOPE)
    x = F.leaky_relu(_conv(sd, key + ".conv.2", x), SLOPE)
    if key + ".seblock.conv1.weight" in sd:
        z = x.mean(dim=(2, 3), keepdim=True)
        z = F.relu(_conv(sd, key + ".seblock.conv1", z))
        z = torch.sigmoid(_conv(sd, key + ".seblock.conv2", z))
        x = x * z
    return x


def _bottom(sd, key, x):
    w = sd[key + ".weight"]
    if w.shape[2] == 4:      # UNet deconv=True: ConvTranspose2d(64, out, 4, 2, 3); weight is [in, out, 4, 4]
        return _deconv(sd, key, x, stride=2, padding=3)
    return _conv(sd, key, x)


def unet1(sd, p, x):
    x1 = unet_conv(sd, p + "conv1", x)
    x2 = F.leaky_relu(_conv(sd, p + "conv1_down", x1, stride=2), SLOPE)
    x2 = unet_conv(sd, p + "conv2", x2)
    x2 = F.leaky_relu(_deconv(sd, p + "conv2_up", x2), SLOPE)
    x3 = F.leaky_relu(_conv(sd, p + "conv3", x1[:, :, 4:-4, 4:-4] + x2), SLOPE)
    return _bottom(sd, p + "conv_bottom", x3)


def unet2(sd, p, x):
    x1 = unet_conv(sd, p + "conv1", x)
    x2 = F.leaky_relu(_conv(sd, p + "conv1_down", x1, stride=2), SLOPE)
    x2 = unet_conv(sd, p + "conv2", x2)
    x3 = F.leaky_relu(_conv(sd, p + "conv2_down", x2, stride=2), SLOPE)
    x3 = unet_conv(sd, p + "conv3", x3)
    x3 = F.leaky_relu(_deconv(sd, p + "conv3_up", x3), SLOPE)
    x4 = unet_conv(sd, p + "conv4", x2[:, :, 4:-4, 4:-4] + x3)
    x4 = F.leaky_relu(_deconv(sd, p + "conv4_up", x4), SLOPE)
    x5 = F.leaky_relu(_conv(sd, p + "conv5", x1[:, :, 16:-16, 16:-16] + x4), SLOPE)
    return _bottom(sd, p + "conv_bottom", x5)


def model_forward(sd, x, no_clip=False):
    """Eval-mode CUNet / UpCUNet (which one is decided by the shape of unet1.conv_bottom.weight)."""
    z1 = unet1(sd, "unet1.", x)
    if not no_clip:
        z1 = torch.clamp(z1, 0.0, 1.0)
    z2 = unet2(sd, "unet2.", z1)
    return torch.clamp(z1[:, :, 20:-20, 20:-20] + z2, 0.0, 1.0)


# name -> (i2i_scale, i2i_offset, blend)   cunet.py:143,177 (blend_size None -> 0: plain overwrite, no blending)
GEOMETRY = {"waifu2x.cunet": (1, 28, 0), "waifu2x.upcunet": (2, 36, 0)}


def conv_stack_forward(sd, x):
    """waifu2x.vgg_7 (vgg_7.py:11-30) / waifu2x.upconv_7 (upconv_7.py:11-35): ``net`` = 3x3 VALID convs with LeakyReLU(0.1)
    between them; the last layer is a plain conv (vgg_7) or ConvTranspose2d(256, 3, 4, 2, 3) (upconv_7); eval clamp."""
    keys = sorted({int(k.split(".")[1]) for k in sd if k.startswith("net.")})
    for i in keys[:-1]:
        x = F.leaky_relu(F.conv2d(x, sd[f"net.{i}.weight"], sd[f"net.{i}.bias"]), 0.1)
    w, b = sd[f"net.{keys[-1]}.weight"], sd[f"net.{keys[-1]}.bias"]
    x = F.conv_transpose2d(x, w, b, stride=2, padding=3) if w.shape[2] == 4 else F.conv2d(x, w, b)
    return x.clamp(0., 1.)


def conv_stack_state_dict(*args, **kwargs):
    from nunif_amd.synthetic import conv_stack_state_dict as f
    return f(*args, **kwargs)


def valid_tile_size(size):
    return size % 4 == 0


def random_state_dict(*args, **kwargs):
    """Seeded test weights: alias of ``nunif_amd.synthetic.cunet_state_dict`` (moved there so that bench.py and the tools do
    not import the oracle for their inputs)."""
    from nunif_amd.synthetic import cunet_state_dict
    return cunet_state_dict(*args, **kwargs)
