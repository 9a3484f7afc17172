"""Seeded synthetic weights and inputs (no pretrained weights or datasets exist offline).

Product-side data generators: ``bench.py``, ``tools/`` and the tests draw their random-init state dicts (in the
reference's / the public checkpoints' key layouts) and synthetic depth maps from here.  Pure torch on the CPU, no HIP.
Every generator conditions the weights so that the random net sits in the numeric regime of a trained one (residual
branches damped, image heads scaled into [0,1]) and makes every bias / bias table non-zero, so that parity tests exercise
them; the reasoning is in each docstring.  ``oracle/*.random_state_dict`` are thin aliases of these functions.
"""
import math

import torch
import torch.nn.functional as F

# ---- swin_unet --------------------------------------------------------------------------------------------------------
# Conditioning of the seeded test weights.  With plain xavier/kaiming init the residual stream of this 14-block
# net doubles every two blocks (rms 0.1 -> 13), attention logits reach +-60 and the softmax becomes a hard argmax:
# a regime no trained net is in, where fp16 storage (the reference's own GPU mode) flips winners and PSNR measures
# chaos instead of kernel correctness (measured: rel. error 1e-2 in the last attention).  So the two residual
# branch outputs (attn.proj, mlp.3) are scaled by BRANCH_GAIN (stream rms ends ~1.3, logits O(1)), and the head is
# scaled so the un-clamped image sits around 0.5 +- 0.2 like a real picture instead of saturating the clamp.
BRANCH_GAIN = 0.8
HEAD_GAIN = 0.12
WINDOW = (6, 6)

# ---- Depth-Anything-V2 ViT-S (published architecture) --------------------------------------------------------------------
EMBED, HEADS, DEPTH, PATCH, MLP = 384, 6, 12, 14, 1536
OUT_CH = (48, 96, 192, 384)
FEAT = 64


def relative_position_index(wh, ww):
    """torchvision's ``relative_position_index`` buffer for a wh x ww window: (yi - yj + wh - 1) * (2 ww - 1) + (xi - xj + ww - 1)."""
    ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
    coords = torch.stack([ys.flatten(), xs.flatten()])
    rel = coords[:, :, None] - coords[:, None, :]
    return ((rel[0] + wh - 1) * (2 * ww - 1) + rel[1] + ww - 1).flatten()


def window_score_bias_input(window):
    """WindowScoreBias buffers (index [N*N], unique normalised offsets [U,2]; nunif/modules/attention.py:347-372)."""
    sh, sw = window
    pos = [(y, x) for y in range(sh) for x in range(sw)]
    delta = [(a[0] - b[0], a[1] - b[1]) for a in pos for b in pos]
    uniq = sorted(set(delta))
    index = torch.tensor([uniq.index(d) for d in delta], dtype=torch.int64)
    ud = torch.tensor(uniq, dtype=torch.float32)
    return index, ud / ud.abs().max()


# regime="hot": the numeric regime of a TRAINED net that the benign conditioning above deliberately avoids:
# relative-position tables N(0, HOT_TABLE_STD) (trained swin tables reach several units: the bias decides the winner), residual
# branches NOT damped (stream rms grows 7-30x over the 14 blocks, |stream| reaches 50-160, logits tens of units), and HOT_OUTLIERS
# channels of the stem output scaled by HOT_OUTLIER_GAIN (the "massive activation" channels every trained transformer carries).  No fp16 engine is
# within 50 dB of fp32 here — the reference's own autocast mode is not — so these weights are used with the relative criterion
# of tests/test_gpu_hot_regime.py (no worse than the emulated fp16 reference), never with the absolute one.
HOT_TABLE_STD = 1.5
HOT_OUTLIERS = (5, 37, 70)
HOT_OUTLIER_GAIN = 6.0          # x 20 .. x 50 on a RANDOM net saturates the whole picture (the outliers feed every later weight at full
                                # strength, which a trained net's weights do not): 47-95 % of the output clamped, nothing left to compare
HOT_HEAD_GAIN = 0.02            # the undamped stream ends at rms 7-30 instead of 1.3: the head is scaled so that the output stays a picture


def swin_unet_state_dict(seed, scale_factor=2, base_dim=96, in_channels=3, out_channels=3, layer_norm=False, regime="benign"):
    """Seeded random weights in the reference's key layout and shapes.

    Magnitudes follow the reference initialisers (kaiming/xavier) but every bias and relative-position
    table is drawn N(0, 0.02) instead of zero so that bias handling is exercised.
    Deterministic for a given torch build (CPU generator).  ``regime="hot"``: see HOT_TABLE_STD above.
    """
    assert regime in ("benign", "hot")
    hot = regime == "hot"
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def normal(shape, std):
        return torch.randn(shape, generator=g) * std

    def uniform(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def lin(key, cin, cout, gain=1.0, bias_mean=0.0):
        sd[key + ".weight"] = uniform((cout, cin), gain * math.sqrt(6.0 / (cin + cout)))
        sd[key + ".bias"] = normal((cout,), 0.02) + bias_mean

    def conv(key, cin, cout, k):
        sd[key + ".weight"] = normal((cout, cin, k, k), math.sqrt(2.0 / (cout * k * k)))
        sd[key + ".bias"] = normal((cout,), 0.02)

    def stage(key, dim, heads, layers):
        for i in range(layers):
            p = f"{key}.block.{i}."
            lin(p + "attn.qkv", dim, dim * 3)
            lin(p + "attn.proj", dim, dim, 1.0 if hot else BRANCH_GAIN)
            sd[p + "attn.relative_position_bias_table"] = normal((121, heads), HOT_TABLE_STD if hot else 0.02)
            sd[p + "attn.relative_position_index"] = relative_position_index(*WINDOW)
            lin(p + "mlp.0", dim, dim * 2)
            lin(p + "mlp.3", dim * 2, dim, 1.0 if hot else BRANCH_GAIN)
            if layer_norm:      # LayerNormNoBias (swin_unet_4xl): weight only, drawn around 1 so that it is exercised
                sd[p + "norm1.weight"] = 1.0 + normal((dim,), 0.1)
                sd[p + "norm2.weight"] = 1.0 + normal((dim,), 0.1)

    c, h = base_dim, base_dim // 16
    P = "unet."
    conv(P + "patch.0", in_channels, c // 2, 3)
    conv(P + "patch.2", c // 2, c, 3)
    if hot:
        for ch in HOT_OUTLIERS:
            sd[P + "patch.2.weight"][ch] *= HOT_OUTLIER_GAIN
            sd[P + "patch.2.bias"][ch] *= HOT_OUTLIER_GAIN
    stage(P + "swin1", c, h, 2)
    conv(P + "down1.conv", c, c * 2, 2)
    stage(P + "swin2", c * 2, h, 2)
    conv(P + "down2.conv", c * 2, c * 2, 2)
    stage(P + "swin3", c * 2, h, 6)
    lin(P + "up2.proj", c * 2, c * 2 * 4)
    stage(P + "swin4", c * 2, h, 2)
    if scale_factor in (1, 2):
        lin(P + "up1.proj", c * 2, c * 4)
        stage(P + "swin5", c, h, 2)
        lin(P + "to_image.proj", c, out_channels * scale_factor ** 2, HOT_HEAD_GAIN if hot else HEAD_GAIN, 0.5)
    else:
        lin(P + "proj2", c, c * 2)
        lin(P + "up1.proj", c * 2, c * 2 * 4)
        stage(P + "swin5", c * 2, h, 2)
        if scale_factor == 8:        # ToImage :96-101: Linear, LeakyReLU(0.2), Linear
            lin(P + "to_image.proj.0", c * 2, out_channels * 64)
            lin(P + "to_image.proj.2", out_channels * 64, out_channels * 64, HOT_HEAD_GAIN if hot else HEAD_GAIN, 0.5)
        else:
            lin(P + "to_image.proj", c * 2, out_channels * scale_factor ** 2, HOT_HEAD_GAIN if hot else HEAD_GAIN, 0.5)
    return sd


def swin_unet_v2_state_dict(seed, scale_factor=2, base_dim=None, lv1_mlp_ratio=2, lv2_mlp_ratio=2, lv2_ratio=2,
                            first_layers=2, last_layers=3):
    """Seeded random weights of ``waifu2x.swin_unet_v2_{1,2,4}x`` in the reference's key layout and shapes
    (waifu2x/models/swin_unet_v2.py SwinUNetV2Base :272-312; ``base_dim`` defaults to the registered model's: 64 / 96 / 128).

    Same conditioning as ``swin_unet_state_dict``: xavier / kaiming magnitudes, the residual-branch outputs (head_proj,
    conv_mlp.w2 — this net has no norm in front of its MLP, so the stream grows multiplicatively — and the PatchDown / PatchUp
    convs beside their shortcuts) damped so that the stream rms stays O(1) over the 13 blocks, every bias N(0, 0.02), norm weights 1 + N(0, 0.1), the score-bias MLP
    drawn wide enough that the bias table matters (+-0.5), the resampling conv = nearest neighbour + N(0, 0.02) so that all 27
    taps are exercised, and ``scale_bias`` sized so that source + scale_bias * residual stays a picture inside [0, 1]."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    C = base_dim or {1: 64, 2: 96, 4: 128}[scale_factor]
    C2 = int(C * lv2_ratio)

    def normal(shape, std):
        return torch.randn(shape, generator=g) * std

    def lin(key, cin, cout, gain=1.0):
        bound = gain * math.sqrt(6.0 / (cin + cout))
        sd[key + ".weight"] = (torch.rand((cout, cin), generator=g) * 2 - 1) * bound
        sd[key + ".bias"] = normal((cout,), 0.02)

    def conv(key, cin, cout, k, gain=1.0):
        sd[key + ".weight"] = normal((cout, cin, k, k), gain * math.sqrt(2.0 / (cin * k * k)))
        sd[key + ".bias"] = normal((cout,), 0.02)

    def wac(p, dim, ws, mlp_ratio, conv_mlp=True):
        lin(p + "mha.mha.qkv_proj", dim, dim * 3)
        lin(p + "mha.mha.head_proj", dim, dim, 0.5)
        sd[p + "relative_bias.index"], sd[p + "relative_bias.delta"] = window_score_bias_input((ws, ws))
        sd[p + "relative_bias.to_bias.0.weight"] = normal((2 * ws, 2), 1.0)
        sd[p + "relative_bias.to_bias.0.bias"] = normal((2 * ws,), 0.3)
        sd[p + "relative_bias.to_bias.2.weight"] = normal((1, 2 * ws), 0.4)
        sd[p + "relative_bias.to_bias.2.bias"] = normal((1,), 0.1)
        sd[p + "norm.weight"] = 1.0 + normal((dim,), 0.1)
        mid = int(dim * mlp_ratio)
        conv(p + "conv_mlp.w1", dim, mid, 1)
        if conv_mlp:
            conv(p + "conv_mlp.w2", mid // 2, dim, 3, 0.35)
        else:
            conv(p + "conv_mlp.w2", mid, dim, 1, 0.2)

    P = "unet."
    conv(P + "ir.path1.0", 3, 16, 3)
    conv(P + "ir.path2.1", 12, 64, 1)
    wac(P + "ir.path2.2.", 64, 8, 1)
    wac(P + "ir.path2.3.", 64, 8, 1)
    conv(P + "patch", 32, C, 3)
    for i in range(first_layers):
        wac(f"{P}wac1.blocks.{i}.", C, [8, 6][i], lv1_mlp_ratio)
    conv(P + "down1.conv", C, C2, 2, 0.5)
    for i in range(4):
        wac(f"{P}wac2.blocks.{i}.", C2, 8, lv2_mlp_ratio)
    conv(P + "up1.proj", C2, C * 4, 1, 0.5)
    for i in range(last_layers):
        wac(f"{P}wac3.blocks.{i}.", C, 8, lv1_mlp_ratio, conv_mlp=i < last_layers - 1)
    conv(P + "to_residual_image.proj", C, 3 * scale_factor ** 2, 1, 0.5)
    sd[P + "to_image.scale_bias"] = torch.full((1,), 0.05)
    w = normal((3 * scale_factor ** 2, 3, 3, 3), 0.02)
    for c in range(3):
        w[c * scale_factor ** 2:(c + 1) * scale_factor ** 2, c, 1, 1] += 1.0
    sd[P + "to_image.resampling.weight"] = w
    return sd


def cunet_state_dict(seed, up=False, in_channels=3, out_channels=3, regime="benign"):
    """Seeded weights in the reference's key layout; biases non-zero; the two image heads are scaled so that z1 and
    the final image sit inside [0,1] like a trained net's (same reasoning as oracle.swin_unet.random_state_dict).
    ``regime="hot"`` (tests/test_gpu_hot_regime.py): every 3x3 conv 1.6x wider than kaiming (activations grow ~25x over the
    cascade), three output channels of each unet's first UNetConv x 8, SE gates driven into saturation (conv2 x 6)."""
    assert regime in ("benign", "hot")
    hot = regime == "hot"
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(key, cin, cout, k, gain=1.0, bias_mean=0.0, transposed=False):
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        sd[key + ".weight"] = torch.randn(shape, generator=g) * (gain * math.sqrt(2.0 / (cout * k * k)))
        sd[key + ".bias"] = torch.randn(cout, generator=g) * 0.02 + bias_mean

    def unet_conv_w(key, cin, mid, cout, se):
        conv(key + ".conv.0", cin, mid, 3)
        conv(key + ".conv.2", mid, cout, 3)
        if se:
            conv(key + ".seblock.conv1", cout, cout // 8, 1)
            conv(key + ".seblock.conv2", cout // 8, cout, 1)

    def bottom(key, deconv, gain):
        if deconv:
            conv(key, 64, out_channels, 4, gain=gain, bias_mean=0.5, transposed=True)
        else:
            conv(key, 64, out_channels, 3, gain=gain, bias_mean=0.5)

    p = "unet1."
    unet_conv_w(p + "conv1", in_channels, 32, 64, False)
    conv(p + "conv1_down", 64, 64, 2)
    unet_conv_w(p + "conv2", 64, 128, 64, True)
    conv(p + "conv2_up", 64, 64, 2, transposed=True)
    conv(p + "conv3", 64, 64, 3)
    bottom(p + "conv_bottom", up, 0.3)
    p = "unet2."
    unet_conv_w(p + "conv1", out_channels, 32, 64, False)
    conv(p + "conv1_down", 64, 64, 2)
    unet_conv_w(p + "conv2", 64, 64, 128, True)
    conv(p + "conv2_down", 128, 128, 2)
    unet_conv_w(p + "conv3", 128, 256, 128, True)
    conv(p + "conv3_up", 128, 128, 2, transposed=True)
    unet_conv_w(p + "conv4", 128, 64, 64, True)
    conv(p + "conv4_up", 64, 64, 2, transposed=True)
    conv(p + "conv5", 64, 64, 3)
    conv(p + "conv_bottom", 64, out_channels, 3, gain=0.12, bias_mean=0.0)
    if hot:
        for k in list(sd):
            if k.endswith(".weight") and sd[k].ndim == 4 and sd[k].shape[-1] == 3 and "conv_bottom" not in k:
                sd[k] = sd[k] * 1.6
            if k.endswith("seblock.conv2.weight"):
                sd[k] = sd[k] * 6.0
        for u in ("unet1.", "unet2."):
            for ch in HOT_OUTLIERS[:3]:
                sd[u + "conv1.conv.2.weight"][ch % 64] *= 8.0
                sd[u + "conv1.conv.2.bias"][ch % 64] *= 8.0
        # the heads see ~25x larger maps: keep z1 and the image pictures
        sd["unet1.conv_bottom.weight"] = sd["unet1.conv_bottom.weight"] * 0.04
        sd["unet2.conv_bottom.weight"] = sd["unet2.conv_bottom.weight"] * 0.02
    return sd


def row_flow_v3_state_dict(seed, regime="benign"):
    """Seeded weights in the reference's key layout, every bias non-zero; the last layer is scaled so that delta sits
    in the range of a trained model's (a few depth pixels), i.e. the warp really moves pixels.
    ``regime="hot"`` (tests/test_hot_regime.py): the residual branches undamped (x2), qkv x2.5 (window logits of tens of units), the
    relative-position bias MLP x3 — the same generator draws, scaled after the fact."""
    assert regime in ("benign", "hot")
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    def lin(key, cout, cin, k=None, std=None, bstd=0.05):
        shape = (cout, cin) if k is None else (cout, cin, k, k)
        fan = cin * (1 if k is None else k * k)
        sd[key + ".weight"] = rnd(*shape, std=std if std is not None else math.sqrt(1.0 / fan))
        sd[key + ".bias"] = rnd(cout, std=bstd)

    lin("blocks.0", 64, 24, 1)
    for bi, (window, hidden) in ((1, ((4, 4), 8)), (2, ((3, 3), 6))):
        p = f"blocks.{bi}."
        lin(p + "mha.mha.qkv_proj", 192, 64)
        lin(p + "mha.mha.head_proj", 64, 64, std=0.5 * math.sqrt(1.0 / 64))
        lin(p + "conv_mlp.0", 64, 64, 1)
        lin(p + "conv_mlp.3", 64, 64, 3, std=0.5 * math.sqrt(1.0 / 576))
        lin(p + "bias.to_bias.0", hidden, 2, std=1.0, bstd=0.3)
        lin(p + "bias.to_bias.2", 1, hidden, std=1.0, bstd=0.3)
        index, delta = window_score_bias_input(window)
        sd[p + "bias.index"], sd[p + "bias.delta"] = index, delta
    lin("last_layer.1", 1, 8, 3, std=2.0 * math.sqrt(1.0 / 72), bstd=1.0)
    sd["delta_scale"] = torch.tensor(1.0 / 127.0)
    if regime == "hot":
        _heat_window_net(sd)
    return sd


def _heat_window_net(sd, qkv=2.5, branch=2.0, bias=1.75):
    """The window-attention nets of iw3 (row_flow_v3, mlbw) in the regime of a trained net: see row_flow_v3_state_dict."""
    for k in list(sd):
        if k.endswith("mha.mha.qkv_proj.weight"):
            sd[k] = sd[k] * qkv
        elif k.endswith(("mha.mha.head_proj.weight", "conv_mlp.3.weight")):
            sd[k] = sd[k] * branch
        elif k.endswith(("bias.to_bias.0.weight", "bias.to_bias.2.weight")):
            sd[k] = sd[k] * bias


def conv_stack_state_dict(seed, kind):
    """Seeded weights of waifu2x.vgg_7 (``kind="vgg_7"``) / waifu2x.upconv_7 in the reference's ``net.N`` key layout; He-scaled
    so that activations keep their range through the stack, non-zero biases, image head centred on 0.5."""
    g = torch.Generator().manual_seed(seed)
    ch = (3, 32, 32, 64, 64, 128, 128, 3) if kind == "vgg_7" else (3, 16, 32, 64, 128, 128, 256, 3)
    sd, n = {}, len(ch) - 1
    for i in range(n):
        cin, cout, last = ch[i], ch[i + 1], i == n - 1
        deconv = last and kind == "upconv_7"
        k = 4 if deconv else 3
        shape = (cin, cout, k, k) if deconv else (cout, cin, k, k)
        fan_in = cin * (4 if deconv else 9)                 # a stride-2 4x4 transposed conv sums 2x2 taps per output pixel
        gain = 0.25 if last else 1.0
        sd[f"net.{2 * i}.weight"] = torch.randn(shape, generator=g) * (gain * math.sqrt(2.0 / (1.01 * fan_in)))
        sd[f"net.{2 * i}.bias"] = torch.randn(cout, generator=g) * 0.05 + (0.5 if last else 0.0)
    return sd


def light_inpaint_state_dict(seed, regime="benign"):
    """Seeded weights of inpaint.light_inpaint_v1 in the reference's key layout: every bias non-zero, LayerNorm weights
    around 1, the token-mixing matrices (proj_spatial) strong enough to matter (the reference initialises them ~1e-5 with
    bias 1), to_image centred on 0.5 so that the net output sits in the image range.
    ``regime="hot"`` (tests/test_hot_regime.py): stronger residual branches and token mixing (proj_out, glu_conv.w2, proj_spatial x1.7), LayerNorm
    gains spread (x [0.5, 2.5]), the image head rescaled to keep the picture in range."""
    assert regime in ("benign", "hot")
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    def lin(key, *shape, std=None, bstd=0.05, bmean=0.0):
        fan = 1
        for s in shape[1:]:
            fan *= s
        sd[key + ".weight"] = rnd(*shape, std=std if std is not None else math.sqrt(1.0 / fan))
        sd[key + ".bias"] = rnd(shape[0], std=bstd) + bmean

    def block(p, C, ws):
        N = ws * ws
        lin(p + "gmlp.gmlp.proj_in", 4 * C, C, std=math.sqrt(2.0 / C))
        sd[p + "gmlp.gmlp.proj_spatial.weight"] = rnd(N, N, 1, std=0.7 / math.sqrt(N))
        sd[p + "gmlp.gmlp.proj_spatial.bias"] = rnd(N, std=0.3) + 0.5
        lin(p + "gmlp.gmlp.proj_out", C, 2 * C, std=0.5 * math.sqrt(1.0 / (2 * C)))
        sd[p + "norm1.weight"] = 1.0 + rnd(C, std=0.1)
        sd[p + "norm2.weight"] = 1.0 + rnd(2 * C, std=0.1)
        lin(p + "glu_conv.w1", C, C, 1, 1, std=math.sqrt(2.0 / C))
        lin(p + "glu_conv.w2", C, C // 2, 3, 3, std=0.5 * math.sqrt(1.0 / (9 * C // 2)))

    sd["mask_bias"] = rnd(1, 96, 1, 1, std=0.3)
    lin("patch.0", 96, 48, 1, 1, std=math.sqrt(2.0 / 48))
    block("enc1.", 96, 16)
    lin("down", 192, 96, 2, 2)
    for i in range(4):
        block(f"enc2.{i}.", 192, 8)
    lin("up", 384, 192, 1, 1)
    block("dec1.", 96, 16)
    lin("to_image.1", 48, 96, 3, 3, std=0.0028 * math.sqrt(1.0 / (9 * 96)), bstd=0.05, bmean=0.5)
    if regime == "hot":
        for k in list(sd):
            if k.endswith(("gmlp.gmlp.proj_out.weight", "glu_conv.w2.weight", "gmlp.gmlp.proj_spatial.weight")):
                sd[k] = sd[k] * 1.7
            elif k.endswith(("norm1.weight", "norm2.weight")):
                n = sd[k].numel()
                sd[k] = sd[k] * torch.linspace(0.5, 2.5, n)[torch.randperm(n, generator=g)]
        sd["to_image.1.weight"] = sd["to_image.1.weight"] * 0.4
    return sd


def light_video_inpaint_state_dict(seed, base_dim=96, lv2_mlp_ratio=1):
    """Seeded weights of inpaint.light_video_inpaint_v1 (small: base_dim 96, lv2_mlp_ratio 1; medium 128 / 2; large 192 / 2) in
    the reference's key layout; same conventions as ``light_inpaint_state_dict``."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    def lin(key, *shape, std=None, bstd=0.05, bmean=0.0):
        fan = 1
        for s in shape[1:]:
            fan *= s
        sd[key + ".weight"] = rnd(*shape, std=std if std is not None else math.sqrt(1.0 / fan))
        sd[key + ".bias"] = rnd(shape[0], std=bstd) + bmean

    def block(p, C, N, ratio):
        V = C * ratio
        lin(p + "gmlp.gmlp.proj_in", 2 * V, C, std=math.sqrt(2.0 / C))
        sd[p + "gmlp.gmlp.proj_spatial.weight"] = rnd(N, N, 1, std=0.7 / math.sqrt(N))
        sd[p + "gmlp.gmlp.proj_spatial.bias"] = rnd(N, std=0.3) + 0.5
        lin(p + "gmlp.gmlp.proj_out", C, V, std=0.5 * math.sqrt(1.0 / V))
        sd[p + "norm1.weight"] = 1.0 + rnd(C, std=0.1)
        sd[p + "norm2.weight"] = 1.0 + rnd(V, std=0.1)
        lin(p + "glu_conv.w1", C, C, 1, 1, std=math.sqrt(2.0 / C))
        lin(p + "glu_conv.w2", C, C // 2, 3, 3, std=0.5 * math.sqrt(1.0 / (9 * C // 2)))

    C, r2 = base_dim, lv2_mlp_ratio
    sd["mask_bias"] = rnd(1, C, 1, 1, std=0.3)
    lin("patch", C, 3, 4, 4, std=math.sqrt(2.0 / 48))
    block("enc1.", C, 256, 2)
    lin("down", 2 * C, C, 2, 2)
    for i, (n, r) in enumerate(((64, r2), (12, 2), (64, r2), (12, 2), (64, r2))):
        block(f"enc2.{i}.", 2 * C, n, r)
    lin("up", 4 * C, 2 * C, 1, 1)
    block("dec1.", C, 256, 2)
    lin("to_image", 48, C, 1, 1, std=0.002 * math.sqrt(1.0 / C), bstd=0.05, bmean=0.5)
    return sd


def mlbw_state_dict(seed, num_layers=2, small=False, hole_mask=False, regime="benign"):
    """Seeded weights in the reference's key layout (every bias non-zero); the output conv is scaled so that the layer
    deltas differ by a few depth pixels and the layer-weight logits really select between them.  ``regime="hot"``: as
    row_flow_v3_state_dict."""
    assert regime in ("benign", "hot")
    g = torch.Generator().manual_seed(seed)
    C = 32 * num_layers
    sd = {}

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    def lin(key, *shape, std=None, bstd=0.05):
        fan = 1
        for s in shape[1:]:
            fan *= s
        sd[key + ".weight"] = rnd(*shape, std=std if std is not None else math.sqrt(1.0 / fan))
        sd[key + ".bias"] = rnd(shape[0], std=bstd)

    lin("lv1_in.1", C // 8, 3, 1, 9, std=math.sqrt(2.0 / 27))
    for i in range(2 if small else 4):
        p = f"lv2.{i}."
        lin(p + "mha.mha.qkv_proj", 3 * C, C)
        lin(p + "mha.mha.head_proj", C, C, std=0.5 * math.sqrt(1.0 / C))
        lin(p + "conv_mlp.0", C, C, 1, 1)
        lin(p + "conv_mlp.3", C, C, 3, 3, std=0.5 * math.sqrt(1.0 / (9 * C)))
        lin(p + "bias.to_bias.0", 8, 2, std=1.0, bstd=0.3)
        lin(p + "bias.to_bias.2", 1, 8, std=1.0, bstd=0.3)
        sd[p + "bias.index"], sd[p + "bias.delta"] = window_score_bias_input((4, 4))
    lin("lv1_out.1", 2 * num_layers + (1 if hole_mask else 0), C // 8, 1, 9, std=2.0 * math.sqrt(1.0 / (9 * C // 8)),
        bstd=1.0)
    if hole_mask:
        sd["lv1_out.1.bias"][2 * num_layers] = -1.7     # logit(0.15): the default threshold cuts through the map
    if regime == "hot":
        _heat_window_net(sd, qkv=2.0, branch=1.5, bias=1.5)       # four blocks deep: milder per block than row_flow's two
    return sd


def depth_aa_state_dict(seed):
    """Seeded weights in the reference's key layout, all biases non-zero; proj_out (zero-initialised in the reference
    constructor) is given small weights so that the net really changes the depth (a few percent of its range)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    def lin(key, *shape, std=None, bstd=0.05):
        fan = 1
        for s in shape[1:]:
            fan *= s
        sd[key + ".weight"] = rnd(*shape, std=std if std is not None else math.sqrt(1.0 / fan))
        sd[key + ".bias"] = rnd(shape[0], std=bstd)

    lin("proj_in", 32, 4, 1, 1, std=0.7)
    for i in range(3):
        p = f"blocks.{i}."
        lin(p + "mha.mha.qkv_proj", 96, 32)
        lin(p + "mha.mha.head_proj", 32, 32, std=0.5 * math.sqrt(1.0 / 32))
        lin(p + "conv_mlp.0", 32, 32, 1, 1)
        lin(p + "conv_mlp.3", 32, 32, 3, 3, std=0.5 * math.sqrt(1.0 / 288))
        lin(p + "bias.to_bias.0", 16, 2, std=1.0, bstd=0.3)
        lin(p + "bias.to_bias.2", 1, 16, std=1.0, bstd=0.3)
        sd[p + "bias.index"], sd[p + "bias.delta"] = window_score_bias_input((8, 8))
    lin("proj_out", 4, 32, 1, 1, std=0.006, bstd=0.003)
    return sd


DEPTH_ANYTHING_ENCODERS = {      # published geometries (Depth-Anything-V2 dpt.py model_configs)
    "vits": dict(embed=384, depth=12, out_channels=(48, 96, 192, 384), features=64),
    "vitb": dict(embed=768, depth=12, out_channels=(96, 192, 384, 768), features=128),
    "vitl": dict(embed=1024, depth=24, out_channels=(256, 512, 1024, 1024), features=256),
}


def depth_anything_v2_state_dict(seed, grid=37, encoder="vits", regime="benign"):
    """Seeded weights in the public checkpoint's key layout (``encoder``: vits / vitb / vitl).  LayerScale gammas are
    O(1) * 0.3 (0.2 for the 24 blocks of vitl) and the residual branches are damped so that the blocks keep the token rms O(1)
    (a trained ViT's regime), every bias is non-zero.
    ``regime="hot"`` (tests/test_gpu_hot_regime.py): DINOv2-style MASSIVE ACTIVATIONS — block 1's MLP writes +-45 into two
    embedding channels of every token (fc2 bias; the rest of the token has rms ~1, so LayerNorm statistics are dominated by two
    of 384 values), a common offset of +12 on every channel from block 3 on (mean >> spread: the E[x^2] - mean^2 form of the
    folded LayerNorm is at its worst), and qkv weights 2x wider (logits tens of units)."""
    assert regime in ("benign", "hot")
    hot = regime == "hot"
    g = torch.Generator().manual_seed(seed)
    sd = {}
    cfg = DEPTH_ANYTHING_ENCODERS[encoder]
    EMBED, DEPTH, OUT_CH, FEAT = cfg["embed"], cfg["depth"], cfg["out_channels"], cfg["features"]
    MLP = 4 * EMBED
    ls = 0.3 if DEPTH == 12 else 0.2

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    def lin(key, *shape, std=None, bstd=0.02, bias=True):
        fan = 1
        for s in shape[1:]:
            fan *= s
        sd[key + ".weight"] = rnd(*shape, std=std if std is not None else math.sqrt(1.0 / fan))
        if bias:
            sd[key + ".bias"] = rnd(shape[0], std=bstd)

    p = "pretrained."
    lin(p + "patch_embed.proj", EMBED, 3, PATCH, PATCH)
    sd[p + "cls_token"] = rnd(1, 1, EMBED, std=0.5)
    sd[p + "pos_embed"] = rnd(1, 1 + grid * grid, EMBED, std=0.3)
    for i in range(DEPTH):
        b = f"{p}blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[b + n + ".weight"] = 1.0 + rnd(EMBED, std=0.1)
            sd[b + n + ".bias"] = rnd(EMBED, std=0.05)
        lin(b + "attn.qkv", 3 * EMBED, EMBED, std=(3.0 if hot else 1.5) * math.sqrt(1.0 / EMBED))
        lin(b + "attn.proj", EMBED, EMBED)
        lin(b + "mlp.fc1", MLP, EMBED)
        lin(b + "mlp.fc2", EMBED, MLP)
        sd[b + "ls1.gamma"] = ls + rnd(EMBED, std=0.05)
        sd[b + "ls2.gamma"] = ls + rnd(EMBED, std=0.05)
    sd[p + "norm.weight"] = 1.0 + rnd(EMBED, std=0.1)
    sd[p + "norm.bias"] = rnd(EMBED, std=0.05)
    if hot:
        b1 = sd[p + "blocks.1.mlp.fc2.bias"]
        b1[7] += 45.0 / float(sd[p + "blocks.1.ls2.gamma"][7])
        b1[200] -= 45.0 / float(sd[p + "blocks.1.ls2.gamma"][200])
        sd[p + "blocks.3.mlp.fc2.bias"] += 12.0 / sd[p + "blocks.3.ls2.gamma"]
    h = "depth_head."
    for i, oc in enumerate(OUT_CH):
        lin(f"{h}projects.{i}", oc, EMBED, 1, 1)
        lin(f"{h}scratch.layer{i + 1}_rn", FEAT, oc, 3, 3, bias=False)
    sd[h + "resize_layers.0.weight"] = rnd(OUT_CH[0], OUT_CH[0], 4, 4, std=math.sqrt(1.0 / OUT_CH[0]))
    sd[h + "resize_layers.0.bias"] = rnd(OUT_CH[0], std=0.02)
    sd[h + "resize_layers.1.weight"] = rnd(OUT_CH[1], OUT_CH[1], 2, 2, std=math.sqrt(1.0 / OUT_CH[1]))
    sd[h + "resize_layers.1.bias"] = rnd(OUT_CH[1], std=0.02)
    lin(h + "resize_layers.3", OUT_CH[3], OUT_CH[3], 3, 3)
    for k in (1, 2, 3, 4):
        r = f"{h}scratch.refinenet{k}."
        lin(r + "out_conv", FEAT, FEAT, 1, 1)
        for u in ("resConfUnit1.", "resConfUnit2."):
            lin(r + u + "conv1", FEAT, FEAT, 3, 3, std=0.7 * math.sqrt(2.0 / (9 * FEAT)))
            lin(r + u + "conv2", FEAT, FEAT, 3, 3, std=0.7 * math.sqrt(2.0 / (9 * FEAT)))
    lin(h + "scratch.output_conv1", FEAT // 2, FEAT, 3, 3)
    lin(h + "scratch.output_conv2.0", 32, FEAT // 2, 3, 3, std=math.sqrt(2.0 / (9 * 32)))
    lin(h + "scratch.output_conv2.2", 1, 32, 1, 1, std=math.sqrt(2.0 / 32), bstd=0.5)
    # a depth map, not a mostly-clipped one: positive mixing weights and bias in the last 1x1 (the ReLU then rarely bites)
    sd[h + "scratch.output_conv2.2.weight"] = sd[h + "scratch.output_conv2.2.weight"].abs() * 0.5
    sd[h + "scratch.output_conv2.2.bias"] = sd[h + "scratch.output_conv2.2.bias"].abs() * 0.2 + 0.1
    return sd


def video_depth_anything_state_dict(seed, grid=37, encoder="vits"):
    """Seeded weights in the published Video-Depth-Anything checkpoint layout: ``pretrained.*`` (DINOv2) + ``head.*`` = the DPT
    head of Depth-Anything V2 plus ``head.motion_modules.{0..3}`` (temporal attention on layer_3, layer_4, path_4, path_3;
    channels out_channels[2], out_channels[3], features, features).  ``proj_out`` is zero-initialised in an UNTRAINED module
    (the module is then the identity); a trained one is not, so it gets weights here and the temporal path changes the output
    by a measurable amount (tests).  The ``pos_encoder.pe`` buffers are left out: they are the sinusoidal table, recomputed."""
    base = depth_anything_v2_state_dict(seed, grid=grid, encoder=encoder)
    sd = {(("head." + k[len("depth_head."):]) if k.startswith("depth_head.") else k): v for k, v in base.items()}
    cfg = DEPTH_ANYTHING_ENCODERS[encoder]
    g = torch.Generator().manual_seed(seed + 7919)

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    chans = (cfg["out_channels"][2], cfg["out_channels"][3], cfg["features"], cfg["features"])
    for i, C in enumerate(chans):
        t = f"head.motion_modules.{i}.temporal_transformer."
        sd[t + "norm.weight"] = 1.0 + rnd(C, std=0.1)
        sd[t + "norm.bias"] = rnd(C, std=0.05)
        sd[t + "proj_in.weight"] = rnd(C, C, std=math.sqrt(1.0 / C))
        sd[t + "proj_in.bias"] = rnd(C, std=0.02)
        b = t + "transformer_blocks.0."
        for a in range(2):
            ab = f"{b}attention_blocks.{a}."
            sd[ab + "to_q.weight"] = rnd(C, C, std=2.0 * math.sqrt(1.0 / C))       # logits of a few units: a softmax that selects
            sd[ab + "to_k.weight"] = rnd(C, C, std=2.0 * math.sqrt(1.0 / C))
            sd[ab + "to_v.weight"] = rnd(C, C, std=math.sqrt(1.0 / C))
            sd[ab + "to_out.0.weight"] = rnd(C, C, std=0.7 * math.sqrt(1.0 / C))
            sd[ab + "to_out.0.bias"] = rnd(C, std=0.02)
            sd[f"{b}norms.{a}.weight"] = 1.0 + rnd(C, std=0.1)
            sd[f"{b}norms.{a}.bias"] = rnd(C, std=0.05)
        sd[b + "ff_norm.weight"] = 1.0 + rnd(C, std=0.1)
        sd[b + "ff_norm.bias"] = rnd(C, std=0.05)
        sd[b + "ff.net.0.proj.weight"] = rnd(8 * C, C, std=math.sqrt(1.0 / C))
        sd[b + "ff.net.0.proj.bias"] = rnd(8 * C, std=0.02)
        sd[b + "ff.net.2.weight"] = rnd(C, 4 * C, std=0.7 * math.sqrt(1.0 / (4 * C)))
        sd[b + "ff.net.2.bias"] = rnd(C, std=0.02)
        sd[t + "proj_out.weight"] = rnd(C, C, std=0.5 * math.sqrt(1.0 / C))
        sd[t + "proj_out.bias"] = rnd(C, std=0.02)
    return sd


def synth_depth(seed, b, h, w, kind="edges"):
    """Synthetic normalised depth maps: the step pattern of the reference's ``_bench`` (forward_warp.py:309-316)
    blurred, plus a ramp / smooth noise so that floor/ceil collisions, holes and layered holes all occur."""
    g = torch.Generator().manual_seed(seed)
    yy = torch.linspace(0, 1, h).view(1, 1, h, 1)
    xx = torch.linspace(0, 1, w).view(1, 1, 1, w)
    if kind == "ramp":
        d = (0.2 + 0.6 * xx + 0.1 * yy).expand(b, 1, h, w).clone()
    elif kind == "const":
        d = torch.full((b, 1, h, w), 0.5)
    else:
        d = torch.zeros(b, 1, h, w)
        d[:, :, h // 8:h - h // 8, w // 8:w - w // 8] = 0.3
        d[:, :, h // 4:h - h // 4, w // 4:w - w // 3] = 0.7
        d[:, :, h // 3:h // 2, w // 2:w - w // 6] = 1.0
        low = torch.rand(b, 1, max(2, h // 32), max(2, w // 32), generator=g)
        d = d * 0.85 + 0.15 * F.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)
        if kind == "smooth_edges":
            d = F.avg_pool2d(F.pad(d, (2, 2, 2, 2), mode="replicate"), 5, stride=1)
    return torch.clamp(d, 0, 1)
