"""Restatement of torchvision's Swin V1 block (``torchvision.models.swin_transformer.SwinTransformerBlock``).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference imports this class from torchvision (pinned ``torchvision==0.22.1`` in
``requirements-torch.txt``; call site ``waifu2x/models/swin_unet.py:9-12,26-36``); torchvision is not
installed in this image and is not vendored under ``/root/reference``, so the published algorithm of
torchvision 0.22 ``ShiftedWindowAttention`` / ``shifted_window_attention`` is restated here (SURVEY.md
Appendix A).  **Parity status: pinned against an independent implementation** — HuggingFace ``transformers``
``SwinLayer`` carrying the same weights (``tests/test_tv_swin_block_vs_hf.py``: max |diff| <= 2e-5 for 6 / 12 heads,
C = 96 / 192, no shift / shift 3, padded and non-square maps, LayerNormNoBias; a wrong shift is shown to fail), and the
reference's own ``SwinUNetBase`` run over that HuggingFace layer reproduces the fixtures made over this class to 7e-7
(``tests/golden/make_golden_hf.py`` -> ``swin_unet_hf.npz``, which the GPU tests read).  Still not compared with a real
torchvision install (none exists where this code runs; ``tests/test_tv_swin_block_live.py`` does it wherever one does).
Also pinned: the parameter names/shapes that the reference's own ``SwinUNetBase`` produces on top of this class
(3 757 431 / 3 758 304 / 4 302 852 parameters for the 1x/2x/4x nets — asserted in tests/test_oracle_vs_reference.py).

State-dict keys (per block): ``norm1.*``, ``attn.qkv.{weight,bias}``, ``attn.proj.{weight,bias}``,
``attn.relative_position_bias_table`` [(2w-1)^2, heads], ``attn.relative_position_index`` [w^4] (buffer),
``norm2.*``, ``mlp.0.{weight,bias}``, ``mlp.3.{weight,bias}``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def relative_position_index(wh, ww):
    """index[i*N + j] = (yi - yj + wh-1) * (2*ww-1) + (xi - xj + ww-1), tokens row-major in the window."""
    ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + (wh - 1)
    dx = xs[:, None] - xs[None, :] + (ww - 1)
    return (dy * (2 * ww - 1) + dx).reshape(-1)


def shift_region_ids(ph, pw, ws, shift):
    """9-region id map used to mask attention across the roll seam (values 0..8)."""
    ids = torch.zeros(ph, pw)
    hb = (0, ph - ws[0], ph - shift[0], ph)
    wb = (0, pw - ws[1], pw - shift[1], pw)
    n = 0
    for a in range(3):
        for b in range(3):
            ids[hb[a]:hb[a + 1], wb[b]:wb[b + 1]] = n
            n += 1
    return ids


def window_partition(x, ws):
    b, h, w, c = x.shape
    x = x.view(b, h // ws[0], ws[0], w // ws[1], ws[1], c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws[0] * ws[1], c)


def window_merge(x, ws, b, h, w):
    c = x.shape[-1]
    x = x.view(b, h // ws[0], w // ws[1], ws[0], ws[1], c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, c)


class ShiftedWindowAttention(nn.Module):
    def __init__(self, dim, window_size, shift_size, num_heads):
        super().__init__()
        self.window_size = list(window_size)
        self.shift_size = list(shift_size)
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        n_rel = (2 * window_size[0] - 1) * (2 * window_size[1] - 1)
        self.relative_position_bias_table = nn.Parameter(torch.zeros(n_rel, num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", relative_position_index(*window_size))

    def forward(self, x):
        ws = self.window_size
        b, h, w, c = x.shape
        pad_r = (ws[1] - w % ws[1]) % ws[1]
        pad_b = (ws[0] - h % ws[0]) % ws[0]
        x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        ph, pw = h + pad_b, w + pad_r
        shift = [0 if ws[0] >= ph else self.shift_size[0], 0 if ws[1] >= pw else self.shift_size[1]]
        shifted = sum(shift) > 0
        if shifted:
            x = torch.roll(x, shifts=(-shift[0], -shift[1]), dims=(1, 2))
        t = window_partition(x, ws)                       # [b*nw, n, c]
        nwin = (ph // ws[0]) * (pw // ws[1])
        n = ws[0] * ws[1]
        hd = c // self.num_heads
        qkv = F.linear(t, self.qkv.weight, self.qkv.bias)
        qkv = qkv.reshape(t.shape[0], n, 3, self.num_heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index]
        attn = attn + bias.view(n, n, -1).permute(2, 0, 1).unsqueeze(0)
        if shifted:
            ids = shift_region_ids(ph, pw, ws, shift).to(x.device)
            ids = ids.view(ph // ws[0], ws[0], pw // ws[1], ws[1]).permute(0, 2, 1, 3).reshape(nwin, n)
            diff = ids[:, None, :] - ids[:, :, None]
            mask = torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))
            attn = attn.view(b, nwin, self.num_heads, n, n) + mask[None, :, None]
            attn = attn.view(-1, self.num_heads, n, n)
        attn = F.softmax(attn, dim=-1)
        t = (attn @ v).transpose(1, 2).reshape(t.shape[0], n, c)
        t = F.linear(t, self.proj.weight, self.proj.bias)
        x = window_merge(t, ws, b, ph, pw)
        if shifted:
            x = torch.roll(x, shifts=(shift[0], shift[1]), dims=(1, 2))
        return x[:, :h, :w, :].contiguous()


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio=4.0, dropout=0.0,
                 attention_dropout=0.0, stochastic_depth_prob=0.0, norm_layer=nn.LayerNorm,
                 attn_layer=None):
        super().__init__()
        assert dropout == 0.0 and attention_dropout == 0.0 and stochastic_depth_prob == 0.0
        self.norm1 = norm_layer(dim)
        self.attn = ShiftedWindowAttention(dim, window_size, shift_size, num_heads)
        self.norm2 = norm_layer(dim)
        hidden = int(dim * mlp_ratio)
        # indices 0 and 3 carry the weights (Linear, GELU, Dropout, Linear, Dropout)
        self.mlp = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Identity(),
                                 nn.Linear(hidden, dim), nn.Identity())
        for m in self.mlp.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.normal_(m.bias, std=1e-6)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x
