"""The algebra behind the depth ViT-S's kernel-less LayerNorm (``GemmOsArgs::stats_out / stats_in``, the folding in
``nunif_amd/csrc/depth_anything.hip``): a producer writes per-token partial (sum, sum of squares) of the fp16 values it stores, the
consumer multiplies the RAW fp16 rows by ``W' = fp16(W diag(gamma))`` and finishes with ``r (W' x - mu wsum) + (b + W beta)``,
``wsum = sum_k W'``.  Restated here in torch with the kernel's precisions (fp16 operands, fp32 accumulation, E[x^2] - mu^2) and
compared with ``F.linear(F.layer_norm(x), W, b)`` — including a residual stream with a few very large channels, where a one-pass
variance is the thing to worry about."""
import pytest
import torch
import torch.nn.functional as F


def folded_linear(x16, w, b, gamma, beta, parts=12, eps=1e-6):
    T, K = x16.shape
    xs = x16.float()
    # producer: one partial per 32 channels, from the values as stored (fp16), fp32 sums
    su = xs.view(T, parts, K // parts).sum(-1)
    sq = (xs * xs).view(T, parts, K // parts).sum(-1)
    mu = su.sum(-1) / K
    var = torch.clamp(sq.sum(-1) / K - mu * mu, min=0.0)
    r = torch.rsqrt(var + eps)
    w16 = (w * gamma[None, :]).half()                       # what the MFMA multiplies
    wsum = w16.float().sum(-1)
    bias = b + (w.half().float() * beta[None, :]).sum(-1)
    acc = xs @ w16.float().t()                              # fp32 accumulation of fp16 x fp16 products
    return r[:, None] * acc - (r * mu)[:, None] * wsum[None, :] + bias[None, :]


@pytest.mark.parametrize("outliers", [False, True])
def test_folded_layernorm_linear_matches_layernorm_then_linear(outliers):
    g = torch.Generator().manual_seed(5)
    T, K, N = 64, 384, 1152
    x = torch.randn(T, K, generator=g)
    if outliers:                                            # DINOv2-style massive activations in a few channels
        x[:, 7] += 300.0
        x[:, 200] -= 150.0
        x[5] *= 20.0
    x16 = x.half()
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g) * 0.1
    gamma = 1.0 + 0.2 * torch.randn(K, generator=g)
    beta = 0.1 * torch.randn(K, generator=g)
    ref = F.linear(F.layer_norm(x16.float(), (K,), gamma, beta, eps=1e-6), w, b)
    out = folded_linear(x16, w, b, gamma, beta)
    # the kernel path it replaces rounds the normalised row to fp16 before the GEMM; the folded form is at least as close to fp32
    two_step = F.linear(F.layer_norm(x16.float(), (K,), gamma, beta, eps=1e-6).half().float(), w.half().float(), b)
    scale = ref.abs().mean()
    assert (out - ref).abs().max() / scale < 2e-2
    assert (out - ref).abs().mean() <= 1.5 * (two_step - ref).abs().mean() + 1e-6
