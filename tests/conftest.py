import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def psnr(a, b):
    """Float PSNR as the reference computes it: 10*log10(1/(mse+1e-6)) on [0,1]
    (nunif/cli/diff_image.py:14-19, nunif/modules/psnr.py:17-19)."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    mse = torch.mean((a - b) ** 2).item()
    return 10.0 * float(np.log10(1.0 / (mse + 1.0e-6)))


def synth_image(seed, c, h, w):
    """Smooth + noise image in [0,1]; must stay identical to tests/golden/make_golden.py::synth_image."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, c, max(2, h // 16 + 1), max(2, w // 16 + 1), generator=g)
    up = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)[0]
    return torch.clamp(up * 0.8 + 0.2 * torch.rand(c, h, w, generator=g), 0, 1)


def hot_image(seed, h, w):
    """synth_image with saturated flats (blocks of exact 0 / 1); must stay identical to tests/golden/make_golden.py::hot_image."""
    x = synth_image(seed, 3, h, w)
    x[:, : h // 3, : w // 2] = 1.0
    x[:, h // 2:, w // 3: 2 * w // 3] = 0.0
    x[1, h // 4: h // 2, w // 2:] = 1.0
    return x


def sd_checksum(sd):
    return float(sum(v.double().sum().item() for v in sd.values() if v.is_floating_point()))


@pytest.fixture(scope="session")
def golden_swin():
    return dict(np.load(os.path.join(GOLDEN, "swin_unet.npz")))


@pytest.fixture(scope="session")
def golden_swin_hf():
    """The reference's SwinUNet* over HuggingFace's SwinLayer (tests/golden/make_golden_hf.py): an oracle-independent pin."""
    return dict(np.load(os.path.join(GOLDEN, "swin_unet_hf.npz")))


@pytest.fixture(scope="session")
def hiplib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    from nunif_amd import _hip, build
    if not os.path.exists(_hip.LIB_PATH):
        build.build(verbose=False)
    return _hip.lib()


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.inference_mode():
        yield
