"""8-way dihedral test-time augmentation.  Mirrors ``nunif/transforms/tta.py`` ``tta_split`` :20-34 / ``tta_merge``
:37-48.  Pure data movement on the device (flips / rot90); the eight renders go through the HIP engine."""
import torch

# (transpose first?, vertical flip?, horizontal flip?) in the reference's order
_VIEWS = [(t, v, h) for t in (False, True) for v in (False, True) for h in (False, True)]


def _apply(x, t, v, h):
    if t:
        x = torch.rot90(x, 1, (1, 2))
    if v:
        x = torch.flip(x, (1,))
    if h:
        x = torch.flip(x, (2,))
    return x


def _invert(x, t, v, h):
    if h:
        x = torch.flip(x, (2,))
    if v:
        x = torch.flip(x, (1,))
    if t:
        x = torch.rot90(x, -1, (1, 2))
    return x


def tta_split(x):
    assert isinstance(x, torch.Tensor) and x.dim() == 3
    return tuple(_apply(x, *view) for view in _VIEWS)


def tta_merge(xs):
    assert len(xs) == 8
    avg = xs[0].clone()
    for y, view in zip(xs[1:], _VIEWS[1:]):
        avg += _invert(y, *view)
    avg *= 1 / 8.0
    return torch.clamp_(avg, 0, 1)
