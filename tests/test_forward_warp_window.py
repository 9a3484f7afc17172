"""The window step of ``fix_layered_holes`` (iw3/forward_warp.py:45-59: running min / max of the warped index over 101 pixels) as
``forward_warp_kernel<DIET, PI>`` evaluates it (nunif_amd/csrc/iw3_warp.hip, round 5), restated in numpy and held against the direct
definition: two overlapping 64-wide windows built by doubling, a thread owning the pixel PAIR (2 t, 2 t + 1) through all levels, and a
40-element apron on the side the window looks at, filled ONCE with the boundary element — every level equals it there, and it lies
inside every window that reaches that far, so reading it changes no minimum / maximum.  (The GPU tests hold the kernel itself
bit-exact against the oracle; this one pins the reasoning.)"""
import numpy as np
import pytest

K_TRIES, APRON = 100, 40
REM = K_TRIES + 1 - 64


def direct(row, left_eye):
    W = len(row)
    out = np.empty_like(row)
    for j in range(W):
        out[j] = row[j:min(j + K_TRIES, W - 1) + 1].min() if left_eye else row[max(j - K_TRIES, 0):j + 1].max()
    return out


def kernel_form(row, left_eye):
    W = len(row)
    assert W % 2 == 0 and W >= 128
    op = np.minimum if left_eye else np.maximum
    edge = row[W - 1] if left_eye else row[0]
    lv = [np.full(W + 2 * APRON, np.nan, row.dtype) for _ in range(2)]        # t0 / t1 with both aprons (only one side is ever read)
    for t in lv:
        if left_eye:
            t[APRON + W:] = edge
        else:
            t[:APRON] = edge
    own = np.empty(W, row.dtype)
    # pass st = 1 on pairs: the partner of the first element is the thread's own second one (left eye) and vice versa
    for j0 in range(0, W, 2):
        o0, o1 = row[j0], row[j0 + 1]
        if left_eye:
            own[j0], own[j0 + 1] = op(o0, o1), op(o1, row[min(j0 + 2, W - 1)])
        else:
            own[j0 + 1], own[j0] = op(o1, o0), op(o0, row[max(j0 - 1, 0)])
    lv[0][APRON:APRON + W] = own
    src, dst = 0, 1
    st = 2
    while st <= 32:
        new = own.copy()
        for j0 in range(0, W, 2):
            q = APRON + (j0 + st if left_eye else j0 - st)
            assert q % 2 == 0                                              # the 8-byte read is aligned
            pair = lv[src][q:q + 2]
            assert not np.isnan(pair).any()                                # never reads what was not written
            new[j0], new[j0 + 1] = op(own[j0], pair[0]), op(own[j0 + 1], pair[1])
        own = new
        lv[dst][APRON:APRON + W] = own
        src, dst = dst, src
        st *= 2
    out = np.empty_like(row)
    for j in range(W):
        out[j] = op(own[j], lv[src][APRON + (j + REM if left_eye else j - REM)])
    return out


@pytest.mark.parametrize("W", [128, 130, 400, 1920, 2200])
def test_pairwise_doubling_with_aprons_is_the_running_window(W):
    rng = np.random.default_rng(W)
    for left_eye in (True, False):
        for trial in range(3):
            row = rng.integers(-1, 60, W).astype(np.float32) + rng.random(W).astype(np.float32)
            row[rng.random(W) < 0.2] = -1.0                                 # holes, as shift_fill leaves them at the borders
            assert np.array_equal(kernel_form(row, left_eye), direct(row, left_eye)), (W, left_eye, trial)
