"""TEST-SIDE stand-in for OpenCV (not installed here): the two calls ``WarpFrame`` makes
(pfrl/wrappers/atari_wrappers.py:136-146) -- RGB -> grey and an area-averaging resize -- in NumPy.
Not pixel-identical to OpenCV; the Atari example scripts only need frames of the right shape,
dtype and variability to run their pipeline."""
import types

import numpy as np

COLOR_RGB2GRAY = 7
INTER_AREA = 3
ocl = types.SimpleNamespace(setUseOpenCL=lambda flag: None)


def cvtColor(frame, code):
    assert code == COLOR_RGB2GRAY
    f = np.asarray(frame, dtype=np.float32)
    return (0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2] + 0.5).astype(np.uint8)


def resize(frame, size, interpolation=INTER_AREA):
    w, h = size
    f = np.asarray(frame, dtype=np.float32)
    H, W = f.shape[:2]
    ys = (np.arange(h + 1) * H) // h
    xs = (np.arange(w + 1) * W) // w
    # mean over the source cell of every destination pixel (integral image)
    ii = np.zeros((H + 1, W + 1), dtype=np.float64)
    ii[1:, 1:] = f.cumsum(0).cumsum(1)
    area = (ys[1:] - ys[:-1])[:, None] * (xs[1:] - xs[:-1])[None, :]
    s = ii[ys[1:]][:, xs[1:]] - ii[ys[:-1]][:, xs[1:]] - ii[ys[1:]][:, xs[:-1]] + ii[ys[:-1]][:, xs[:-1]]
    return (s / area + 0.5).astype(np.uint8)
