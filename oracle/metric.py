"""CPU restatement of the reference's PSNR (third_party/xiuminglib/xiuminglib/metric.py:105-151, img.py:600-611).

ORACLE = test infrastructure (see oracle/__init__.py).  Pinned: tests/golden/io_metric.npz holds what the reference's
own xm.metric.PSNR returns on seeded images (tests/golden/make_io_metric_golden.py imports and runs it)."""
import numpy as np


def rgb2lum(im):
    """img.py:600-611."""
    return 0.2126 * im[:, :, 0] + 0.7152 * im[:, :, 1] + 0.0722 * im[:, :, 2]


def psnr(im1, im2, mask=None, drange=1.0):
    """metric.py:118-151: float64, luma for 3-channel inputs, masked mean of squared differences."""
    im1 = np.asarray(im1).astype(float); im2 = np.asarray(im2).astype(float)
    if im1.ndim == 2:
        im1, im2 = im1[:, :, None], im2[:, :, None]
    if im1.shape[2] == 3:
        im1, im2 = rgb2lum(im1)[:, :, None], rgb2lum(im2)[:, :, None]
    mask = np.ones(im1.shape, bool) if mask is None else np.asarray(mask).astype(bool).reshape(im1.shape)
    se = np.square(im1[mask] - im2[mask])
    mse = np.sum(se) / np.sum(mask)
    return 10 * np.log10((drange ** 2) / mse)
