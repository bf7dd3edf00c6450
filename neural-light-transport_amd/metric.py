"""Image metrics with the reference's call signatures (third_party/xiuminglib/xiuminglib/metric.py) on libnlt_hip.so."""
import math

import numpy as np
import torch

from . import _capi as C


DEVICE = 'cuda'


def _on_device(x):
    if torch.is_tensor(x):
        return x if x.device.type == torch.device(DEVICE).type else x.to(DEVICE)
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEVICE)


class PSNR:
    """xm.metric.PSNR (metric.py:105-151): Peak Signal-to-Noise Ratio in dB on luma (0.2126 r + 0.7152 g + 0.0722 b for
    3-channel inputs), float64 arithmetic, optional H x W logical mask.  `dtype` fixes the dynamic range the way
    metric.Base does: 1 for float types, max - min for unsigned integer types."""

    def __init__(self, dtype):
        self.dtype = np.dtype(dtype)
        if self.dtype.kind == 'f':
            self.drange = 1.
        elif self.dtype.kind == 'u':
            info = np.iinfo(self.dtype)
            self.drange = float(info.max - info.min)
        else:
            raise NotImplementedError(self.dtype.kind)

    def __call__(self, im1, im2, mask=None):
        """im1, im2: [H,W] / [H,W,1] / [H,W,3] torch CUDA tensors (float32 storage) in [0, drange] -> float dB.
        NumPy arrays / host tensors (what the reference's vis_batch hands over) go up to `DEVICE` first: the sums are
        always the kernel's."""
        im1, im2 = _on_device(im1), _on_device(im2)
        if tuple(im1.shape) != tuple(im2.shape):
            raise AssertionError("The two images are not even of the same shape")
        if im1.dim() == 3 and im1.shape[2] not in (1, 3):
            raise NotImplementedError("%d-channel images" % im1.shape[2])
        a, b = im1.float().contiguous(), im2.float().contiguous()
        m = None if mask is None else mask.to(device=a.device, dtype=torch.uint8).contiguous()
        if m is not None and tuple(m.shape[:2]) != tuple(a.shape[:2]):
            raise AssertionError("Mask must be of shape %s, but is of shape %s" % (tuple(a.shape[:2]), tuple(m.shape)))
        se, n = C.psnr_sums(a, b, m).tolist()
        mse = se / n
        return 10 * math.log10((self.drange ** 2) / mse) if mse > 0 else float('inf')
