"""data_gen/util.py:45-70 on HIP."""
import torch

from .. import _capi as C


def remap(src, mapping, force_kbg=True):
    """data_gen/util.py:45-58: cv2.remap(src, mapping*w, mapping*h, INTER_LINEAR) with the top-left
    source texel (where the background samples from) forced black.
    src [h,w(,c)] uint8 or float32; mapping [H,W,>=2] in [0,1], x first (float64/32/16)."""
    return C.remap_bilinear(src.contiguous(), mapping.contiguous(), force_kbg)


def add_b_ch(img_rg):
    """data_gen/util.py:61-64."""
    assert img_rg.dim() == 3 and img_rg.shape[2] == 2, "Input should be HxWx2"
    return torch.cat((img_rg, torch.zeros_like(img_rg[:, :, :1])), 2)


def to_float16(data):
    """data_gen/util.py:67-70 save_float16_npy, minus the file write."""
    return data.to(torch.float16)
