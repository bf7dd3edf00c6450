"""Writes tests/golden/io_metric.npz and tests/golden/png/*.png by IMPORTING and running the reference's own Python
(importable in the build container: PIL + NumPy only):

  xiuminglib/io/img.py:12-33      load(path, as_array=True)   -- what nlt/datasets/nlt.py:118-130 decodes captures with
  xiuminglib/metric.py:105-151    PSNR(np.float32)            -- what nlt/models/nlt.py:64,259-269 reports per sample
  xiuminglib/img.py:600-611       rgb2lum

Run in the build container only (/root/reference does not exist on the GPU box):
    python tests/golden/make_io_metric_golden.py
"""
import os
import sys

import numpy as np
from PIL import Image

REF = '/root/reference'
sys.path.insert(0, os.path.join(REF, 'third_party', 'xiuminglib'))
import xiuminglib as xm           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PNG = os.path.join(OUT, 'png')
os.makedirs(PNG, exist_ok=True)
rng = np.random.default_rng(97)
out = {}

# ---- PNG fixtures as data_gen writes them (uint8 RGB / gray; RGBA and 16-bit for the refusal / slicing paths)
imgs = {'rgb8': rng.integers(0, 256, (12, 10, 3), dtype=np.uint8),
        'gray8': rng.integers(0, 256, (12, 10), dtype=np.uint8),
        'rgba8': rng.integers(0, 256, (6, 8, 4), dtype=np.uint8),
        'gray16': rng.integers(0, 65536, (6, 8), dtype=np.uint16)}
for name, arr in imgs.items():
    Image.fromarray(arr).save(os.path.join(PNG, name + '.png'))
    out['load_' + name] = xm.io.img.load(os.path.join(PNG, name + '.png'), as_array=True)
    assert np.array_equal(out['load_' + name], arr)

# ---- PSNR on luma, float32 images in [0, 1], with and without a mask
psnr = xm.metric.PSNR(np.float32)
cases = []
for i, (h, w, c) in enumerate([(16, 12, 3), (33, 17, 3), (8, 8, 1), (64, 64, 3)]):
    a = rng.random((h, w, c), dtype=np.float32) if c == 3 else rng.random((h, w), dtype=np.float32)
    b = np.clip(a + 0.05 * rng.standard_normal(a.shape).astype(np.float32), 0, 1).astype(np.float32)
    mask = rng.random((h, w)) > 0.4
    out['psnr_a%d' % i], out['psnr_b%d' % i], out['psnr_m%d' % i] = a, b, mask
    cases.append((psnr(a, b), psnr(a, b, mask=mask)))
out['psnr_values'] = np.array(cases, np.float64)
out['lum_in'] = rng.random((5, 7, 3))
out['lum_out'] = xm.img.rgb2lum(out['lum_in'])

np.savez_compressed(os.path.join(OUT, 'io_metric.npz'), **out)
print('wrote io_metric.npz:', {k: v.shape for k, v in out.items()})
