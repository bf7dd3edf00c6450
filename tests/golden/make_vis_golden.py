"""Writes tests/golden/vis.npz by IMPORTING and running the reference's own reporting helpers (run unmodified):

  xiuminglib/img.py:635-667      linear2srgb          (on float32 images, as Model.vis_batch feeds it)
  xiuminglib/io/img.py:61-85     write_arr            (x255, truncated to uint8, written with PIL; read back here)

Inputs are float32 camera-space images in [0, 1] with the awkward values seeded in: 0, 1, the sRGB threshold and its float32
neighbours, k/255 (where x255 lands on or just below an integer), and random values.

Run in the build container only (/root/reference does not exist on the GPU box):
    python tests/golden/make_vis_golden.py
"""
import os
import sys
import tempfile

import numpy as np

REF = '/root/reference'
sys.path.insert(0, os.path.join(REF, 'third_party', 'xiuminglib'))
import xiuminglib as xm           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(207)
h, w = 24, 40
special = np.array([0, 1, 0.0031308, np.nextafter(np.float32(0.0031308), np.float32(0)), np.nextafter(np.float32(0.0031308), np.float32(1)),
                    0.5, 1e-8, 254.5 / 255, 0.999999], np.float32)
lin = np.concatenate((special, (np.arange(256) / 255).astype(np.float32),
                      np.nextafter((np.arange(1, 256) / 255).astype(np.float32), np.float32(0)),
                      rng.random(h * w * 3 - special.size - 256 - 255, dtype=np.float32)))
lin = rng.permutation(lin).reshape(h, w, 3).astype(np.float32)
dark = (rng.random((h, w, 3), dtype=np.float32) * np.float32(0.01)).astype(np.float32)     # mostly the linear branch
out = {'lin': lin, 'dark': dark}
with tempfile.TemporaryDirectory() as d:
    for tag, im in (('lin', lin), ('dark', dark)):
        srgb = xm.img.linear2srgb(im)
        assert srgb.dtype == np.float32
        out[tag + '_srgb'] = srgb
        for space, arr in (('srgb', srgb), ('raw', im)):
            p = os.path.join(d, '%s_%s.png' % (tag, space))
            ret = xm.io.img.write_arr(arr, p)
            back = xm.io.img.load(p, as_array=True)
            assert np.array_equal(ret, back)
            out['%s_%s_u8' % (tag, space)] = back
np.savez_compressed(os.path.join(OUT, 'vis.npz'), **out)
print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
