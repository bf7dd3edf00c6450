"""CPU: the pieces pinned by IMPORTING the reference's xiuminglib (tests/golden/make_io_metric_golden.py):
  * `datasets.nlt.read_png` == xm.io.img.load(path, as_array=True) on committed PNG fixtures (uint8 RGB / gray / RGBA and
    a 16-bit gray) -- the decoder under `load_store` (nlt/datasets/nlt.py:118-130);
  * `oracle.metric.psnr` / `rgb2lum` == xm.metric.PSNR(np.float32) / xm.img.rgb2lum."""
import os

import numpy as np
import pytest

from nlt_amd.datasets.nlt import read_png
from oracle import metric as M

G = os.path.join(os.path.dirname(__file__), 'golden')
D = np.load(os.path.join(G, 'io_metric.npz'))


@pytest.mark.parametrize('name', ['rgb8', 'gray8', 'rgba8', 'gray16'])
def test_png_reader_matches_xiuminglib_load(name):
    got = read_png(os.path.join(G, 'png', name + '.png'))
    ref = D['load_' + name]
    assert got.dtype == ref.dtype and np.array_equal(got, ref)


def test_oracle_psnr_and_luma_match_xiuminglib():
    np.testing.assert_allclose(M.rgb2lum(D['lum_in']), D['lum_out'], rtol=0, atol=0)
    for i, (plain, masked) in enumerate(D['psnr_values']):
        a, b, m = D['psnr_a%d' % i], D['psnr_b%d' % i], D['psnr_m%d' % i]
        assert abs(M.psnr(a, b) - plain) <= 1e-12 * abs(plain)
        assert abs(M.psnr(a, b, mask=m) - masked) <= 1e-12 * abs(masked)


def test_product_psnr_dynamic_range_rules():
    from nlt_amd.metric import PSNR
    assert PSNR(np.float32).drange == 1.0 and PSNR('uint8').drange == 255.0 and PSNR(np.uint16).drange == 65535.0
    with pytest.raises(NotImplementedError):
        PSNR(np.int32)
