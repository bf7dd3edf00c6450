"""Regenerates tests/golden/*.npz from the reference tree's own golden DATA files.

Run in the build container only (`/root/reference` does not exist on the GPU box):
    python tests/golden/make_fixtures.py
Sources (data, not code):
  third_party/robust_loss/data/wavelet_golden.mat   (wavelet_test.py:146-165)
  third_party/robust_loss/data/partition_spline.npz (distribution.py:142-147)
"""
import os
import numpy as np
import scipy.io

REF = '/root/reference/third_party/robust_loss/data'
OUT = os.path.dirname(os.path.abspath(__file__))

d = scipy.io.loadmat(os.path.join(REF, 'wavelet_golden.mat'))
out = {'I_color': d['I_color']}
pyr = d['pyr_color'][0, :].tolist()
for lvl in range(len(pyr) - 1):
    for b, band in enumerate(pyr[lvl].flatten()):
        out['band_%d_%d' % (lvl, b)] = band
out['resid'] = pyr[-1]
np.savez_compressed(os.path.join(OUT, 'wavelet_golden.npz'), **out)

z = np.load(os.path.join(REF, 'partition_spline.npz'))
np.savez_compressed(os.path.join(OUT, 'partition_spline.npz'),
                    x_scale=z['x_scale'], values=z['values'], tangents=z['tangents'])
print('wrote', os.listdir(OUT))
