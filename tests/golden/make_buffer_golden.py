"""Writes tests/golden/buffer_assembly.npz by IMPORTING and running the reference's own Python
for the buffer-assembly rows that are importable in the build container (no TF / cv2 / Blender):

  data_gen/get_neighbors.py:52-71            get_neighbors              (run unmodified)
  xiuminglib/img.py:289-431                  grid_query_unstruct        (run unmodified, except that
      its `cv2` module global -- None here, cv2 is not installable -- is replaced by a shim whose
      distanceTransform(x, DIST_L1, 3) is scipy.ndimage.distance_transform_cdt(x, 'taxicab'))
  xiuminglib/img.py:11-54                    normalize_uint / denormalize_float (run unmodified)
  nlt/util/net.py:18-56                      gen_feat_n                 (run unmodified)

Run in the build container only (/root/reference does not exist on the GPU box):
    python tests/golden/make_buffer_golden.py
"""
import os
import sys
import types

import numpy as np
import scipy.ndimage

REF = '/root/reference'
sys.path.insert(0, os.path.join(REF, 'third_party', 'xiuminglib'))
sys.path.insert(0, os.path.join(REF, 'data_gen'))
import xiuminglib as xm           # noqa: E402
import get_neighbors as gn        # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
out = {}

# ---- get_neighbors: random positions (no ties) and an exact-tie lattice (first minimum wins)
rng = np.random.default_rng(61)
cand = rng.normal(size=(9, 3))
ref = np.concatenate((cand, rng.normal(size=(5, 3))), 0)          # trainvali + test, as in main()
named = lambda a, p: [{'name': '%s%03d' % (p, i), 'position': list(map(float, x))} for i, x in enumerate(a)]
nn = gn.get_neighbors(named(ref, 'r'), named(cand, 'c'))
out['knn_ref'], out['knn_cand'] = ref, cand
out['knn_nn'] = np.array([int(nn['r%03d' % i][1:]) for i in range(len(ref))], np.int32)
lat = np.array([[x, y, 0] for y in range(3) for x in range(3)], np.float64)   # 3x3 integer lattice: many exact ties
nn = gn.get_neighbors(named(lat, 'r'), named(lat, 'c'))
out['knn_lat'] = lat
out['knn_lat_nn'] = np.array([int(nn['r%03d' % i][1:]) for i in range(len(lat))], np.int32)

# ---- grid_query_unstruct (nearest, max_l1_interp=4), cv2.distanceTransform shimmed with SciPy
shim = types.SimpleNamespace(
    DIST_L1=1,
    distanceTransform=lambda src, dist_type, mask: scipy.ndimage.distance_transform_cdt(src, metric='taxicab').astype(np.float32))
xm.img.cv2 = shim
method = {'func': 'griddata', 'func_underlying': 'nearest', 'fill_value': (0,), 'max_l1_interp': 4}
for tag, (h, w, p) in {'a': (24, 20, 90), 'b': (16, 16, 40), 'c': (33, 47, 500)}.items():
    centres = rng.random((3, 2))
    uvs = np.clip(centres[rng.integers(0, 3, p)] + 0.12 * rng.normal(size=(p, 2)), -0.05, 1.05)
    vals = rng.random((p, 2))
    out['gq_%s_uvs' % tag], out['gq_%s_vals' % tag] = uvs, vals
    out['gq_%s_res' % tag] = np.array([h, w])
    out['gq_%s_out' % tag] = xm.img.grid_query_unstruct(uvs, vals, (h, w), method=method)

# ---- normalize_uint / denormalize_float
u8 = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
u16 = rng.integers(0, 65536, (4, 6), dtype=np.uint16)
f = rng.random((9, 8))
f[0, :4] = [0.0, 1.0, 1 / 255, 254.999999 / 255]
out['norm_u8'], out['norm_u8_out'] = u8, xm.img.normalize_uint(u8)
out['norm_u16'], out['norm_u16_out'] = u16, xm.img.normalize_uint(u16)
out['denorm_f'], out['denorm_f_out'] = f, xm.img.denormalize_float(f)

# ---- channel schedule: nlt/util/net.py:18-56 imported and run
sys.path.insert(0, os.path.join(REF, 'nlt', 'util'))
import net as refnet              # noqa: E402
combos = [(a, b, c) for a in (4, 8, 16, 32) for b in (16, 64, 128, 256, 1024) for c in (1, 3, 4) if b >= a and b >= c]
out['feat_n_args'] = np.array(combos, np.int32)
sched = [refnet.gen_feat_n(*x) for x in combos]
out['feat_n_len'] = np.array([len(x) for x in sched], np.int32)
out['feat_n_flat'] = np.array([v for x in sched for v in x], np.int32)

np.savez_compressed(os.path.join(OUT, 'buffer_assembly.npz'), **out)
print('wrote buffer_assembly.npz:', {k: v.shape for k, v in out.items()})
