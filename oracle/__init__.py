"""CPU oracle for the NLT hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A restatement, in NumPy and torch-CPU primitives, of the arithmetic that
google/neural-light-transport runs on its UV-texture-space relighting path
(`nlt/models/nlt.py:89-199` and callees).  The reference is TensorFlow-2.2
eager Python; TensorFlow, TF-Addons and OpenCV are NOT installed in the build
environment, so the reference cannot be imported and every TF/TFA/cv2
primitive is restated here from its published semantics.

Parity status (be honest about it):
  * Barron / CDF9-7 wavelet / partition-spline pieces: PINNED against the
    reference's own golden data (`tests/golden/wavelet_golden.npz`,
    `tests/golden/partition_spline.npz`, closed-form tests).
  * k-NN (get_neighbors), the UV-index map (grid_query_unstruct) and
    normalize_uint / denormalize_float: PINNED against outputs of the
    reference's OWN Python, imported and run in the build container by
    tests/golden/make_buffer_golden.py (cv2.distanceTransform substituted by
    SciPy's taxicab transform) -> tests/golden/buffer_assembly.npz.
  * conv / deconv / resampler / resize / cv2.remap / cosine maps / diffuse
    base: PARITY UNPINNED -- the reference ships no tests or vectors for them
    and TF / TF-Addons / cv2 / Blender cannot run here.  They are cross-checked
    against independent naive NumPy loops written from the documented
    semantics (tests/test_oracle_ops.py, tests/test_oracle_buffers.py).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package.  The product package (`neural-light-transport_amd/`)
never does, and fails loudly if its HIP library is missing.
"""
