"""CPU oracle of NLT's texel-buffer assembly (SURVEY.md 8a rows a-B1 .. a-B6).

ORACLE = test infrastructure (see oracle/__init__.py); never imported by the product.

Every function restates one piece of the reference's offline data generation / data loading in
NumPy float64 (the reference's own precision) with a FIXED operation order, so that the HIP
kernels -- compiled with fp contraction off -- reproduce the integer / uint8 results bit-exactly.

Pinning (tests/test_oracle_buffers.py, fixtures made by tests/golden/make_buffer_golden.py which
IMPORTS the reference's Python where that is possible in the build container):
  * knn_indices            <- data_gen/get_neighbors.py:52-71 imported and run           (pinned)
  * uv_index_map           <- xiuminglib/img.py:289-431 imported and run, with cv2's
                              distanceTransform(DIST_L1) substituted by SciPy's taxicab
                              chamfer transform (cv2 is not installable here)             (pinned up to that substitution)
  * normalize_uint / denormalize_float <- xiuminglib/img.py:11-54 imported and run       (pinned)
  * view/light cosines, diffuse base, remap, cv2.resize: the reference needs Blender's mathutils / cv2
    (absent) -> restated from the source lines cited below                               (parity unpinned)
"""
import numpy as np


# ----------------------------------------------------------------------------
# third_party/xiuminglib/xiuminglib/img.py:11-54
# ----------------------------------------------------------------------------
def normalize_uint(arr):
    """img.py:11-29: uint8/uint16 -> float64 / dtype max."""
    if arr.dtype not in (np.uint8, np.uint16):
        raise TypeError(arr.dtype)
    return arr.astype(float) / np.iinfo(arr.dtype).max


def denormalize_float(arr, uint_type='uint8'):
    """img.py:32-54: float in [0,1] -> uint by TRUNCATION of arr * max (no rounding)."""
    if arr.min() < 0 or arr.max() > 1:
        raise ValueError("values outside [0, 1]")
    return (arr * np.iinfo(uint_type).max).astype(uint_type)


# ----------------------------------------------------------------------------
# third_party/xiuminglib/xiuminglib/img.py:88-118 (xm.img.resize -> cv2.resize, default INTER_LINEAR), as `_load_data`
# applies it to the NORMALISED float64 buffers when the stored resolution differs from uvh / (imh, imw)
# (nlt/datasets/nlt.py:138-146,162-170).
# ----------------------------------------------------------------------------
def resize_target(h, w, new_h=None, new_w=None):
    """img.py:103-116: the missing side keeps the aspect ratio, truncated."""
    if new_h is None and new_w is None:
        raise ValueError("At least one of new height or width must be given")
    if new_h is None:
        new_h = int(h / w * new_w)
    elif new_w is None:
        new_w = int(w / h * new_h)
    return new_h, new_w


def _cv_linear_taps(dst, src):
    """OpenCV resize(), INTER_LINEAR, per axis (imgproc/src/resize.cpp, the xofs / alpha table loop): source index of
    the first tap and the weight of the second one.  fx is formed in float32, out-of-range taps are clamped with
    weight 0 (left / top edge) or collapse onto the last sample (right / bottom edge)."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    lo = sx < 0
    sx = np.where(lo, 0, sx); fx = np.where(lo, np.float32(0), fx)
    hi = sx >= src - 1
    sx = np.where(hi, src - 1, sx); fx = np.where(hi, np.float32(0), fx)
    return sx, fx.astype(np.float32), hi


def cv_resize_linear(arr, new_h, new_w):
    """cv2.resize(arr, (new_w, new_h)) for a float64 array [h,w] or [h,w,c]: horizontal pass then vertical pass, tap
    weights held in float32 (OpenCV's `AT` for CV_64F), sums in float64, one IEEE operation at a time."""
    a = np.asarray(arr, dtype=np.float64)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    h, w = a.shape[:2]
    if (new_h, new_w) == (h, w):
        return np.asarray(arr, dtype=np.float64).copy()             # cv2 copies when the size is unchanged
    sx, fx, xhi = _cv_linear_taps(new_w, w)
    sy, fy, yhi = _cv_linear_taps(new_h, h)
    a0 = (np.float32(1) - fx).astype(np.float64)[None, :, None]
    a1 = fx.astype(np.float64)[None, :, None]
    x1 = np.minimum(sx + 1, w - 1)
    rows = a[:, sx, :] * a0 + a[:, x1, :] * a1                         # HResizeLinear
    rows = np.where(xhi[None, :, None], a[:, sx, :], rows)             # dx >= xmax: D = S[sx] * 1
    b0 = (np.float32(1) - fy).astype(np.float64)[:, None, None]
    b1 = fy.astype(np.float64)[:, None, None]
    y1 = np.minimum(sy + 1, h - 1)
    out = b0 * rows[sy] + b1 * rows[y1]                                # VResizeLinear
    return out[:, :, 0] if squeeze else out


# ----------------------------------------------------------------------------
# data_gen/render.py:209-228 (view cosines), :231-276 (light cosines), :164,170 (quantisation)
# ----------------------------------------------------------------------------
def _dot3(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _normalized(v):
    """mathutils.Vector.normalized(): v / |v| (zero vector stays zero)."""
    n = np.sqrt(_dot3(v, v))
    safe = np.where(n > 0, n, 1.0)
    return v / safe[..., None]


def cosine_map(src_loc, locs, normals, valid, occluded=None):
    """cos = <normalize(src_loc - p), normalize(n)> at every camera pixel that hit the object
    (`valid`: loc is not None and the hit object is the subject, render.py:219-220,267-268);
    0 elsewhere and, for the light, at pixels whose shadow ray is blocked (render.py:270-271).
    locs, normals [Hc,Wc,3] float64; valid / occluded [Hc,Wc] bool.  Returns float64 [Hc,Wc]."""
    src = np.asarray(src_loc, np.float64).reshape(1, 1, 3)
    p2s = _normalized(src - locs)
    cos = _dot3(p2s, _normalized(normals))
    keep = valid.astype(bool)
    if occluded is not None:
        keep = keep & ~occluded.astype(bool)
    return np.where(keep, cos, 0.0)


def quantize_unit(x):
    """render.py:164,170: denormalize_float(np.clip(x, 0, 1)) -> uint8 (truncating)."""
    return denormalize_float(np.clip(x, 0, 1))


# ----------------------------------------------------------------------------
# data_gen/postproc.py:53-76 (albedo, diffuse base)
# ----------------------------------------------------------------------------
def albedo_from_frames(rgb_u8_frames):
    """postproc.py:53-64: albedo = (sum over frames of rgb/255) / max of that sum.
    rgb_u8_frames [F,H,W,3] uint8 -> float64 [H,W,3].  Frames are added in order."""
    rgb_sum = np.zeros(rgb_u8_frames.shape[1:], np.float64)
    for f in range(rgb_u8_frames.shape[0]):
        rgb_sum += normalize_uint(rgb_u8_frames[f])
    return rgb_sum / rgb_sum.max()


def diffuse_base(albedo, lvis_u8):
    """postproc.py:66-76: diffuse = albedo * (lvis/255) per channel, clipped, truncated to uint8."""
    lvis = normalize_uint(lvis_u8)
    d = albedo * lvis[..., None]
    return (np.clip(d, 0, 1) * 255).astype(np.uint8)


# ----------------------------------------------------------------------------
# data_gen/util.py:45-58 remap = cv2.remap(src, map*w, map*h, INTER_LINEAR), src[0,0] = 0
# ----------------------------------------------------------------------------
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS            # cv2: coordinates are quantised to 1/32 pixel
INTER_REMAP_COEF_BITS = 15
INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS


def _remap_coords(mapping, h, w):
    """util.py:47-50: (mapping * size).astype(float32); then cv2's cvRound(x * 32) (round half to
    even) split into the integer texel and the 5-bit fraction."""
    mx = (mapping[..., 0] * w).astype(np.float32)
    my = (mapping[..., 1] * h).astype(np.float32)
    sx = np.rint(mx.astype(np.float64) * INTER_TAB_SIZE)       # float32 * 32 is exact in float64
    sy = np.rint(my.astype(np.float64) * INTER_TAB_SIZE)
    sx = np.clip(sx, -2 ** 31, 2 ** 31 - 1).astype(np.int64)
    sy = np.clip(sy, -2 ** 31, 2 ** 31 - 1).astype(np.int64)
    ix = np.clip(sx >> INTER_BITS, -32768, 32767)              # saturate_cast<short>
    iy = np.clip(sy >> INTER_BITS, -32768, 32767)
    return ix, iy, (sx & (INTER_TAB_SIZE - 1)), (sy & (INTER_TAB_SIZE - 1))


def _taps(src, ix, iy):
    """The four bilinear taps with BORDER_CONSTANT = 0 outside the image."""
    h, w = src.shape[:2]
    out = []
    for dy in (0, 1):
        for dx in (0, 1):
            x, y = ix + dx, iy + dy
            ok = (x >= 0) & (x < w) & (y >= 0) & (y < h)
            v = src[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1)]
            ok = ok if src.ndim == 2 else ok[..., None]
            out.append(np.where(ok, v, 0))
    return out


def remap_u8(src_u8, mapping, force_kbg=True):
    """data_gen/util.py:45-58 for an 8-bit source (cvis/lvis/rgb, render.py:174-176; diffuse,
    postproc.py:80): cv2's fixed-point bilinear -- weights = round(w * 2^15) from the 32x32
    fraction table (exact for bilinear), result = (sum + 2^14) >> 15.
    src [h,w] or [h,w,C] uint8; mapping [H,W,>=2] in [0,1] (x first).  -> uint8 [H,W(,C)]."""
    src = src_u8.copy()
    if force_kbg:
        src[0, 0, ...] = 0                                     # util.py:53-55
    h, w = src.shape[:2]
    ix, iy, fx, fy = _remap_coords(mapping, h, w)
    w00 = (INTER_TAB_SIZE - fy) * (INTER_TAB_SIZE - fx) * (INTER_REMAP_COEF_SCALE >> (2 * INTER_BITS))
    w01 = (INTER_TAB_SIZE - fy) * fx * (INTER_REMAP_COEF_SCALE >> (2 * INTER_BITS))
    w10 = fy * (INTER_TAB_SIZE - fx) * (INTER_REMAP_COEF_SCALE >> (2 * INTER_BITS))
    w11 = fy * fx * (INTER_REMAP_COEF_SCALE >> (2 * INTER_BITS))
    t00, t01, t10, t11 = [t.astype(np.int64) for t in _taps(src, ix, iy)]
    ex = (lambda a: a) if src.ndim == 2 else (lambda a: a[..., None])
    acc = t00 * ex(w00) + t01 * ex(w01) + t10 * ex(w10) + t11 * ex(w11)
    return ((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS).astype(np.uint8)


def remap_f32(src_f32, mapping, force_kbg=True):
    """Same call on a float32 source: cv2 keeps the 1/32-pixel coordinate quantisation and uses the
    float table w = (1-fy)(1-fx) ... ; accumulated left to right in float32."""
    src = src_f32.astype(np.float32).copy()
    if force_kbg:
        src[0, 0, ...] = 0
    h, w = src.shape[:2]
    ix, iy, fx, fy = _remap_coords(mapping, h, w)
    s = np.float32(1.0 / INTER_TAB_SIZE)
    ax, ay = fx.astype(np.float32) * s, fy.astype(np.float32) * s
    one = np.float32(1)
    w00, w01, w10, w11 = (one - ay) * (one - ax), (one - ay) * ax, ay * (one - ax), ay * ax
    t00, t01, t10, t11 = _taps(src, ix, iy)
    ex = (lambda a: a) if src.ndim == 2 else (lambda a: a[..., None])
    acc = t00 * ex(w00)
    acc = acc + t01 * ex(w01)
    acc = acc + t10 * ex(w10)
    acc = acc + t11 * ex(w11)
    return acc.astype(np.float32)


# ----------------------------------------------------------------------------
# third_party/xiuminglib/xiuminglib/img.py:289-431 grid_query_unstruct
#   (method griddata / nearest, max_l1_interp) as used by data_gen/render.py:279-351
# ----------------------------------------------------------------------------
def occupancy_indices(uvs, h, w):
    """img.py:389-393: the INTEGER row/column a sample lands on (truncation toward zero) and the
    in-canvas flag.  uvs [P,2] float64 (u right, v up)."""
    ri = ((1 - uvs[:, 1]) * (h - 1)).astype(int)
    ci = (uvs[:, 0] * (w - 1)).astype(int)
    ok = (ri >= 0) & (ri < h) & (ci >= 0) & (ci < w)
    return ri, ci, ok


def l1_distance_to_occupied(has_value):
    """cv2.distanceTransform(1 - has_value, DIST_L1, 3) (img.py:394): exact city-block distance to
    the nearest occupied pixel; two-pass chamfer.  A canvas with no occupied pixel gives 'far'."""
    h, w = has_value.shape
    far = h + w + 1
    d = np.where(has_value > 0, 0, far).astype(np.int64)
    for i in range(h):
        for j in range(w):
            if i > 0:
                d[i, j] = min(d[i, j], d[i - 1, j] + 1)
            if j > 0:
                d[i, j] = min(d[i, j], d[i, j - 1] + 1)
    for i in range(h - 1, -1, -1):
        for j in range(w - 1, -1, -1):
            if i < h - 1:
                d[i, j] = min(d[i, j], d[i + 1, j] + 1)
            if j < w - 1:
                d[i, j] = min(d[i, j], d[i, j + 1] + 1)
    return d


def uv_index_map(uvs, values, grid_res, max_l1_interp=4, fill_value=0.0, return_index=False):
    """grid_query_unstruct(uvs, values, (h,w), {'func':'griddata','func_underlying':'nearest',
    'fill_value':(0,), 'max_l1_interp':4}) (render.py:326-348).

    Every grid texel (i,j) at (u,v) = (j/(w-1), 1 - i/(h-1)) takes the value of the sample nearest
    in (u,v) (Euclidean; ties -> lowest sample index, SciPy's KD-tree leaves ties unspecified) if
    its L1 distance to an occupied texel is <= max_l1_interp, else fill_value.
    Returns float64 [h,w,M] (and the int32 sample index map, -1 = filled, when return_index)."""
    values = np.asarray(values, np.float64)
    if values.ndim == 1:
        values = values[:, None]
    h, w = grid_res
    ri, ci, ok = occupancy_indices(uvs, h, w)
    has_value = np.zeros((h, w), np.uint8)
    has_value[ri[ok], ci[ok]] = 1
    trusted = l1_distance_to_occupied(has_value) <= max_l1_interp
    gu = np.linspace(0, 1, w)
    gv = 1 - np.linspace(0, 1, h)
    idx = np.full((h, w), -1, np.int32)
    for i in range(h):
        du = gu[None, :] - uvs[:, 0:1]                       # [P,w]
        dv = gv[i] - uvs[:, 1:2]                             # [P,1]
        d2 = du * du + dv * dv
        idx[i] = np.argmin(d2, axis=0)                       # first minimum = lowest index
    idx[~trusted] = -1
    out = np.where((idx >= 0)[..., None], values[np.maximum(idx, 0)], fill_value)
    return (out, idx) if return_index else out


# ----------------------------------------------------------------------------
# data_gen/get_neighbors.py:52-71, generalised from 1 to k neighbours
# ----------------------------------------------------------------------------
def knn_indices(ref_pos, cand_pos, k=1):
    """For every reference position the indices of the k nearest candidates with NON-ZERO
    distance, nearest first; equal distances keep candidate order (strict `<`, first wins,
    get_neighbors.py:63-65).  Distances are compared squared, (dx^2 + dy^2) + dz^2 in float64.
    Returns int32 [P,k]; -1 where fewer than k candidates qualify."""
    ref = np.asarray(ref_pos, np.float64); cand = np.asarray(cand_pos, np.float64)
    out = np.full((ref.shape[0], k), -1, np.int32)
    for p in range(ref.shape[0]):
        d = ref[p] - cand
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        order = np.argsort(d2, kind='stable')
        order = order[d2[order] != 0][:k]
        out[p, :len(order)] = order
    return out


# ----------------------------------------------------------------------------
# nlt/datasets/nlt.py:115-184 _load_data, on frames already decoded to uint8
# ----------------------------------------------------------------------------
def assemble_batch(store, ids, nn_ids, mode='train'):
    """The float32 buffers `_load_data` returns for frames `ids` [N] with neighbours `nn_ids`
    [N,k] (-1 = missing neighbour -> zeros, nlt.py:152-157), taken from a resident uint8 store
    {'diffuse','rgb' [F,H,W,3], 'cvis','lvis' [F,H,W]} at the stored resolution (cv2 resize is the
    identity then, nlt.py:139-146).  uint8 -> float64 / 255 -> float32 (nlt.py:131-136,173-181)."""
    f32 = lambda a: normalize_uint(a).astype(np.float32)
    ids = np.asarray(ids); nn_ids = np.asarray(nn_ids)
    base = f32(store['diffuse'][ids])
    cvis = f32(store['cvis'][ids])[..., None]
    lvis = f32(store['lvis'][ids])[..., None]
    rgb = np.zeros_like(base) if mode == 'test' else f32(store['rgb'][ids])      # nlt.py:126-128
    ok = (nn_ids >= 0)[..., None, None, None]
    safe = np.maximum(nn_ids, 0)
    nn_base = np.where(ok, f32(store['diffuse'][safe]), np.float32(0))
    nn_rgb = np.where(ok, f32(store['rgb'][safe]), np.float32(0))
    return {'base': base, 'cvis': cvis, 'lvis': lvis, 'rgb': rgb, 'nn_base': nn_base, 'nn_rgb': nn_rgb}
