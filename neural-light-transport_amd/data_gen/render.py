"""data_gen/render.py:209-351 on HIP: view / light cosine maps and the bidirectional UV <-> camera
mapping.  `intersect` is the dict render.py builds from xm.blender.camera.backproject_to_3d
(render.py:142-148), here in DENSE form over the imh x imw pixel grid (row-major, xys order):
    'locs', 'normals'  [P,3] float64 CUDA        'valid' [P] uint8 (loc is not None and the hit
    'face_i' [P] int64 (only for the mapping)     object is the subject, render.py:219,267,302)
    'occluded' [P] uint8 (only for the light: the result of the reference's BVH shadow rays,
                          render.py:238-254, which stay in Blender)
The reference's list-of-Vector form is converted with `dense_intersect`."""
import numpy as np
import torch

from .. import _capi as C


def dense_intersect(intersect, obj_name, device='cuda', occluded=None):
    """render.py's {'obj_names','locs','normals','face_i'} lists (None where the ray missed) -> dense."""
    n = len(intersect['locs'])
    valid = np.array([l is not None and o == obj_name for l, o in zip(intersect['locs'], intersect['obj_names'])])
    zero = (0.0, 0.0, 0.0)
    locs = np.array([tuple(l) if v else zero for l, v in zip(intersect['locs'], valid)], np.float64).reshape(n, 3)
    normals = np.array([tuple(x) if v else zero for x, v in zip(intersect['normals'], valid)], np.float64).reshape(n, 3)
    out = {'locs': torch.from_numpy(locs).to(device), 'normals': torch.from_numpy(normals).to(device),
           'valid': torch.from_numpy(valid.astype(np.uint8)).to(device)}
    if 'face_i' in intersect:
        out['face_i'] = np.array([-1 if (f is None or not v) else f for f, v in zip(intersect['face_i'], valid)], np.int64)
    if occluded is not None:
        out['occluded'] = torch.from_numpy(np.asarray(occluded, np.uint8)).to(device)
    return out


def _im_hw(xys):
    xys = np.asarray(xys)
    return int(xys[:, 1].max()) + 1, int(xys[:, 0].max()) + 1


def calc_view_cosines(cam_loc, xys, intersect, obj_name=None):
    """render.py:209-228 -> float64 [imh,imw]."""
    imh, imw = _im_hw(xys)
    cos, _ = C.cosine_map(intersect['locs'], intersect['normals'], intersect['valid'], None, cam_loc, want_u8=False)
    return cos.view(imh, imw)


def calc_light_cosines(light_loc, xys, cam_intersect, obj=None):
    """render.py:231-276 -> float64 [imh,imw]; cast shadows come in through cam_intersect['occluded']."""
    imh, imw = _im_hw(xys)
    cos, _ = C.cosine_map(cam_intersect['locs'], cam_intersect['normals'], cam_intersect['valid'],
                          cam_intersect.get('occluded'), light_loc, want_u8=False)
    return cos.view(imh, imw)


def cosines_to_uint8(src_loc, intersect, imh, imw, light=False):
    """render.py:162-171 in one launch: the cosine map clipped to [0,1] and TRUNCATED to uint8."""
    _, q = C.cosine_map(intersect['locs'], intersect['normals'], intersect['valid'],
                        intersect.get('occluded') if light else None, src_loc, want_float=False)
    return q.view(imh, imw)


def grid_query_unstruct(uvs, values, grid_res, method=None):
    """xiuminglib/img.py:289-431 for the one method NLT uses (griddata / nearest + max_l1_interp)."""
    method = method or {'func': 'griddata'}
    if method.get('func') != 'griddata' or method.get('func_underlying', 'linear') != 'nearest':
        raise NotImplementedError(method)
    fill = method.get('fill_value', (0,))
    if len(set(fill)) != 1:
        raise NotImplementedError("per-channel fill values")
    max_l1 = method.get('max_l1_interp', np.inf)
    if max_l1 is None or not np.isfinite(max_l1):
        raise NotImplementedError("unbounded max_l1_interp")
    if values.dim() == 1:
        values = values.view(-1, 1)
    h, w = grid_res
    out = C.uv_index_map(uvs.contiguous(), values.contiguous(), h, w, int(max_l1), float(fill[0]))
    return out[:, :, 0] if out.shape[2] == 1 else out


def calc_bidir_mapping(cached_unwrap, obj_name, xys, intersect, uvs, max_l1_interp=4):
    """render.py:279-351.  cached_unwrap: the {face: [[loop, vert, u, v], ...]} table of
    data_gen/uv_unwrap.py:53-74 (dict, or a path to its pickle).  Returns (uv2cam [imh,imw,2],
    cam2uv [uvs,uvs,2]) float64 in [0,1]."""
    if isinstance(cached_unwrap, str):
        import pickle
        with open(cached_unwrap, 'rb') as h:
            cached_unwrap = pickle.load(h)
    imh, imw = _im_hw(xys)
    xys = np.asarray(xys)
    face_i = np.asarray(intersect['face_i'])
    hit = np.nonzero(face_i >= 0)[0]
    cam_locs, uv_list = [], []
    for p in hit:                                           # pixel-major, then the face's vertices: the
        uv = np.asarray(cached_unwrap[int(face_i[p])])[:, 2:]      # reference's sample order (render.py:299-317)
        uv_list.append(uv)
        cam_locs.append(np.repeat(xys[p:p + 1].astype(np.float64), uv.shape[0], 0))
    uv = np.vstack(uv_list)
    xy = np.vstack(cam_locs)
    dev = intersect['valid'].device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(dev)
    method = {'func': 'griddata', 'func_underlying': 'nearest', 'fill_value': (0,), 'max_l1_interp': max_l1_interp}
    # UV -> camera: locations are camera pixels (v up), values the face-vertex UVs (y down)   render.py:305-309
    uv2cam = grid_query_unstruct(t(np.stack((xy[:, 0] / float(imw), 1 - xy[:, 1] / float(imh)), 1)),
                                 t(np.stack((uv[:, 0], 1 - uv[:, 1]), 1)), (imh, imw), method)
    # camera -> UV: locations are the face-vertex UVs, values the camera pixel (y down)       render.py:311-315
    cam2uv = grid_query_unstruct(t(uv), t(np.stack((xy[:, 0] / float(imw), xy[:, 1] / float(imh)), 1)),
                                 (uvs, uvs), method)
    return uv2cam, cam2uv
