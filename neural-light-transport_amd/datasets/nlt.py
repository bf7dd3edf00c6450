"""nlt/datasets/nlt.py on a uint8 frame store resident in HBM.

The reference decodes six PNGs per sample on tf.data worker threads every step (cache = False in
the released config).  With 288 GB of HBM the whole dragon capture fits on the device as uint8, so
a batch is assembled by ONE gather/convert pass (nlt_assemble_batch): frame ids in, float32 texel
buffers out, bit-identical to `_load_data`'s uint8 -> float64/255 -> float32.
`load_store` reads the reference's on-disk capture (the `<data_root>.json` index of data_gen/postproc.py:89-122,
per-sample PNGs, `uv2cam.npy` fp16 maps, `nn.json`) once into that store."""
import json
import re
from itertools import product
from os.path import exists, join

import numpy as np
import torch

from .. import _capi as C


def read_png(path):
    """xm.io.img.load(path, as_array=True) (xiuminglib/io/img.py:12-33): the decoded PIL image as it is, any mode / depth."""
    from PIL import Image
    with open(path, 'rb') as h:
        img = Image.open(h)
        img.load()
    return np.array(img)


def load_store(data_root, device='cuda', ids=None):
    """Decodes a capture laid out as the reference writes it (data_gen/render.py:196-206, postproc.py:66-122) into
    the resident uint8 store.  Paths in `<data_root>.json` are relative to data_root (nlt/datasets/nlt.py:36-45).
    Test samples have no rgb / rgb_camspc (postproc.py:104-107): their slots stay zero."""
    status = data_root.rstrip('/') + '.json'
    if not exists(status):
        raise FileNotFoundError(("Data status JSON not found at \n\t%s\nRun "
                                 "$REPO/data_gen/postproc.py to generate it") % status)
    with open(status) as h:
        paths = json.load(h)
    ids = sorted(paths) if ids is None else list(ids)

    def png(path, channels):
        a = read_png(path)
        if a.dtype != np.uint8:                                      # 16-bit PNG: PIL modes I;16 / I;16B / I
            if a.dtype.kind in 'ui' and a.dtype.itemsize in (2, 4) and a.min() >= 0 and a.max() <= 65535:
                a = a.astype(np.uint16)
            else:                                                    # xm.img.normalize_uint takes uint8 / uint16 only (img.py:11-29)
                raise NotImplementedError("%s: %s PNGs (normalize_uint takes uint8 / uint16)" % (path, a.dtype))
        if channels == 3:
            if a.ndim == 2:
                a = np.dstack([a] * 3)
            return a[:, :, :3]                                   # [:, :, :3] as nlt/datasets/nlt.py:121,128-130
        return a if a.ndim == 2 else a[:, :, 0]

    cols = {k: [] for k in ('diffuse', 'rgb', 'cvis', 'lvis', 'rgb_camspc', 'uv2cam')}
    nn, complete = {}, []
    for id_ in ids:
        p = {k: (v if k == 'complete' else join(data_root, v)) for k, v in paths[id_].items()}
        complete.append(bool(p['complete']))
        if not p['complete']:
            for k in cols:
                cols[k].append(None)
            continue
        cols['diffuse'].append(png(p['diffuse'], 3))
        cols['cvis'].append(png(p['cvis'], 1))
        cols['lvis'].append(png(p['lvis'], 1))
        cols['uv2cam'].append(np.load(p['uv2cam']))
        cols['rgb'].append(png(p['rgb'], 3) if 'rgb' in p else None)
        cols['rgb_camspc'].append(png(p['rgb_camspc'], 3) if 'rgb_camspc' in p else None)
        with open(p['nn']) as h:
            nn[id_] = json.load(h)

    def stack(key, dtype):
        ref = next((a for a in cols[key] if a is not None), None)
        if ref is None:
            raise ValueError("no complete sample provides '%s'" % key)
        for a in cols[key]:
            if a is not None and a.shape != ref.shape:
                raise NotImplementedError("'%s' comes in several resolutions (%s vs %s): the reference resizes with cv2"
                                          % (key, a.shape, ref.shape))
        if dtype is None:                                            # texel buffers: uint8 stays uint8 (the resident fast path);
            depths = {a.dtype for a in cols[key] if a is not None}   # 16-bit captures are held as int32 (no CUDA uint16 arithmetic)
            if len(depths) > 1:
                raise NotImplementedError("'%s' mixes 8- and 16-bit PNGs" % key)
            dtype = np.uint8 if depths == {np.dtype(np.uint8)} else np.int32
        out = np.zeros((len(ids),) + ref.shape, dtype)
        for i, a in enumerate(cols[key]):
            if a is not None:
                out[i] = a
        return torch.from_numpy(out).to(device)

    store = {'ids': ids, 'nn': nn, 'complete': complete}
    for key in ('diffuse', 'rgb', 'cvis', 'lvis', 'rgb_camspc'):
        store[key] = stack(key, None)
    store['uv2cam'] = stack('uv2cam', np.float16)
    return store


class ResidentTexels:
    """The UV-space texel buffers of a batch (base, cvis, lvis, rgb, nn_base, nn_rgb of the model's 11-tuple) still in
    the resident uint8 store: frame ids instead of float tensors.  `Model.call` hands this to the fused front kernel,
    which converts uint8 -> float64 / 255 -> float32 (`_load_data`, nlt/datasets/nlt.py:131-136,173-181) in registers;
    any consumer that needs the float tensors asks for them (`materialize`, bit-identical to the eager batch)."""

    def __init__(self, store, ids, nn_ids, test_mode=False):
        self.diffuse, self.rgb, self.cvis, self.lvis = store['diffuse'], store['rgb'], store['cvis'], store['lvis']
        self.uv2cam = store['uv2cam']                                # fp16 [F,hc,wc,2]: the warp reads it in place too
        self.ids, self.nn_ids, self.test_mode = ids, nn_ids, test_mode
        self.n, self.k = ids.numel(), nn_ids.shape[1]
        self.h, self.w = self.cvis.shape[1:3]
        self.hc, self.wc = self.uv2cam.shape[1:3]
        self._float, self._base, self._warp = None, None, None

    def warp_float(self):
        """warp [n,hc,wc,2] float32 (fp16 `.npy` -> float32, never resized: nlt/datasets/nlt.py:125,147-148,176)."""
        if self._warp is None:
            self._warp = self.uv2cam[self.ids.long()].float()
        return self._warp

    def key(self):
        return tuple(t.data_ptr() for t in (self.diffuse, self.rgb, self.cvis, self.lvis, self.ids, self.nn_ids)) + (self.n, self.k)

    def base_float(self, out=None):
        """base [n,h,w,3] float32 alone (the UV -> camera warp of Model.call needs it).  With a caller-owned `out`
        (a buffer the caller shares between batches) the gather is redone every call and nothing is cached here: a
        cached alias would silently show whatever batch filled that buffer last."""
        if self._float is not None:
            return self._float['base']
        if out is not None:
            return C.gather_frames_u8(self.diffuse, self.ids, out=out)
        if self._base is None:
            self._base = C.gather_frames_u8(self.diffuse, self.ids)
        return self._base

    def materialize(self):
        """All six float buffers, as Dataset.load_batch(resident=False) returns them."""
        if self._float is None:
            self._float = C.assemble_batch(self.diffuse, self.rgb, self.cvis, self.lvis, self.ids, self.nn_ids,
                                           test_mode=self.test_mode)
            self._float['warp'] = self.warp_float()
        return self._float


class Dataset:
    """store: {'ids': [str], 'nn': {id: {'cam','light'}}, 'diffuse','rgb' [F,H,W,3] uint8 CUDA,
    'cvis','lvis' [F,H,W] uint8, 'uv2cam' [F,imh,imw,2] fp16, 'rgb_camspc' [F,imh,imw,3] uint8,
    'complete': [bool]}.  ids follow the reference's '{trainvali|test}_{i:09d}_{cam}_{light}'."""

    def __init__(self, config, mode, store=None, k=1, device='cuda', ring=0):
        if mode not in ('train', 'vali', 'test'):
            raise ValueError("Invalid mode: {provided}. Allowed modes: {allowed}".format(
                provided=mode, allowed=('train', 'vali', 'test')))
        if store is None:                                           # nlt/datasets/nlt.py:35-45
            store = load_store(config.get('DEFAULT', 'data_root'), device)
        self.config, self.mode, self.store, self.k = config, mode, store, k
        # Staging ring: load_batch fills one of `ring` persistent buffer sets instead of allocating ~20 fresh tensors per
        # step, so the addresses a batch arrives at repeat every `ring` steps and the model's recorded launch tape (keyed
        # by input addresses) keeps replaying in a real data loop.  A returned batch stays valid for `ring - 1` further
        # load_batch calls.  OPT-IN (bench.py, a training loop that consumes each batch before asking for the next): the
        # default ring = 0 hands out fresh tensors every call, as the reference's tf.data pipeline does, so a vali / vis loop
        # may keep any number of batches (and the `to_vis` tensors Model.call builds from them).
        self.ring, self._slots, self._turn = int(ring), {}, 0
        self._fstore = self._resized_store()
        self.index = {id_: i for i, id_ in enumerate(store['ids'])}
        self.bs = 1 if mode == 'test' else config.getint('DEFAULT', 'bs')     # datasets/base.py
        self.files = self._glob()
        assert self.files, "No files to process into a dataset"           # nlt/datasets/base.py:38

    TEXEL_KEYS = ('diffuse', 'rgb', 'cvis', 'lvis')

    def _resized_store(self):
        """The released capture is 8-bit at the resolution it is trained at: `_load_data`'s cv2.resize calls copy, and
        batches come straight out of the uint8 store.  For anything else -- another stored resolution than uvh /
        (imh, imw), 16-bit PNGs -- every buffer is normalised and resized ONCE here, exactly as `_load_data` would per
        sample (nlt.py:131-146: normalize_uint -> xm.img.resize = cv2 INTER_LINEAR on float64 -> float32;
        csrc/assemble.hip resize_cv_kernel), into a float32 store the batches are gathered from (4x the bytes, no
        resident-uint8 fast path).  Returns None in the native case."""
        s = self.store
        uvh = self.config.getint('DEFAULT', 'uvh', fallback=0) or s['cvis'].shape[1]
        h, w = s['cvis'].shape[1:3]
        uvw = int(w / h * uvh)                                       # xm.img.resize(arr, new_h=uvh): aspect kept, truncated
        imh = self.config.getint('DEFAULT', 'imh', fallback=0) or s['rgb_camspc'].shape[1]
        imw = self.config.getint('DEFAULT', 'imw', fallback=0) or s['rgb_camspc'].shape[2]
        native = all(s[k_].dtype == torch.uint8 for k_ in self.TEXEL_KEYS + ('rgb_camspc',)) and (h, w) == (uvh, uvw) \
            and tuple(s['rgb_camspc'].shape[1:3]) == (imh, imw)
        if native:
            return None
        fs = {}
        for k_ in self.TEXEL_KEYS + ('rgb_camspc',):
            src = s[k_] if s[k_].dim() == 4 else s[k_].unsqueeze(-1)
            oh, ow = (imh, imw) if k_ == 'rgb_camspc' else (uvh, uvw)
            fs[k_] = C.resize_cv_linear(src.contiguous(), oh, ow)
        return fs

    def _load_batch_resized(self, ids, fid_h, nn_h):
        fs, dev = self._fstore, self._fstore['cvis'].device
        li, nn = fid_h.long().to(dev), nn_h.long().to(dev)
        test = self.mode == 'test'
        ok = (nn >= 0).view(nn.shape + (1, 1, 1)).float()
        safe = nn.clamp(min=0)
        base, cvis, lvis = fs['diffuse'][li], fs['cvis'][li], fs['lvis'][li]
        rgb = torch.zeros_like(base) if test else fs['rgb'][li]
        rgb_c = torch.zeros((len(ids),) + tuple(fs['rgb_camspc'].shape[1:]), device=dev) if test else fs['rgb_camspc'][li]
        nn_base, nn_rgb = fs['diffuse'][safe] * ok, fs['rgb'][safe] * ok
        nn_rgb_c = fs['rgb_camspc'][safe[:, 0]] * ok[:, 0]
        warp = self.store['uv2cam'][li].float()                      # never resized (nlt.py:147-148)
        nn_names = [self.store['ids'][j] if j >= 0 else 'incomplete-data' for j in nn_h[:, 0].tolist()]
        return (list(ids), base, cvis, lvis, warp, rgb, rgb_c, nn_names, nn_base, nn_rgb, nn_rgb_c)

    def _glob(self):
        """nlt/datasets/nlt.py:54-86: hold-out split by camera x light."""
        g = lambda k: self.config.get('DEFAULT', k, fallback='').split(',')
        holdout = ['%s_%s' % x for x in product(g('holdout_cam'), g('holdout_light'))]
        complete = self.store.get('complete')
        ids = [id_ for i, id_ in enumerate(self.store['ids'])
               if id_.startswith('test' if self.mode == 'test' else 'trainvali') and (complete is None or complete[i])]
        if self.mode == 'test':
            return ids
        keep = []
        for id_ in ids:
            cam_light = '_'.join(id_.split('_')[-2:])
            if (self.mode == 'vali') == (cam_light in holdout):
                keep.append(id_)
        return keep

    def _get_nn_id(self, nn):
        """nlt/datasets/nlt.py:88-100."""
        rx = re.compile(r'trainvali_\d\d\d\d\d\d\d\d\d_{cam}_{light}'.format(**nn))
        matched = [x for x in self.store['ids'] if rx.search(x) is not None]
        if not matched:
            return None
        if len(matched) == 1:
            return matched[0]
        raise ValueError("Found {n} matches:\n\t{matches}".format(n=len(matched), matches=matched))

    def _nn_indices(self, id_):
        nns = self.store['nn'][id_]
        nns = nns if isinstance(nns, (list, tuple)) else [nns]
        out = []
        for nn in list(nns)[:self.k]:
            nn_id = self._get_nn_id(nn)
            out.append(-1 if nn_id is None else self.index[nn_id])       # missing neighbour -> zeros (:152-157)
        return out + [-1] * (self.k - len(out))

    def _slot(self, n):
        """The staging buffers the next batch of n samples goes into (None: allocate fresh ones)."""
        if self.ring <= 0:
            return None
        slots = self._slots.setdefault(n, [])
        i = self._turn % self.ring
        self._turn += 1
        while len(slots) <= i:
            slots.append({})
        return slots[i]

    def load_batch(self, ids, resident=False):
        """`_load_data` (nlt.py:115-184) for a list of sample ids, as the model's 11-tuple.
        resident=True leaves the six UV-space texel buffers in the uint8 store and the uv2cam map in its fp16 store:
        entry 1 (base) is a ResidentTexels and entries 2, 3, 4, 5, 8, 9 are None -- `Model.call` feeds the stores to the
        fused front kernel (29 B per texel read instead of 116 written + 116 read at k = 4) and to the warp, and
        materialises floats only where something asks for them (`ResidentTexels.materialize`: training, a per-frame obs_override;
        the inference mode -- one given map per level -- reads the stores in place too, nlt_front_ovr_forward_u8)."""
        s = self.store
        dev = s['cvis'].device
        n = len(ids)
        slot = self._slot(n)
        fid_l = [self.index[i] for i in ids]
        nn_l = [self._nn_indices(i) for i in ids]
        nn_h = torch.tensor(nn_l, dtype=torch.int32)
        if self._fstore is not None:
            return self._load_batch_resized(ids, torch.tensor(fid_l, dtype=torch.int32), nn_h)
        if slot is None:
            both = torch.tensor(fid_l + [x for row in nn_l for x in row], dtype=torch.int32).to(dev)
        else:
            # ONE host-to-device copy of (frame ids | neighbour ids) per batch, out of a pinned staging buffer of the slot
            if 'ids_dev' not in slot:
                slot['ids_dev'] = torch.empty(n + n * self.k, device=dev, dtype=torch.int32)
                slot['ids_pin'] = torch.empty(n + n * self.k, dtype=torch.int32, pin_memory=dev.type == 'cuda')
                slot['ids_ev'] = torch.cuda.Event() if dev.type == 'cuda' else None
            elif slot['ids_ev'] is not None:
                slot['ids_ev'].synchronize()                         # the slot's previous upload has left the pinned buffer
            pin = slot['ids_pin']
            pin[:n] = torch.tensor(fid_l, dtype=torch.int32)
            pin[n:] = nn_h.reshape(-1)
            both = slot['ids_dev']
            both.copy_(pin, non_blocking=True)
            if slot['ids_ev'] is not None:
                slot['ids_ev'].record()
        fid, nnid = both[:n], both[n:].view(n, self.k)
        test = self.mode == 'test'
        cam_shape = (n,) + tuple(s['rgb_camspc'].shape[1:])
        warp = None                                                  # (store-resident batches: the warp reads the fp16 store)
        if slot is None:
            if not resident:
                warp = s['uv2cam'][fid.long()].float()                           # never resized (nlt.py:147-148)
            rgb_c = torch.zeros(cam_shape, device=dev) if test else C.gather_frames_u8(s['rgb_camspc'], fid)   # nlt.py:126-128
            nn_rgb_c = C.gather_frames_u8(s['rgb_camspc'], nnid[:, 0].contiguous())
        else:
            if 'rgb_c' not in slot:
                slot['rgb_c'] = torch.zeros(cam_shape, device=dev)
                slot['nn_rgb_c'] = torch.empty(cam_shape, device=dev)
                slot['nn0'] = torch.empty(n, device=dev, dtype=torch.int32)
            rgb_c, nn_rgb_c = slot['rgb_c'], slot['nn_rgb_c']
            if not resident:
                if 'warp' not in slot:
                    slot['warp'] = torch.empty((n,) + tuple(s['uv2cam'].shape[1:]), device=dev, dtype=torch.float32)
                warp = slot['warp']
                warp.copy_(s['uv2cam'][fid.long()])                              # fp16 -> fp32 on the way in
            if not test:
                C.gather_frames_u8(s['rgb_camspc'], fid, out=rgb_c)
            slot['nn0'].copy_(nnid[:, 0])
            C.gather_frames_u8(s['rgb_camspc'], slot['nn0'], out=nn_rgb_c)
        nn_names = [s['ids'][j] if j >= 0 else 'incomplete-data' for j in nn_h[:, 0].tolist()]
        if resident:
            res = ResidentTexels(s, fid, nnid, test_mode=test)
            return (list(ids), res, None, None, None, None, rgb_c, nn_names, None, None, nn_rgb_c)
        b = C.assemble_batch(s['diffuse'], s['rgb'], s['cvis'], s['lvis'], fid, nnid, test_mode=test,
                             out=slot.get('texels') if slot is not None else None)
        if slot is not None:
            slot['texels'] = b
        return (list(ids), b['base'], b['cvis'], b['lvis'], warp, b['rgb'], rgb_c, nn_names,
                b['nn_base'], b['nn_rgb'], nn_rgb_c)
