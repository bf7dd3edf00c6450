"""nlt/datasets/nlt.py on a uint8 frame store resident in HBM.

The reference decodes six PNGs per sample on tf.data worker threads every step (cache = False in
the released config).  With 288 GB of HBM the whole dragon capture fits on the device as uint8, so
a batch is assembled by ONE gather/convert pass (nlt_assemble_batch): frame ids in, float32 texel
buffers out, bit-identical to `_load_data`'s uint8 -> float64/255 -> float32.
PNG / .npy file reading is not part of this class: `store` holds decoded arrays."""
import re
from itertools import product

import torch

from .. import _capi as C


class Dataset:
    """store: {'ids': [str], 'nn': {id: {'cam','light'}}, 'diffuse','rgb' [F,H,W,3] uint8 CUDA,
    'cvis','lvis' [F,H,W] uint8, 'uv2cam' [F,imh,imw,2] fp16, 'rgb_camspc' [F,imh,imw,3] uint8,
    'complete': [bool]}.  ids follow the reference's '{trainvali|test}_{i:09d}_{cam}_{light}'."""

    def __init__(self, config, mode, store, k=1):
        if mode not in ('train', 'vali', 'test'):
            raise ValueError(mode)
        self.config, self.mode, self.store, self.k = config, mode, store, k
        self.index = {id_: i for i, id_ in enumerate(store['ids'])}
        self.bs = 1 if mode == 'test' else config.getint('DEFAULT', 'bs')     # datasets/base.py
        self.files = self._glob()

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

    def load_batch(self, ids):
        """`_load_data` (nlt.py:115-184) for a list of sample ids, as the model's 11-tuple."""
        s = self.store
        dev = s['cvis'].device
        fid = torch.tensor([self.index[i] for i in ids], dtype=torch.int32, device=dev)
        nnid = torch.tensor([self._nn_indices(i) for i in ids], dtype=torch.int32, device=dev)
        b = C.assemble_batch(s['diffuse'], s['rgb'], s['cvis'], s['lvis'], fid, nnid, test_mode=self.mode == 'test')
        li = fid.long()
        warp = s['uv2cam'][li].float()                                           # never resized (nlt.py:147-148)
        if self.mode == 'test':
            rgb_c = torch.zeros((len(ids),) + tuple(s['rgb_camspc'].shape[1:]), device=dev)     # nlt.py:126-128
        else:
            rgb_c = C.gather_frames_u8(s['rgb_camspc'], fid)
        nn_rgb_c = C.gather_frames_u8(s['rgb_camspc'], nnid[:, 0].contiguous())
        nn_names = [s['ids'][j] if j >= 0 else 'incomplete-data' for j in nnid[:, 0].tolist()]
        return (list(ids), b['base'], b['cvis'], b['lvis'], warp, b['rgb'], rgb_c, nn_names,
                b['nn_base'], b['nn_rgb'], nn_rgb_c)

