"""On-disk capture format: a tiny capture written the way data_gen/render.py + postproc.py lay it
out (per-sample directories of PNGs, uv2cam.npy as float16, nn.json, the <root>.json index with relative paths and a
`complete` flag) is read back by nlt_amd.datasets.nlt.load_store, and batches assembled from it equal
oracle/buffers.assemble_batch (the C-ABI adapters are the CPU emulation of tests/fake_capi.py)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

import fake_capi
import nlt_amd
from nlt_amd.datasets import nlt as D
from oracle import buffers as B


def write_capture(root, uv=16, im=8, seed=0):
    rng = np.random.default_rng(seed)
    cams, lights = ['P01', 'P02'], ['L1', 'L2']
    ids = ['trainvali_%09d_%s_%s' % (i, c, l) for i, (c, l) in enumerate((c, l) for c in cams for l in lights)] + ['test_000000000_P09_L9']
    data = {}
    index = {}
    for n, id_ in enumerate(ids):
        d = os.path.join(root, id_)
        os.makedirs(d)
        U8 = lambda *s: rng.integers(0, 256, s, dtype=np.uint8)
        a = {'diffuse': U8(uv, uv, 3), 'cvis': U8(uv, uv), 'lvis': U8(uv, uv), 'uv2cam': rng.random((im, im, 2)).astype(np.float16)}
        entry = {k: os.path.join(id_, k + ('.npy' if k == 'uv2cam' else '.png')) for k in a}
        if id_.startswith('trainvali_'):
            a['rgb'] = U8(uv, uv, 3); a['rgb_camspc'] = np.dstack((U8(im, im, 3), np.full((im, im), 255, np.uint8)))   # RGBA on disk
            entry['rgb'] = os.path.join(id_, 'rgb.png'); entry['rgb_camspc'] = os.path.join(id_, 'rgb_camspc.png')
        for k, v in a.items():
            if k == 'uv2cam':
                np.save(os.path.join(d, 'uv2cam.npy'), v)
            else:
                Image.fromarray(v).save(os.path.join(d, k + '.png'))
        with open(os.path.join(d, 'nn.json'), 'w') as h:
            json.dump({'cam': 'P02', 'light': 'L1'}, h)
        entry['nn'] = os.path.join(id_, 'nn.json')
        entry['complete'] = n != 1                                  # one incomplete sample: skipped by _glob
        index[id_] = entry
        data[id_] = a
    with open(root.rstrip('/') + '.json', 'w') as h:
        json.dump(index, h)
    return ids, data


def test_load_store_and_batches(tmp_path, monkeypatch):
    fake_capi.install(monkeypatch)
    root = str(tmp_path / 'capture')
    ids, data = write_capture(root)
    store = D.load_store(root, device='cpu')
    assert store['ids'] == sorted(ids) and store['diffuse'].dtype == torch.uint8 and store['uv2cam'].dtype == torch.float16
    i = store['ids'].index(ids[0])
    assert np.array_equal(store['rgb_camspc'][i].numpy(), data[ids[0]]['rgb_camspc'][:, :, :3])      # alpha dropped
    assert not store['rgb'][store['ids'].index('test_000000000_P09_L9')].any()                        # test sample: no rgb
    cfg = nlt_amd.make_config(data_root=root, uvh=16, uvw=16, imh=8, imw=8, holdout_cam='P02', holdout_light='L2', bs=2)
    ds = D.Dataset(cfg, 'train', device='cpu')
    assert ids[1] not in ds.files and ids[3] not in ds.files and len(ds.files) == 2                    # incomplete / held out
    b = ds.load_batch(ds.files)
    nn_id = ids[2]                                                   # trainvali_..._P02_L1
    st = {k: store[k].numpy() for k in ('diffuse', 'rgb', 'cvis', 'lvis')}
    fid = [store['ids'].index(x) for x in ds.files]
    ref = B.assemble_batch(st, fid, [[store['ids'].index(nn_id)]] * 2)
    assert np.array_equal(b[1].numpy(), ref['base']) and np.array_equal(b[9].numpy(), ref['nn_rgb'])
    assert b[4].dtype == torch.float32 and np.array_equal(b[4].numpy(), store['uv2cam'][fid].float().numpy())
    assert b[7] == [nn_id, nn_id]
    with pytest.raises(NotImplementedError):
        D.Dataset(nlt_amd.make_config(data_root=root, uvh=32), 'train', device='cpu')
    with pytest.raises(FileNotFoundError):
        D.load_store(str(tmp_path / 'missing'), device='cpu')
    with pytest.raises(ValueError):
        D.Dataset(cfg, 'bogus', store=store)
