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
    # another training resolution than the stored one: every buffer normalised and resized as `_load_data` does per sample
    # (nlt.py:131-146: normalize_uint -> cv2.resize INTER_LINEAR on float64 -> float32), warp left alone
    cfg2 = nlt_amd.make_config(data_root=root, uvh=24, uvw=24, imh=12, imw=12, holdout_cam='P02', holdout_light='L2', bs=2)
    ds2 = D.Dataset(cfg2, 'train', device='cpu')
    b2 = ds2.load_batch(ds2.files)
    rs = lambda a, h, w: B.cv_resize_linear(B.normalize_uint(a), h, w).astype(np.float32)
    for j, id_ in enumerate(ds2.files):
        assert np.array_equal(b2[1][j].numpy(), rs(data[id_]['diffuse'], 24, 24))
        assert np.array_equal(b2[2][j].numpy()[..., 0], rs(data[id_]['cvis'], 24, 24))
        assert np.array_equal(b2[5][j].numpy(), rs(data[id_]['rgb'], 24, 24))
        assert np.array_equal(b2[6][j].numpy(), rs(data[id_]['rgb_camspc'][:, :, :3], 12, 12))
        assert np.array_equal(b2[8][j, 0].numpy(), rs(data[nn_id]['diffuse'], 24, 24))
        assert np.array_equal(b2[10][j].numpy(), rs(data[nn_id]['rgb_camspc'][:, :, :3], 12, 12))
    assert tuple(b2[4].shape) == (2, 8, 8, 2)                        # "always warp first and then resize" (nlt.py:147-148)
    with pytest.raises(FileNotFoundError):
        D.load_store(str(tmp_path / 'missing'), device='cpu')
    with pytest.raises(ValueError):
        D.Dataset(cfg, 'bogus', store=store)


def test_sixteen_bit_capture(tmp_path, monkeypatch):
    """16-bit PNGs (xm.io.img.load keeps the depth, normalize_uint divides by 65535: xiuminglib/img.py:11-29)."""
    fake_capi.install(monkeypatch)
    root = str(tmp_path / 'capture16')
    rng = np.random.default_rng(5)
    ids = ['trainvali_%09d_P01_L%d' % (i, i) for i in range(2)]
    index, data = {}, {}
    for id_ in ids:
        d = os.path.join(root, id_)
        os.makedirs(d)
        U16 = lambda *s: rng.integers(0, 65536, s).astype(np.uint16)
        a = {'cvis': U16(8, 8), 'lvis': U16(8, 8)}
        for k, v in a.items():
            Image.fromarray(v).save(os.path.join(d, k + '.png'))                 # mode I;16
        for k in ('diffuse', 'rgb', 'rgb_camspc'):                              # PIL writes no 16-bit RGB: grey, replicated by the reader
            a[k] = U16(8, 8)
            Image.fromarray(a[k]).save(os.path.join(d, k + '.png'))
        np.save(os.path.join(d, 'uv2cam.npy'), rng.random((8, 8, 2)).astype(np.float16))
        with open(os.path.join(d, 'nn.json'), 'w') as h:
            json.dump({'cam': 'P01', 'light': 'L0'}, h)
        index[id_] = dict({k: os.path.join(id_, k + '.png') for k in a}, uv2cam=os.path.join(id_, 'uv2cam.npy'),
                          nn=os.path.join(id_, 'nn.json'), complete=True)
        data[id_] = a
    with open(root + '.json', 'w') as h:
        json.dump(index, h)
    ds = D.Dataset(nlt_amd.make_config(data_root=root, uvh=8, uvw=8, imh=8, imw=8, bs=2), 'train', device='cpu')
    assert ds.store['cvis'].dtype == torch.int32
    b = ds.load_batch(ids)
    f32 = lambda a: B.normalize_uint(a).astype(np.float32)
    for j, id_ in enumerate(ids):
        assert np.array_equal(b[2][j].numpy()[..., 0], f32(data[id_]['cvis']))
        assert np.array_equal(b[1][j].numpy(), np.dstack([f32(data[id_]['diffuse'])] * 3))
        assert np.array_equal(b[8][j, 0].numpy(), np.dstack([f32(data[ids[0]]['diffuse'])] * 3))
