"""CPU: `Model.vis_batch` / `compile_batch_vis` / `nlt_test.infer(..., outroot)` (reference nlt/models/nlt.py:207-342,
nlt/nlt_test.py:78-94).  The bytes of the PNGs are pinned to the reference's own helpers (tests/golden/vis.npz:
xiuminglib's linear2srgb + write_arr imported and run by tests/golden/make_vis_golden.py) and checked against
oracle/buffers.denormalize_float on a rendered batch; the model runs on the TEST-ONLY C-ABI emulation."""
import json
import os
import pickle
from os.path import exists, join

import warnings

import numpy as np
import pytest
import torch
from PIL import Image

from nlt_amd import metric, nlt_test
from nlt_amd.datasets.nlt import read_png
from nlt_amd.util import vis as V
from oracle import buffers as OB
from oracle import metric as OM
from oracle import nlt_oracle as O
import fake_capi
from test_host_orchestration import cpu_batch as _cpu_batch, make


def cpu_batch(batch, nn, tag='s'):
    """... with sample / neighbour ids the way a loader names them (the synthetic batches carry none)."""
    b = list(_cpu_batch(batch, nn))
    n = b[1].shape[0]
    if b[0] is None:
        b[0] = [('%s%03d' % (tag, i)).encode() for i in range(n)]          # bytes: what `x.numpy()` of a tf.string gives
    if b[7] is None:
        b[7] = ['nn_%s%03d' % (tag, i) for i in range(n)]
    return tuple(b)

GOLD = np.load(join(os.path.dirname(__file__), 'golden', 'vis.npz'))


def fake_psnr(monkeypatch):
    """metric.PSNR on CPU tensors: the sums from the oracle's float64 luma arithmetic."""
    from nlt_amd import _capi as C

    def psnr_sums(a, b, m=None):
        a, b = a.numpy().astype(np.float64), b.numpy().astype(np.float64)
        if a.ndim == 3 and a.shape[2] == 3:
            a, b = OM.rgb2lum(a), OM.rgb2lum(b)
        keep = np.ones(a.shape[:2], bool) if m is None else m.numpy().astype(bool)
        return torch.tensor([np.square(a.reshape(keep.shape)[keep] - b.reshape(keep.shape)[keep]).sum(), keep.sum()], dtype=torch.float64)
    monkeypatch.setattr(C, 'psnr_sums', psnr_sums)
    monkeypatch.setattr(metric, 'DEVICE', 'cpu')


@pytest.mark.parametrize('tag', ['lin', 'dark'])
def test_linear2srgb_and_write_arr_match_the_reference_bit_for_bit(tmp_path, tag):
    im = GOLD[tag]
    srgb = V.linear2srgb(im)
    assert srgb.dtype == np.float32 and np.array_equal(srgb.view(np.uint32), GOLD[tag + '_srgb'].view(np.uint32))
    for space, arr in (('srgb', srgb), ('raw', im)):
        p = str(tmp_path / ('%s.png' % space))
        ret = V.write_arr(arr, p)
        assert ret.dtype == np.uint8 and np.array_equal(ret, GOLD['%s_%s_u8' % (tag, space)])
        assert np.array_equal(read_png(p), ret)
        assert np.array_equal(ret, OB.denormalize_float(arr))
    assert im is not srgb and np.array_equal(im, GOLD[tag])                    # the input is left alone


def test_helpers_refuse_what_the_reference_refuses(tmp_path):
    with pytest.raises(ValueError):
        V.linear2srgb(np.full((2, 2, 3), 1.5, np.float32))
    with pytest.raises(ValueError):
        V.linear2srgb(np.zeros((2, 2), np.float32))
    with pytest.raises(ValueError):
        V.linear2srgb(np.zeros((2, 2, 4), np.float32))
    with pytest.raises(TypeError):
        V.linear2srgb(np.zeros((2, 2, 3), np.uint8))
    with pytest.raises(AssertionError):
        V.write_arr(np.full((2, 2, 3), -0.1, np.float32), str(tmp_path / 'x.png'))
    with pytest.raises(AssertionError):
        V.make_apng([np.zeros((4, 4, 3), np.float32)], outpath=str(tmp_path / 'x'))
    with pytest.raises(TypeError):
        V.make_apng([7], outpath=str(tmp_path / 'x'))
    assert V.to_str(b'abc') == 'abc' and V.to_str('x') == 'x' and V.to_str(np.array(b'q')) == 'q'
    one = (np.arange(16, dtype=np.uint8) * 16).reshape(4, 4, 1)                 # 1-channel images are written as grey RGB
    V.write_img(one, str(tmp_path / 'g.png'))
    assert np.array_equal(read_png(str(tmp_path / 'g.png')), np.dstack([one] * 3))


@pytest.mark.parametrize('linear', [False, True])
@pytest.mark.parametrize('mode', ['test', 'vali'])
def test_vis_batch_writes_the_reference_s_files(monkeypatch, tmp_path, mode, linear):
    fake_capi.install(monkeypatch)
    fake_psnr(monkeypatch)
    om, pm = make(256, 64, 64)
    pm.config.set('DEFAULT', 'linear_space', str(linear))
    batch, nn = O.synth_batch(2, 64, 64, 64, 64, 64, 64, k=1, seed=77)
    _, _, _, to_vis = pm.call(cpu_batch(batch, nn), mode)
    outdir = str(tmp_path / 'vis')
    raw = str(tmp_path / 'raw' / 'batch.pkl')
    pm.vis_batch(to_vis, outdir, mode, dump_raw_to=raw)
    names = ['base', 'pred', 'nn'] + ([] if mode == 'test' else ['gt'])
    for i in range(2):
        for name in names:
            src = np.clip(to_vis[name + '_camspc'][i].numpy(), 0, 1)
            want = OB.denormalize_float(V.linear2srgb(src) if linear else src)
            got = read_png(join(outdir, '%d_%s.png' % (i, name)))
            assert got.dtype == np.uint8 and got.shape == (64, 64, 3) and np.array_equal(got, want), (i, name)
        assert exists(join(outdir, '%d_gt.png' % i)) == (mode != 'test')
        with Image.open(join(outdir, '%d_base-vs-pred.apng' % i)) as a:
            assert a.format == 'PNG' and a.n_frames == 2 and a.size == (64, 64) and a.info['duration'] == 1000
        assert exists(join(outdir, '%d_gt-vs-pred.apng' % i)) == (mode != 'test')
        with open(join(outdir, '%d_metadata.json' % i)) as h:
            text = h.read()
        md = json.loads(text)
        assert md['id'] == V.to_str(to_vis['id'][i]) and md['nn_id'] == V.to_str(to_vis['nn_id'][i])
        assert text == json.dumps(md, indent=4, sort_keys=True)
        if mode == 'test':
            assert set(md) == {'id', 'nn_id'}
        else:                                                   # PSNR on the clipped LINEAR maps (nlt.py:258-268)
            gt, pred, base = (np.clip(to_vis[k][i].numpy(), 0, 1) for k in ('gt_camspc', 'pred_camspc', 'base_camspc'))
            assert md['pred_psnr'] == pytest.approx(OM.psnr(gt, pred), rel=1e-12)
            assert md['base_psnr'] == pytest.approx(OM.psnr(gt, base), rel=1e-12)
    with open(raw, 'rb') as h:
        back = pickle.load(h)
    assert set(back) == set(to_vis) and np.array_equal(back['pred_camspc'], to_vis['pred_camspc'].numpy())
    with pytest.raises(ValueError):
        pm.vis_batch(to_vis, outdir, 'eval')
    with pytest.raises(ValueError):                             # a synthetic batch without ids: said so, not a TypeError deep inside
        pm.vis_batch(dict(to_vis, id=None), outdir, mode)


def test_compile_batch_vis_webpage_and_frame_roll_up(monkeypatch, tmp_path):
    fake_capi.install(monkeypatch)
    fake_psnr(monkeypatch)
    om, pm = make(256, 64, 64)
    dirs = {'vali': [], 'test': []}
    ids = {}
    for mode in dirs:
        for b in range(2):
            batch, nn = O.synth_batch(2, 64, 64, 64, 64, 64, 64, k=1, seed=90 + b)
            batch = list(cpu_batch(batch, nn))
            batch[0] = ['%s_%03d' % (mode, 7 - 2 * b - j) for j in range(2)]          # ids NOT in batch order
            _, _, _, to_vis = pm.call(tuple(batch), mode)
            d = str(tmp_path / mode / ('batch%09d' % b))
            pm.vis_batch(to_vis, d, mode)
            dirs[mode].append(d)
            for j, id_ in enumerate(batch[0]):
                ids[id_] = join(d, '%d_pred.png' % j)
    link = pm.compile_batch_vis(dirs['vali'], str(tmp_path / 'vali_all'), 'vali', file_explorer='http://host')
    assert link == 'http://host' + str(tmp_path / 'vali_all') + '.html'
    html = open(str(tmp_path / 'vali_all.html')).read()
    assert html.count('<tr>') == 4 and 'NLT (vali)' in html and html.count('<img ') == 12
    for d in dirs['vali']:
        for i in range(2):
            assert join(d, '%d_base-vs-pred.apng' % i) in html and join(d, '%d_gt-vs-pred.apng' % i) in html and join(d, '%d_nn.png' % i) in html
    assert 'pred_psnr' in html and 'Nearest Neighbor' in html
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        link = pm.compile_batch_vis(dirs['test'], str(tmp_path / 'test_all'), 'test', fps=5)
    # the link names a file that EXISTS: the .mp4 where matplotlib + ffmpeg encoded one, the .apng otherwise (advisor r05)
    assert link in (str(tmp_path / 'test_all') + '.mp4', str(tmp_path / 'test_all') + '.apng') and os.path.exists(link)
    roll = json.load(open(str(tmp_path / 'test_all.frames.json')))
    want = sorted(k for k in ids if k.startswith('test'))
    assert roll['ids'] == want and roll['frames'] == [ids[k] for k in want] and roll['fps'] == 5
    with Image.open(str(tmp_path / 'test_all.apng')) as a:
        assert a.n_frames == 4
        for f, k in enumerate(want):                            # frames in id order, pixels = the written predictions
            a.seek(f)
            assert np.array_equal(np.array(a.convert('RGB')), read_png(ids[k]))
    os.remove(ids[want[0]])                                     # a missing prediction is skipped with a warning
    with pytest.warns(UserWarning):
        pm.compile_batch_vis(dirs['test'], str(tmp_path / 'test_again'), 'test')
    assert json.load(open(str(tmp_path / 'test_again.frames.json')))['ids'] == want[1:]
    with pytest.raises(AssertionError):
        pm.compile_batch_vis([str(tmp_path / 'nothing')], str(tmp_path / 'none'), 'train')
    with pytest.raises(ValueError):
        pm.compile_batch_vis(dirs['test'], str(tmp_path / 'x'), 'eval')


def test_the_reference_s_infer_loop_runs_as_written(monkeypatch, tmp_path):
    """nlt/nlt_test.py:78-94 transliterated (tf.tile -> Tensor.repeat) on this model, then `nlt_test.infer(..., outroot)`:
    the same files."""
    fake_capi.install(monkeypatch)
    om, pm = make(256, 64, 64)
    train = [O.synth_batch(2, 64, 64, 64, 64, 64, 64, k=1, seed=60)]
    tests = [cpu_batch(*O.synth_batch(n, 64, 64, 64, 64, 64, 64, k=1, seed=61 + n)) for n in (2, 1, 2)]
    feat_agg = nlt_test.extract_feat(pm, [cpu_batch(b, nn) for b, nn in train])
    model, datapipe, outroot = pm, tests, str(tmp_path / 'ref_loop')
    # ---- the reference's loop body
    batch_i = 0
    for batch in datapipe:
        outdir = join(outroot, 'batch{i:09d}'.format(i=batch_i))
        bs = batch[0].shape[0] if hasattr(batch[0], 'shape') else len(batch[0])
        obs_override = [x.repeat(bs, 1, 1, 1) for x in feat_agg]
        _, _, _, to_vis = model.call(batch, 'test', obs_override=obs_override)
        outdir = outdir.format(i=batch_i)
        model.vis_batch(to_vis, outdir, 'test')
        batch_i += 1
    # ----
    seen = []
    assert nlt_test.infer(pm, tests, feat_agg, str(tmp_path / 'ours'), on_batch=lambda i, v: seen.append(i)) == []
    assert seen == [0, 1, 2]
    for b, n in enumerate((2, 1, 2)):
        for i in range(n):
            for name in ('base', 'pred', 'nn'):
                f = join('batch%09d' % b, '%d_%s.png' % (i, name))
                assert np.array_equal(read_png(join(outroot, f)), read_png(join(str(tmp_path / 'ours'), f)))
            assert exists(join(str(tmp_path / 'ours'), 'batch%09d' % b, '%d_metadata.json' % i))
        assert not exists(join(outroot, 'batch%09d' % b, '%d_pred.png' % n))
    out = nlt_test.infer(pm, tests, feat_agg, str(tmp_path / 'lanes'), lanes=2)                 # writing under a pipeline too
    assert out == [] and np.array_equal(read_png(join(str(tmp_path / 'lanes'), 'batch000000002', '1_pred.png')),
                                        read_png(join(outroot, 'batch000000002', '1_pred.png')))
    assert len(nlt_test.infer(pm, tests, feat_agg)) == 3                                         # no outroot: dicts back
