"""Generates tests/golden/ref_orchestration.npz by IMPORTING AND RUNNING the reference's own model code:

    /root/reference/nlt/models/{base,nlt}.py      Model.__init__, call, _call, compute_loss
    /root/reference/nlt/networks/{base,seq,convnet,elements}.py, util/{net,img,tensor}.py, losses.py:L2
    /root/reference/nlt/nlt_test.py               extract_feat

on seeded synthetic batches, with the weights of `oracle.OracleModel(seed)`.  TensorFlow / TF-Addons / cv2 / absl are not
installable in this image; `tests/tf_shim/` provides the small TF surface that code touches, each primitive delegating
to `oracle/tf_ops.py` (tests/tf_shim/README.md).  So what this file pins is the ORCHESTRATION -- by the reference's own
source, executed -- not the TF kernels.

Run here (needs /root/reference):   python tests/golden/make_ref_orchestration.py
Consumers: tests/test_oracle_ref_orchestration.py (CPU: OracleModel == this file), tests/test_gpu_model.py (HIP <= 1e-4).
Stored: float32 outputs (full at 64^2; strided samples + norms at depth 1024 / 256^2), the seeds that regenerate the inputs.
"""
import os
import sys
from configparser import ConfigParser

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('NLT_REFERENCE', '/root/reference')
sys.path[:0] = [os.path.join(ROOT, 'tests', 'tf_shim'), os.path.join(REF, 'nlt'), os.path.join(REF, 'third_party', 'xiuminglib'), ROOT]

import tensorflow as tf                                   # noqa: E402  (tests/tf_shim)
import models                                             # noqa: E402  (the reference's nlt/models)
import nlt_test as ref_nlt_test                           # noqa: E402  (the reference's nlt/nlt_test.py)
from oracle import nlt_oracle as O                        # noqa: E402

assert models.__file__.startswith(REF) and ref_nlt_test.__file__.startswith(REF), "not the reference's modules"

# one entry per fixture: name -> (model kwargs, batch kwargs, what to run)
CASES = {
    'd256_train':     dict(depth=256, uv=64, im=32, cam=16, n=2, mode='train', wseed=101, bseed=201),
    'd256_vali':      dict(depth=256, uv=64, im=32, cam=32, n=2, mode='vali', wseed=102, bseed=202),
    'd256_test':      dict(depth=256, uv=64, im=32, cam=32, n=1, mode='test', wseed=103, bseed=203),
    'd256_override':  dict(depth=256, uv=64, im=32, cam=32, n=2, mode='test', wseed=104, bseed=204, override=True),
    'd256_no_obs':    dict(depth=256, uv=64, im=32, cam=32, n=2, mode='train', wseed=105, bseed=205, use_obs=False),
    'd256_no_base':   dict(depth=256, uv=64, im=32, cam=32, n=2, mode='train', wseed=106, bseed=206, skip_connect_base=False),
    'd64_train':      dict(depth=64, uv=32, im=32, cam=32, n=3, mode='train', wseed=107, bseed=207),
    'd1024_train':    dict(depth=1024, uv=256, im=32, cam=32, n=1, mode='train', wseed=108, bseed=208, sample=4),
    'd1024_override': dict(depth=1024, uv=256, im=32, cam=32, n=1, mode='test', wseed=109, bseed=209, sample=4, override=True),
}


def make_config(c):
    cfg = ConfigParser()
    cfg['DEFAULT'] = {k: str(v) for k, v in dict(
        imh=c['im'], imw=c['im'], uvh=c['uv'], uvw=c['uv'], depth0=16, depth=c['depth'], kernel=2, stride=2, norm='none',
        act='leakyrelu', pool='none', use_obs=c.get('use_obs', True), skip_connect_base=c.get('skip_connect_base', True),
        loss='l2', bs=c['n']).items()}       # the keys of nlt/config/dragon_specular.ini that the model reads
    return cfg


def T_(x):
    return None if x is None else tf.Tensor(x)


def ref_model(c):
    """The reference's Model with the oracle's weights (Keras layouts both sides: arrays go over unchanged)."""
    om = O.OracleModel(depth=c['depth'], uvh=c['uv'], uvw=c['uv'], imh=c['im'], imw=c['im'], seed=c['wseed'],
                       use_obs=c.get('use_obs', True), skip_connect_base=c.get('skip_connect_base', True))
    m = models.get_model_class('nlt')(make_config(c))
    m.register_trainable()
    w = om.numpy_weights()
    for name in ('query', 'obs'):
        layers = m.net[name].layers
        assert len(layers) == len(w[name]), (name, len(layers), len(w[name]))
        for layer, lw in zip(layers, w[name]):
            convs = [layer] if hasattr(layer, 'set_weights') else [l for l in layer.layers if hasattr(l, 'set_weights')]
            assert len(convs) == len(lw)
            for conv, (k, b) in zip(convs, lw):
                conv.set_weights([k, b])
    return om, m


class Pipe(list):
    """What nlt_test.extract_feat needs of a tf.data pipeline: iteration and take()."""

    def take(self, n):
        return Pipe(self[:n])


def main():
    out = {}
    for name, c in CASES.items():
        om, m = ref_model(c)
        batch, nn = O.synth_batch(c['n'], c['uv'], c['uv'], c['cam'], c['cam'], c['im'], c['im'], k=1, seed=c['bseed'])
        tb = tuple(T_(t) if torch.is_tensor(t) else t for t in batch)
        override = None
        if c.get('override'):
            # nlt_test.py:97-127 on two training batches (unequal sizes), then nlt_test.py:83-86
            train = [O.synth_batch(nf, c['uv'], c['uv'], c['cam'], c['cam'], c['im'], c['im'], k=1, seed=c['bseed'] + 1000 + nf)[0]
                     for nf in ((2, 3) if c['depth'] < 1024 else (1, 2))]
            ref_nlt_test.FLAGS.n_obs_batches = -1
            feat_agg = ref_nlt_test.extract_feat(m, Pipe([tuple(T_(t) if torch.is_tensor(t) else t for t in b) for b in train]))
            ref_nlt_test.FLAGS.n_obs_batches = 1
            feat_one = ref_nlt_test.extract_feat(m, Pipe([tuple(T_(t) if torch.is_tensor(t) else t for t in b) for b in train]))
            for l, f in enumerate(feat_agg):
                assert tuple(f.shape)[0] == 1
                a = f.numpy().astype(np.float32)
                fs = max(1, a.shape[1] // 16) if c['depth'] >= 1024 else 1       # (depth 1024 / 256^2: strided samples + the norm)
                out['%s/feat_agg_%d' % (name, l)] = a[:, ::fs, ::fs]
                out['%s/feat_agg_norm_%d' % (name, l)] = np.array(np.linalg.norm(a.astype(np.float64)))
            out['%s/feat_first_batch_only_norms' % name] = np.array([float(np.linalg.norm(f.numpy().astype(np.float64))) for f in feat_one])
            bs = tb[1].shape[0]
            override = [tf.tile(x, (bs, 1, 1, 1)) for x in feat_agg]
        with torch.no_grad():
            pred_c, gt_c, kw, to_vis = m.call(tb, c['mode'], **({'obs_override': override} if override else {}))
        assert (kw == {}) if c['mode'] != 'test' else (kw is None and gt_c is None)
        st = c.get('sample', 1)
        pred = to_vis['pred'].numpy()
        out['%s/pred' % name] = pred[:, ::st, ::st].astype(np.float32)
        out['%s/pred_norm' % name] = np.array(np.linalg.norm(pred.astype(np.float64)))
        out['%s/pred_camspc' % name] = pred_c.numpy().astype(np.float32)
        out['%s/base_camspc' % name] = to_vis['base_camspc'].numpy().astype(np.float32)
        if gt_c is not None:
            out['%s/gt_camspc' % name] = gt_c.numpy().astype(np.float32)
            loss = m.compute_loss(pred_c, gt_c, keep_batch=True)         # trainvali.py:274-276
            out['%s/loss_per_example' % name] = loss.numpy().astype(np.float64)
            out['%s/loss_scalar' % name] = np.array(m.compute_loss(pred_c, gt_c).numpy(), np.float64)
        # the layer list itself (convnet.py:30-90 as built by the reference)
        q = m.net['query']
        out['%s/is_contracting' % name] = np.array(q.is_contracting, np.int32)
        out['%s/spatsize_changes' % name] = np.array(q.spatsize_changes, np.float64)
        out['%s/n_obs_layers' % name] = np.array(len(m.net['obs'].layers))
        out['%s/meta' % name] = np.array([c['depth'], c['uv'], c['im'], c['cam'], c['n'], c['wseed'], c['bseed'], st,
                                          int(c.get('use_obs', True)), int(c.get('skip_connect_base', True)), int(bool(c.get('override')))])
        out['%s/mode' % name] = np.array(c['mode'])
        print(name, 'pred norm %.6f' % out['%s/pred_norm' % name], flush=True)
    for mode in ('training', 'bogus'):
        try:
            om, m = ref_model(CASES['d256_test'])
            m.call((None,) * 11, mode)
        except ValueError:
            out['bad_mode_raises_ValueError/' + mode] = np.array(1)
    path = os.path.join(HERE, 'ref_orchestration.npz')
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB, %d arrays)" % (path, os.path.getsize(path) / 1024, len(out)))


if __name__ == '__main__':
    main()
