"""Inference orchestration of the reference's nlt/nlt_test.py:78-127 on the HIP kernels.

`extract_feat` runs the OBSERVATION path alone over training batches and averages every level's feature map
over all frames; `infer` renders test batches with those averages standing in for the per-frame observation
features (`obs_override`).  Unlike the reference, which still pushes a placeholder neighbour through the
observation network on every test batch and throws the result away (nlt/models/nlt.py:154-155,172-173), the
plan skips the observation convs entirely when an override is given, and the running average never
concatenates all frames' features (nlt_test.py:116-121 keeps every frame of every level alive)."""
import logging
from os.path import join

import torch

from . import _capi as C

logger = logging.getLogger('nlt_test')


def _obs_features(model, x):
    """[N,H,W,3] -> list of per-level observation feature maps, each [N,h,w,C] (nlt_test.py:108-114)."""
    feats = []
    for layer in model.net['obs'].layers:
        x = layer(x.contiguous())
        feats.append(x)
    return feats


def _mean_over_frames(x, weights=None):
    """[F,h,w,C] -> [1,h,w,C] = sum_f w_f x_f / F with the observation-mean kernel (one group of F members)."""
    f, h, w, c = x.shape
    out = torch.empty((1, h, w, c), device=x.device, dtype=torch.float32)
    C.obs_mean_forward(x.contiguous(), weights, 1, f, h * w, c, out, c)
    return out


def extract_feat(model, datapipe, n_obs_batches=-1):
    """nlt_test.py:97-127.  datapipe: iterable of the model's 11-tuples (training batches).  Returns one
    [1,h,w,C] tensor per encoder level: the mean of that level's observation features over every frame seen."""
    per_batch, counts = [], []
    for i, batch in enumerate(datapipe):
        if 0 < n_obs_batches <= i:
            break
        base, rgb = batch[1], batch[5]
        x = rgb - base                                              # nlt_test.py:107
        per_batch.append([_mean_over_frames(f) for f in _obs_features(model, x)])
        counts.append(base.shape[0])
    if not per_batch:
        raise ValueError("extract_feat needs at least one batch")
    total = float(sum(counts))
    nb = len(per_batch)
    # mean over all frames = sum_b (n_b / total) * mean_b: one more pass of the same kernel, weights nb * n_b / total
    w = torch.tensor([[nb * c / total for c in counts]], device=per_batch[0][0].device, dtype=torch.float32)
    return [_mean_over_frames(torch.cat([pb[level] for pb in per_batch], 0), w) for level in range(len(per_batch[0]))]


def infer(model, datapipe, feat_agg, outroot=None, report_every=10, on_batch=None, lanes=1, threads=False):
    """nlt_test.py:78-94: renders every test batch with the aggregated observation features and visualises it into
    `<outroot>/batch<i:09d>` (`model.vis_batch(to_vis, outdir, 'test')`, as the reference).  outroot = None: nothing is
    written; the `to_vis` dicts are returned (or handed to `on_batch(i, to_vis)`; with outroot AND on_batch both happen).
    lanes > 1: that many batches in flight on the GPU (pipeline.RenderPipeline; same results; a `datapipe` that reuses
    staging buffers needs lanes + 1 slots; threads: one host thread per lane) -- the PNG encoding of batch i then runs on
    the host while batches i + 1 ... are on the GPU."""
    outs, done = [], [0]

    def sink(i, to_vis):
        if outroot is not None:
            model.vis_batch(to_vis, join(outroot, 'batch{i:09d}'.format(i=i)), 'test')
        if on_batch is not None:
            on_batch(i, to_vis)
        elif outroot is None:
            outs.append(to_vis)
        done[0] += 1
        if done[0] % report_every == 0:
            logger.info("Done inferring %d batches", done[0])

    if lanes > 1:
        from .pipeline import RenderPipeline
        with RenderPipeline(model, lanes, threads=threads) as pipe:      # (lane threads / streams released on the way out)
            pipe.render(datapipe, 'test', on_batch=lambda i, r: sink(i, r[3]), obs_override=feat_agg)
        return outs
    for i, batch in enumerate(datapipe):
        _, _, _, to_vis = model.call(batch, 'test', obs_override=feat_agg)
        sink(i, to_vis)
    return outs
