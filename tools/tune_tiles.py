#!/usr/bin/env python
"""Per-launch timing of the forward plan for every MFMA wave tile (and the direct algorithm).
Run on the GPU box: python tools/tune_tiles.py > gpurun_out/tune.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nlt_amd
from nlt_amd import capi
from nlt_amd.engine import OpTimer
from nlt_amd.models import get_model_class
import bench

uv, cam, frames, k = (int(os.environ.get(x, d)) for x, d in (('UV', 1024), ('CAM', 512), ('FRAMES', 4), ('K', 4)))
dev = torch.device('cuda', 0)
model = get_model_class('nlt')(nlt_amd.make_config(uvh=uv, uvw=uv, imh=cam, imw=cam)).build(dev)
model.register_trainable()
batch = bench.synth_device_batch(frames, uv, cam, k, dev, 7)
res = {}
variants = [('auto', 0)] + [('%dx%d' % (r, c), 16 * r + c) for r in (1, 2, 4) for c in (1, 2, 4)] + [('direct', -1)]
for name, hint in variants:
    model.conv_algo = capi.ALGO_DIRECT if hint < 0 else capi.ALGO_AUTO
    model.plan.tile_hints = {'*': hint} if hint > 0 else {}
    for _ in range(2):
        model.call(batch, 'test')
    t = OpTimer(); model.plan.timer = t
    for _ in range(3):
        model.call(batch, 'test')
    for label, r in t.collect().items():
        res.setdefault(label, {})[name] = (r[1] / r[0], r[2])
    model.plan.timer = None
names = [v[0] for v in variants]
print("%-12s " % "launch" + " ".join("%8s" % n for n in names) + "   best  algGB/s(best)")
tot_auto = tot_best = 0.0
for label, d in res.items():
    row = [d.get(n, (float('nan'), 0))[0] for n in names]
    best = min((v, n) for v, n in zip(row, names) if v == v)
    tot_auto += d['auto'][0]; tot_best += best[0]
    print("%-12s " % label + " ".join("%8.4f" % v for v in row) + "   %-6s %8.1f" % (best[1], d['auto'][1] / best[0] / 1e6))
print("sum auto %.3f ms, sum best-per-launch %.3f ms" % (tot_auto, tot_best))
