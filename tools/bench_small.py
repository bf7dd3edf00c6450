"""Experiment: the released small shapes (BASELINE config 1 / 2 at 4 frames) -- eager launch tapes vs hipGraph replay, 1..8
batches in flight (pipeline.RenderPipeline).  Each lane renders ONE fixed batch (a graph is tied to its input addresses).
    python tools/bench_small.py [--config 1|2]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
import nlt_amd                                                   # noqa: E402
from nlt_amd.models import get_model_class                       # noqa: E402
from nlt_amd.pipeline import RenderPipeline                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', type=int, default=1)
ap.add_argument('--frames', type=int, default=4)
ap.add_argument('--steps', type=int, default=240)
ap.add_argument('--modes', default='eager,threads,graph')
ap.add_argument('--lanes', default='1,2,4,8')
a = ap.parse_args()
depth, uv = ((1024, 256), (256, 512))[a.config - 1]
dev = torch.device('cuda', 0)
cfg = nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=uv, imw=uv, bs=a.frames)
model = get_model_class('nlt')(cfg).build(dev)
model.register_trainable()
batches = bench.identity_batches(a.frames, uv, uv, 1, dev, nb=8)
for i in range(6):
    model.call(batches[i % 3], 'test')
torch.cuda.synchronize()
texels = a.frames * uv * uv
for graphs, threads in [x for x in ((False, False), (False, True), (True, False)) if ('graph' if x[0] else 'threads' if x[1] else 'eager') in a.modes.split(',')]:
    for lanes in [int(x) for x in a.lanes.split(',')]:
        if threads and lanes == 1:
            continue
        model.use_graphs = graphs
        model._graph = None
        pipe = RenderPipeline(model, lanes, threads=threads)
        for _ in range(4):
            for t in [pipe.submit(batches[i % lanes], 'test') for i in range(2 * lanes)]:
                t.result()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tickets = [pipe.submit(batches[i % lanes], 'test') for i in range(a.steps)]
        t_host = time.perf_counter() - t0
        for t in tickets[-lanes:]:
            t.result()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        pipe.close()
        print('config %d  %-7s lanes %d: %.4f ms / step  %.1f Mtexels/s   (host enqueue %.4f ms / step)'
              % (a.config, 'graph' if graphs else ('threads' if threads else 'eager'), lanes, dt * 1e3, texels / dt / 1e6, t_host / a.steps * 1e3))
