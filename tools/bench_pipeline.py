"""Experiment: do two whole forward passes (BASELINE config 3) overlap on one GPU?  Two Model instances (own plans, buffers and
side streams), each called under its own HIP stream, calls interleaved -- against one model called back to back.
    python tools/bench_pipeline.py [--lanes 2] [--steps 100]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from nlt_amd.models import get_model_class                       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--lanes', type=int, default=2)
ap.add_argument('--steps', type=int, default=100)
a = ap.parse_args()
sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device('cuda', 0)
lanes = []
for i in range(a.lanes):
    cfg, ds, id_lists = bench.make_loader(args, dev, args.k, 'train', seed=100 + i)
    m = get_model_class('nlt')(cfg).build(dev)
    m.register_trainable()
    batches = [ds.load_batch(ids) for ids in id_lists]
    for j in range(6):
        m.call(batches[j % len(batches)], 'test')                # plan-time trials, launch tapes
    lanes.append((m, batches, torch.cuda.Stream(device=dev)))
torch.cuda.synchronize()
texels = args.frames * args.uv * args.uv


def timed(fn, steps):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        fn(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def one(s):
    m, b, _ = lanes[0]
    m.call(b[s % len(b)], 'test')


def piped(s):
    m, b, st = lanes[s % len(lanes)]
    with torch.cuda.stream(st):
        m.call(b[(s // len(lanes)) % len(b)], 'test')


t1 = timed(one, a.steps)
t2 = timed(piped, a.steps)
t1b = timed(one, a.steps)
print('one lane      %.4f ms / step  %.1f Mtexels/s' % (t1 * 1e3, texels / t1 / 1e6))
print('%d lanes       %.4f ms / step  %.1f Mtexels/s  (x%.3f)' % (len(lanes), t2 * 1e3, texels / t2 / 1e6, t1 / t2))
print('one lane again %.4f ms / step' % (t1b * 1e3))
