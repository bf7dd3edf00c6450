"""Characterises the fresh-box effect on multi-lane rendering: run as the FIRST heavy process on a box.
A new 4-lane pipeline is measured every ~2 s, first with the GPU idle in between, then (a second new pipeline) with
single-lane steps in between."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from nlt_amd.models import get_model_class                       # noqa: E402
from nlt_amd.pipeline import RenderPipeline                      # noqa: E402

sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device('cuda', 0)
cfg, ds, id_lists = bench.make_loader(args, dev, args.k, 'train', seed=100)
model = get_model_class('nlt')(cfg).build(dev)
model.register_trainable()
batches = [ds.load_batch(ids) for ids in id_lists]
t00 = time.perf_counter()
single = bench.time_forward(model, batches, 50)
print('t=%5.1f one batch at a time %.4f ms' % (time.perf_counter() - t00, single * 1e3), flush=True)


def measure(pipe, lanes, steps=60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tk = [pipe.submit(batches[i % len(batches)], 'test') for i in range(steps)]
    for t in tk[-lanes:]:
        t.result()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for phase, between in (('idle', 'sleep'), ('load', 'steps')):
    pipe = RenderPipeline(model, 4)
    for _ in range(2):
        for t in [pipe.submit(batches[i % len(batches)], 'test') for i in range(8)]:
            t.result()
    for rep in range(9):
        print('t=%5.1f new pipeline, %s between: 4 lanes %.4f ms' % (time.perf_counter() - t00, between, measure(pipe, 4)), flush=True)
        if between == 'sleep':
            time.sleep(2.0)
        else:
            t_end = time.perf_counter() + 2.0
            while time.perf_counter() < t_end:
                for i in range(30):
                    model.call(batches[i % len(batches)], 'test')
                torch.cuda.synchronize()
    pipe.close()
print('t=%5.1f one batch at a time %.4f ms' % (time.perf_counter() - t00, bench.time_forward(model, batches, 50) * 1e3))
