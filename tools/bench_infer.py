"""The reference's inference mode (nlt_test.extract_feat -> nlt_test.infer, Model.call(obs_override=feat_agg)) timed:
   python tools/bench_infer.py depth uv frames cam [general] [lanes=N] [quiet]
   config 3's UV size: 256 1024 4 512      config 2: 256 512 4 512"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlt_amd                                                   # noqa: E402
import bench                                                     # noqa: E402
from nlt_amd import nlt_test                                     # noqa: E402
from nlt_amd.engine import OpTimer                               # noqa: E402
from nlt_amd.models import get_model_class                       # noqa: E402

depth, uv, n, cam = (int(x) for x in sys.argv[1:5])
flags = sys.argv[5:]
general, quiet = 'general' in flags, 'quiet' in flags
lanes = max([int(f.split('=')[1]) for f in flags if f.startswith('lanes=')] + [1])
dev = torch.device('cuda')
pm = get_model_class('nlt')(nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=cam, imw=cam)).build(dev)
pm.register_trainable()
pm.plan.fuse_override = not general
train = [bench.synth_device_batch(n, uv, cam, 1, dev, seed=10 + i) for i in range(2)]       # feat_agg from 2 x n training frames
batches = [bench.synth_device_batch(n, uv, cam, 1, dev, seed=i) for i in range(3)]
agg = nlt_test.extract_feat(pm, train)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    agg = nlt_test.extract_feat(pm, train)
torch.cuda.synchronize()
print("extract_feat over %d frames: %.3f ms (%.1f Mtexels/s)" % (2 * n, 1e3 * (time.perf_counter() - t0) / 3,
                                                                 2 * n * uv * uv * 3 / (time.perf_counter() - t0) / 1e6))
t0 = time.perf_counter()
pm.call(batches[0], 'test', obs_override=agg)
torch.cuda.synchronize()
print("first call (override maps + plan-time trials): %.1f ms" % (1e3 * (time.perf_counter() - t0)))
for i in range(9):
    pm.call(batches[i % 3], 'test', obs_override=agg)
torch.cuda.synchronize()
steps = 10 if quiet else 100
if lanes > 1:
    from nlt_amd.pipeline import RenderPipeline
    with RenderPipeline(pm, lanes) as pipe:
        pipe.render([batches[i % 3] for i in range(12)], 'test', obs_override=agg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.render([batches[i % 3] for i in range(steps)], 'test', obs_override=agg)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
else:
    t0 = time.perf_counter()
    for i in range(steps):
        pm.call(batches[i % 3], 'test', obs_override=agg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
print("infer depth %d uv %d frames %d (%s plan, %d lane%s): %.4f ms / step = %.1f Mtexels/s (tape replays %d)"
      % (depth, uv, n, 'general' if general else 'fused', lanes, 's' * (lanes > 1), 1e3 * dt, n * uv * uv / dt / 1e6, pm.plan.tape_replays))
if not quiet:
    t = OpTimer(); pm.plan.timer = t
    for i in range(3):
        pm.call(batches[i % 3], 'test', obs_override=agg)
    rec = t.collect(); pm.plan.timer = None
    tab = sorted(((r[1] / r[0], l) for l, r in rec.items()), reverse=True)
    tot = sum(x for x, _ in tab)
    for x, l in tab[:32]:
        print("%-12s %8.4f ms %5.1f%%  %7.1f TFLOP/s  %7.1f GB/s" % (l, x, 100 * x / tot, t.flops.get(l, 0) / x / 1e9,
                                                                    t.moved.get(l, rec[l][2]) / x / 1e6))
    print("sum of launches %.3f ms (%d launches)" % (tot, len(tab)))
