"""Per-launch table + wall time of a forward at another BASELINE config:  python tools/bench_cfg.py depth uv frames k cam [graph]
   config 1 (dragon_sss shape): 1024 256 4 1 256      config 2 (relight only): 256 512 4 1 512"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlt_amd                                                   # noqa: E402
import bench                                                     # noqa: E402
from nlt_amd.engine import OpTimer                               # noqa: E402
from nlt_amd.models import get_model_class                       # noqa: E402

depth, uv, n, k, cam = (int(x) for x in sys.argv[1:6])
graph = len(sys.argv) > 6 and sys.argv[6] == 'graph'
quiet = len(sys.argv) > 6 and sys.argv[6] == 'quiet'          # (counter passes: no per-launch survey)
dev = torch.device('cuda')
pm = get_model_class('nlt')(nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=cam, imw=cam)).build(dev)
pm.register_trainable()
pm.use_graphs = graph
if os.environ.get('ONE_STREAM') == '1':
    pm.plan.two_streams = False
batches = [bench.synth_device_batch(n, uv, cam, k, dev, seed=i) for i in range(3)]
if cam == uv:                                                    # relight only: identity warp
    jj, ii = torch.meshgrid(torch.arange(cam, device=dev), torch.arange(cam, device=dev), indexing='xy')
    wp = torch.stack((jj / cam, ii / cam), -1)[None].repeat(n, 1, 1, 1).float().contiguous()
    batches = [b[:4] + (wp,) + b[5:] for b in batches]
for i in range(9):
    pm.call(batches[i % 3], 'test')
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 10 if quiet else 100
for i in range(steps):
    pm.call(batches[i % 3], 'test')
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("depth %d uv %d frames %d k %d: %.4f ms / step = %.1f Mtexels/s (%s, tape replays %d)"
      % (depth, uv, n, k, 1e3 * dt, n * uv * uv / dt / 1e6, 'hipGraph' if graph else 'eager', pm.plan.tape_replays))
if not graph and not quiet:
    t = OpTimer(); pm.plan.timer = t
    for i in range(3):
        pm.call(batches[i % 3], 'test')
    rec = t.collect(); pm.plan.timer = None
    tab = sorted(((r[1] / r[0], l) for l, r in rec.items()), reverse=True)
    tot = sum(x for x, _ in tab)
    for x, l in tab[:28]:
        print("%-12s %8.4f ms %5.1f%%  %7.1f TFLOP/s" % (l, x, 100 * x / tot, t.flops.get(l, 0) / x / 1e9))
    print("sum of launches %.3f ms (%d launches)" % (tot, len(tab)))
