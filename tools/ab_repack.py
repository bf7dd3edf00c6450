"""What the one-launch weight refresh costs and what it refreshes:  python tools/ab_repack.py  (NLT_GRAD_RANGES=2|3)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlt_amd                                                   # noqa: E402
import bench                                                     # noqa: E402
from nlt_amd import trainvali                                    # noqa: E402
from nlt_amd.models import get_model_class                       # noqa: E402

dev = torch.device('cuda')
n, uv, cam = 4, 1024, 512
pm = get_model_class('nlt')(nlt_amd.make_config(depth=256, uvh=uv, uvw=uv, imh=cam, imw=cam, loss='l2')).build(dev)
pm.register_trainable()
opt = trainvali.make_optimizer(pm, pm.config)
batches = [bench.synth_device_batch(n, uv, cam, 1, dev, seed=i) for i in range(3)]
for i in range(9):
    trainvali.distributed_train_step(pm, batches[i % 3], opt, n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30):
    trainvali.distributed_train_step(pm, batches[i % 3], opt, n)
torch.cuda.synchronize()
print("ranges %s: train step %.3f ms" % (pm.bucket_ranges, (time.perf_counter() - t0) / 30 * 1e3))
reg = pm.pack_registry
kinds = {}
for (layer, key, buf, d) in reg.entries.values():
    k = (d.get('kind'), d.get('mode'))
    kinds[k] = (kinds.get(k, (0, 0))[0] + 1, kinds.get(k, (0, 0))[1] + buf.numel())
print("entries %d, elements %d" % (len(reg.entries), sum(b.numel() for _, _, b, _ in reg.entries.values())))
for k, v in sorted(kinds.items(), key=lambda kv: -kv[1][1]):
    print("  kind/mode %s: %d buffers, %d elements" % (k, v[0], v[1]))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reg.refresh(); torch.cuda.synchronize()
e0.record()
for _ in range(20):
    reg.refresh()
e1.record(); torch.cuda.synchronize()
print("repack alone: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
