import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, nlt_amd, bench
from nlt_amd.models import get_model_class
dev = torch.device('cuda', 0)
model = get_model_class('nlt')(nlt_amd.make_config(uvh=1024, uvw=1024, imh=512, imw=512)).build(dev)
model.register_trainable()
batch = bench.synth_device_batch(4, 1024, 512, 4, dev, 1)
for _ in range(5): model.call(batch, 'test')
torch.cuda.synchronize()
# enqueue-only time: tiny problem so the GPU is never the bottleneck
small = bench.synth_device_batch(1, 64, 32, 4, dev, 2)
m2 = get_model_class('nlt')(nlt_amd.make_config(uvh=64, uvw=64, imh=32, imw=32)).build(dev); m2.register_trainable()
for _ in range(5): m2.call(small, 'test')
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): m2.call(small, 'test')
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host enqueue per step (64^2 problem): %.3f ms; with drain %.3f ms' % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
t0 = time.perf_counter()
for _ in range(50): model.call(batch, 'test')
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('1024^2: enqueue loop %.3f ms/step, total %.3f ms/step' % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
