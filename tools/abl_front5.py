import os, sys, torch
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import nlt_amd
from nlt_amd import capi as C
from nlt_amd.models import get_model_class
sys.path.insert(0, 'tools')
from ab_front import time_it
n, h, w = 4, 1024, 1024
pm = get_model_class('nlt')(nlt_amd.make_config(depth=256, uvh=h, uvw=w, imh=512, imw=512)).build('cuda')
blob, blob_l2 = pm.plan._front_weights(torch.device('cuda'))
g = torch.Generator(device='cuda').manual_seed(0)
res = {}
for k in (1, 4):
    U = lambda *s: torch.rand(s, device='cuda', generator=g)
    fl = (U(n, h, w, 3), U(n, h, w, 1), U(n, h, w, 1), U(n, k, h, w, 3), U(n, k, h, w, 3))
    E = lambda *s: torch.empty(s, device='cuda')
    outs = (E(n, h // 2, w // 2, 32), E(n, h, w, 3), E(n, h // 4, w // 4, 32), E(n, k, h // 4, w // 4, 32))
    res[k] = time_it(lambda: C.front5_forward(*fl, n, k, h, w, blob, blob_l2, True, 0.3, *outs, 9))
per = (res[4] - res[1]) / 3
print("ABL %s: k1 %.4f k4 %.4f  per-obs %.4f else %.4f" % (os.environ.get('NLT_F5_ABL', '0'), res[1], res[4], per, res[1] - per))
