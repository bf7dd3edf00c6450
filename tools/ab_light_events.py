"""Light (no system-scope fence) vs fenced events in the train step's backward plan, one rank, bitwise:
   python tools/ab_light_events.py [uv] [steps] [hog]     hog: a second stream runs a memory hog beside the plan (contention)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlt_amd                                                   # noqa: E402
import bench                                                     # noqa: E402
from nlt_amd import capi                                         # noqa: E402
from nlt_amd.models import get_model_class                       # noqa: E402

uv = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
hog = len(sys.argv) > 3 and sys.argv[3] == 'hog'
dev = torch.device('cuda')
cfg = nlt_amd.make_config(depth=256, uvh=uv, uvw=uv, imh=uv // 2, imw=uv // 2, bs=4)
batch = bench.synth_device_batch(4, uv, uv // 2, 1, dev, seed=3)
base, cvis, lvis, warp, nn_base, nn_rgb = batch[1], batch[2], batch[3], batch[4], batch[8], batch[9]
g = torch.Generator(device=dev).manual_seed(77)
dpred = (torch.rand(tuple(base.shape), device=dev, generator=g) - 0.5).contiguous()
hs = torch.cuda.Stream()
hb = torch.empty(64 << 20, device=dev)
res = {}
for leg, light in (("light", True), ("fenced", False), ("light2", True)):
    capi.LIGHT_EVENTS = light
    model = get_model_class('nlt')(cfg).build(dev)
    model.register_trainable()
    gw = torch.Generator(device=dev).manual_seed(4321)
    with torch.no_grad():
        for c in model._conv_layers():
            c.bias.uniform_(-0.1, 0.1, generator=gw)
        model.flat_params.copy_(res['w']) if 'w' in res else None
    res.setdefault('w', model.flat_params.detach().clone())
    model.mark_weights_updated()
    if 'tune' in res:
        model.plan.import_tuning(res['tune'])             # the SAME plan-time choices in every leg: same kernels, same summation orders
    outs = []
    for i in range(steps):
        if hog:
            with torch.cuda.stream(hs):
                for _ in range(8):
                    hb.mul_(1.0001)
        with torch.no_grad():
            model._render(base, cvis, lvis, warp, nn_rgb, nn_base, None, None, False, inference=False)
            model.flat_grads.zero_()
            model.plan.backward(dpred, base, cvis, lvis, nn_rgb, nn_base, None, generation=model.plan.generation)
        torch.cuda.synchronize()
        outs.append(model.flat_grads.clone())
    res[leg] = outs
    print(leg, "self-consistent:", all(torch.equal(o, outs[0]) for o in outs[1:]), "replays", model.plan.tape_replays, flush=True)
    if leg == "light":
        res['tune'] = model.plan.export_tuning()
    del model
for a in ("light", "light2"):
    same = [torch.equal(x, y) for x, y in zip(res[a], res["fenced"])]
    d = max(float((x - y).abs().max()) for x, y in zip(res[a], res["fenced"]))
    print(a, "vs fenced: bit-identical per step", same, "max abs diff %.3e" % d, "grad max %.3e" % float(res["fenced"][0].abs().max()))
