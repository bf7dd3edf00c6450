"""CPU, world_size 2, 4 and 8 over gloo: the data-parallel train step (frames sharded across ranks, the gradient sum as TWO
all-reduces of fixed ranges of the flat bucket -- the first issued from inside the backward plan --, identical Adam step
on every rank, scalar loss all-reduce) reproduces the single-process oracle on the full batch
(nlt/trainvali.py:267-325).  Device kernels are replaced by the TEST-ONLY C-ABI emulation; the collective path is the
real one."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLOBAL = {2: 4, 4: 7, 8: 32}                   # global batch per world size (7 over 4 ranks: shards 2, 2, 2, 1; 32 over 8 = BASELINE config 4)


def _shard(world, rank):
    per = -(-GLOBAL[world] // world)
    return slice(per * rank, min(per * (rank + 1), GLOBAL[world]))


def _worker(rank, world, port, outdir, mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import nlt_amd
    from nlt_amd import trainvali
    from oracle import nlt_oracle as O
    import fake_capi
    from test_host_orchestration import make, cpu_batch

    class MP:                                   # minimal monkeypatch stand-in
        def setattr(self, obj, name, val):
            setattr(obj, name, val)
    fake_capi.install(MP())
    if mode == 'branch':                                                    # a layer-by-layer config: norm = layer (generic.py)
        from test_host_generic import make as make_branch
        om, pm = make_branch(32, 64, 32, loss='l2', norm='layer')
        assert pm.generic
    else:
        om, pm = make(256, 64, 32, loss='l2')
    pm.build('cpu'); pm.register_trainable()
    gbs = GLOBAL[world]
    batch, nn = O.synth_batch(gbs, 64, 64, 32, 32, 32, 32, k=1 if world == 8 else 2, seed=5)   # the GLOBAL batch
    sl = _shard(world, rank)                                                # contiguous shard per rank
    shard = tuple(t[sl] if torch.is_tensor(t) else t for t in batch)
    nn_s = [(b[sl], r[sl]) for b, r in nn]
    opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
    fired, trace = [], []
    wgrad_now = pm.plan._wgrad_now
    pm.plan._wgrad_now = lambda label, *a, **kw: (trace.append(label), wgrad_now(label, *a, **kw))[1]
    if mode == 'graphed':
        step = trainvali.GraphedTrainStep(pm, opt, gbs, warmup=0)         # off the GPU: degrades to the eager step
        run = lambda: step(cpu_batch(shard, nn_s))
    else:
        def run():
            real = dist.all_reduce

            def spy(t, *a, **kw):                                         # order and sizes of the collectives
                fired.append(t.numel())
                trace.append(t.numel())
                return real(t, *a, **kw)
            dist.all_reduce = spy
            try:
                return trainvali.distributed_train_step(pm, cpu_batch(shard, nn_s), opt, global_bs=gbs, overlap=(mode == 'overlap'))
            finally:
                dist.all_reduce = real
    losses = []
    for _ in range(2):
        loss, _ = run()
        losses.append(float(loss))
    lv, _ = trainvali.distributed_vali_step(pm, cpu_batch(shard, nn_s), gbs)
    base = pm.flat_params.data_ptr()
    offs = [((v.data_ptr() - base) // 4, v.numel()) for c in pm._conv_layers() for v in (c.kernel, c.bias)]
    torch.save({'losses': losses, 'vali': float(lv), 'params': pm.flat_params.detach().clone(),
                'grad': pm.flat_params.grad.clone(), 'offs': offs, 'fired': fired, 'split': pm.bucket_split, 'ranges': list(pm.bucket_ranges), 'trace': trace},
               os.path.join(outdir, 'r%d.pt' % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('world,mode', [(2, 'overlap'), (2, 'after'), (4, 'overlap'), (2, 'graphed'), (2, 'branch'), (8, 'overlap')])
def test_data_parallel_train_step(world, mode):
    """(8, overlap) is north_star's config 4 partition: global batch 32, 4 frames per rank, k = 1."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import nlt_oracle as O
    with tempfile.TemporaryDirectory() as td:
        port = 29500 + (os.getpid() * 7 + world * 13 + len(mode)) % 2000
        mp.spawn(_worker, args=(world, port, td, mode), nprocs=world, join=True)
        r = [torch.load(os.path.join(td, 'r%d.pt' % i)) for i in range(world)]
    # every rank holds bit-identical weights and gradients after the all-reduces + Adam step
    for x in r[1:]:
        assert torch.equal(r[0]['params'], x['params']) and torch.equal(r[0]['grad'], x['grad'])
        assert r[0]['losses'] == x['losses'] and r[0]['vali'] == x['vali']
    if mode == 'branch':
        assert r[0]['fired'] == [r[0]['params'].numel(), 1] * 2      # layer-by-layer configs: the whole bucket at once
    elif mode != 'graphed':
        # per step: the bucket's three ranges in backward-completion order (expanding blocks | encoder levels 6..3 | the rest),
        # then the scalar loss
        n, split, rg = r[0]['params'].numel(), r[0]['split'], r[0]['ranges']
        assert rg[0] == 0 and rg[1] == split and rg[-1] == n and len(rg) == 4 and rg == sorted(rg)
        sizes = [rg[i + 1] - rg[i] for i in range(3)]
        assert all(x > 0 for x in sizes) and r[0]['fired'] == (sizes + [1]) * 2
        assert sizes[2] < 0.02 * n < sizes[0] < sizes[1]               # what stays exposed behind the backward is the small tail
        # every slot lies inside exactly one range, and the ranges hold what their names say
        for off, cnt in r[0]['offs']:
            assert sum(rg[i] <= off and off + cnt <= rg[i + 1] for i in range(3)) == 1
        # WHERE in the backward the collectives start (one step's trace: weight-gradient labels and collective sizes)
        tr = r[0]['trace']
        step = tr[:tr.index(1) + 1]                                     # up to and including the loss all-reduce of step 0
        at = [step.index(x) for x in sizes]
        lev = lambda lab: int(lab.split('.')[1][1:])
        before = lambda i: [lev(x) for x in step[:at[i]] if isinstance(x, str) and x.endswith('.wgrad')]
        after = lambda i: [lev(x) for x in step[at[i]:] if isinstance(x, str) and x.endswith('.wgrad')]
        if mode == 'overlap':
            assert set(before(0)) >= {7, 8, 9, 10, 11} and min(before(0)) >= 7 and max(after(0)) <= 6   # (the last block: fused end)
            assert set(before(1)) >= {3, 4, 5, 6} and all(l < 3 for l in after(1)) and after(1)       # range 1 leaves mid-backward
            assert at[2] > max(i for i, x in enumerate(step) if isinstance(x, str))               # range 2 behind the backward
        else:
            assert at[0] > max(i for i, x in enumerate(step) if isinstance(x, str))
    # ... and they equal the single-process oracle on the full batch
    gbs = GLOBAL[world]
    om = (O.OracleModel(depth=32, uvh=64, uvw=64, imh=32, imw=32, seed=1, loss='l2', norm='layer') if mode == 'branch' else
          O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32, seed=1, loss='l2'))
    batch, nn = O.synth_batch(gbs, 64, 64, 32, 32, 32, 32, k=1 if world == 8 else 2, seed=5)
    opt = O.KerasAdamAMSGrad(om.parameters(), 1e-3)
    ref_losses = [float(O.train_step(om, opt, batch, global_bs=gbs, nn_list=nn)[0]) for _ in range(2)]
    np.testing.assert_allclose(r[0]['losses'], ref_losses, rtol=1e-5)
    got = r[0]['params']
    for p, (off, cnt) in zip(om.parameters(), r[0]['offs']):       # through the product's slot table
        assert cnt == p.numel()
        assert float((got[off:off + cnt] - p.detach().reshape(-1)).abs().max()) < 2e-5
