"""CPU, world_size 2 over gloo: the data-parallel train step (frames sharded across ranks, ONE
all-reduce(sum) of the flat gradient bucket, identical Adam step on every rank, scalar loss
all-reduce) reproduces the single-process oracle on the full batch (nlt/trainvali.py:267-325).
Device kernels are replaced by the TEST-ONLY C-ABI emulation; the collective path is the real one."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outdir):
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
    om, pm = make(256, 64, 32, loss='l2')
    pm.build('cpu'); pm.register_trainable()
    batch, nn = O.synth_batch(4, 64, 64, 32, 32, 32, 32, k=2, seed=5)     # GLOBAL batch of 4 frames
    sl = slice(2 * rank, 2 * rank + 2)                                      # contiguous shard per rank
    shard = tuple(t[sl] if torch.is_tensor(t) else t for t in batch)
    nn_s = [(b[sl], r[sl]) for b, r in nn]
    opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
    losses = []
    for _ in range(2):
        loss, _ = trainvali.distributed_train_step(pm, cpu_batch(shard, nn_s), opt, global_bs=4)
        losses.append(float(loss))
    lv, _ = trainvali.distributed_vali_step(pm, cpu_batch(shard, nn_s), 4)
    torch.save({'losses': losses, 'vali': float(lv), 'params': pm.flat_params.detach().clone(),
                'grad': pm.flat_params.grad.clone()}, os.path.join(outdir, 'r%d.pt' % rank))
    dist.destroy_process_group()


def test_data_parallel_train_step_world2():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import nlt_oracle as O
    with tempfile.TemporaryDirectory() as td:
        port = 29500 + os.getpid() % 2000
        mp.spawn(_worker, args=(2, port, td), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(td, 'r0.pt')), torch.load(os.path.join(td, 'r1.pt'))
    # every rank holds bit-identical weights and gradients after the all-reduce + Adam step
    assert torch.equal(r0['params'], r1['params']) and torch.equal(r0['grad'], r1['grad'])
    assert r0['losses'] == r1['losses'] and r0['vali'] == r1['vali']
    # ... and they equal the single-process oracle on the full batch of 4
    om = O.OracleModel(depth=256, uvh=64, uvw=64, imh=32, imw=32, seed=1, loss='l2')
    batch, nn = O.synth_batch(4, 64, 64, 32, 32, 32, 32, k=2, seed=5)
    opt = O.KerasAdamAMSGrad(om.parameters(), 1e-3)
    ref_losses = [float(O.train_step(om, opt, batch, global_bs=4, nn_list=nn)[0]) for _ in range(2)]
    np.testing.assert_allclose(r0['losses'], ref_losses, rtol=1e-5)
    flat_ref = torch.cat([p.detach().reshape(-1) for p in om.parameters()])
    got = r0['params']
    # compare through the product's slot layout: every oracle tensor appears contiguously, in order
    off = 0
    for p in om.parameters():
        n = p.numel()
        assert float((got[off:off + n] - p.detach().reshape(-1)).abs().max()) < 2e-5
        off += (n + 3) // 4 * 4
