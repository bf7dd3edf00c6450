"""-m gpu: pipeline.RenderPipeline (several batches in flight on one GPU, one lane of render state per batch) must return,
bit for bit, what Model.call returns for every batch of a sequence -- float batches, store-resident batches, obs_override
rendering (nlt_test.infer) -- while lanes share one set of weights and packed fragments."""
import pytest
import torch

import nlt_amd
from nlt_amd.datasets import get_dataset_class
from nlt_amd.datasets.synth import synthetic_store
from nlt_amd.models import get_model_class
from nlt_amd.pipeline import RenderPipeline

pytestmark = pytest.mark.gpu


def _setup(uv=128, cam=64, k=3, frames=12, bs=2, depth=256):
    store = synthetic_store(frames, uv, cam, seed=5, k=k)
    cfg = nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=cam, imw=cam, bs=bs)
    pm = get_model_class('nlt')(cfg).build('cuda')
    pm.register_trainable()
    g = torch.Generator(device='cuda').manual_seed(9)
    with torch.no_grad():
        for c in pm._conv_layers():
            c.bias.uniform_(-0.1, 0.1, generator=g)
    ds = get_dataset_class('nlt')(cfg, 'train', store, k=k, ring=0)          # ring = 0: every batch in its own buffers
    id_lists = [store['ids'][i * bs:(i + 1) * bs] for i in range(frames // bs)]
    return pm, ds, id_lists


def _same(a, b):
    assert torch.equal(a[0], b[0])
    for key in ('pred', 'base_camspc', 'pred_camspc'):
        assert torch.equal(a[3][key], b[3][key]), key


@pytest.mark.parametrize('lanes,kind', [(2, 'plain'), (3, 'plain'), (4, 'plain'), (3, 'threads'), (4, 'threads'),
                                        (2, 'graphs'), (3, 'graphs')])
@pytest.mark.parametrize('resident', [False, True])
def test_pipelined_batches_equal_sequential_calls(lanes, kind, resident):
    """plain: every lane driven by the caller's thread; threads: one host thread per lane (native tape replay, thread-local
    launch tapes); graphs: single-stream lanes replaying one hipGraph per (lane, input addresses), outputs handed out as copies."""
    pm, ds, id_lists = _setup()
    batches = [ds.load_batch(ids, resident=resident) for ids in id_lists]
    ref = [pm.call(b, 'test') for b in batches]
    torch.cuda.synchronize()
    pipe = RenderPipeline(pm, lanes, threads=kind == 'threads', graphs=kind == 'graphs')
    for rounds in range(3):                                      # eager pass, recorded launch tapes, replays
        tickets = [pipe.submit(b, 'test') for b in batches]
        outs = [t.result() for t in tickets]
        torch.cuda.synchronize()
        for a, b in zip(ref, outs):
            _same(a, b)
    if kind == 'graphs' and not resident:
        assert pipe._lanes[0] is not pm and all(len(l._graphs) == len(batches) // lanes and
                                                all(g['hits'] >= 1 for g in l._graphs.values()) for l in pipe._lanes)
        assert pm.use_graphs is False and pm.plan.two_streams is True               # the caller's model keeps its launch mode
    assert all(l is not None and l.plan is not pm.plan for l in pipe._lanes[1:])
    assert pipe._lanes[1].plan.lds_hints == pm.plan.lds_hints and pipe._lanes[1].plan.tile_hints == pm.plan.tile_hints
    assert pipe._lanes[1].net is pm.net and pipe._lanes[1].flat_params is pm.flat_params       # one set of weights
    # render(): the reference's loop shape, results in order
    outs = pipe.render(batches, 'test')
    torch.cuda.synchronize()
    for a, b in zip(ref, outs):
        _same(a, b)
    pipe.close()


def test_a_failing_lane_raises_from_the_ticket():
    pm, ds, id_lists = _setup(frames=4)
    batches = [ds.load_batch(ids) for ids in id_lists]
    pipe = RenderPipeline(pm, 2, threads=True)
    for _ in range(2):
        [t.result() for t in [pipe.submit(b, 'test') for b in batches]]
    bad = list(batches[1]); bad[1] = bad[1][:, :-1]                       # a base map of the wrong height
    good, broken = pipe.submit(batches[0], 'test'), pipe.submit(tuple(bad), 'test')
    good.result()
    with pytest.raises(Exception):
        broken.result()
    with pytest.raises(ValueError):
        pipe.submit(batches[0], 'train')
    pipe.close()


def test_pipeline_sees_a_weight_update():
    pm, ds, id_lists = _setup(frames=8)
    batches = [ds.load_batch(ids) for ids in id_lists]
    pipe = RenderPipeline(pm, 2)
    for _ in range(2):
        [t.result() for t in [pipe.submit(b, 'test') for b in batches]]
    with torch.no_grad():
        pm.flat_params.mul_(1.01)                                # (an optimizer step / checkpoint load)
    outs = [t.result() for t in [pipe.submit(b, 'test') for b in batches]]
    torch.cuda.synchronize()
    ref = [pm.call(b, 'test') for b in batches]
    torch.cuda.synchronize()
    for a, b in zip(ref, outs):
        _same(a, b)


def test_nlt_test_infer_with_lanes():
    from nlt_amd import nlt_test
    pm, ds, id_lists = _setup(uv=64, cam=32, k=1, frames=8)
    batches = [ds.load_batch(ids) for ids in id_lists]
    agg = nlt_test.extract_feat(pm, batches[:2])
    ref = nlt_test.infer(pm, batches, agg)
    out = nlt_test.infer(pm, batches, agg, lanes=3)
    seen = []
    nlt_test.infer(pm, batches, agg, on_batch=lambda i, v: seen.append((i, v['pred'])), lanes=2)
    torch.cuda.synchronize()
    assert len(out) == len(ref) == len(seen) and [i for i, _ in seen] == list(range(len(ref)))
    for a, b, (_, c) in zip(ref, out, seen):
        assert torch.equal(a['pred'], b['pred']) and torch.equal(a['pred_camspc'], b['pred_camspc']) and torch.equal(a['pred'], c)


def test_pipeline_rejects_what_it_cannot_do():
    pm, _, _ = _setup(frames=2)
    with pytest.raises(ValueError):
        RenderPipeline(pm, 0)
