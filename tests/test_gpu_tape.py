"""-m gpu: the launch tape (nlt_amd/_capi.py, engine.RenderPlan): replaying a step's recorded C calls gives the same
results as issuing them through the Python adapters, follows in-place weight updates, and is dropped when it must be."""
import numpy as np
import pytest
import torch

import nlt_amd
from nlt_amd import trainvali
from oracle import nlt_oracle as O
from gpu_util import make_pair, to_device_batch

pytestmark = pytest.mark.gpu


def _model(tape, seed=5, loss='l2', like=None):
    """like = an already exercised model whose plan-time choices the new one takes over (same kernels, same summation
    orders: what two legs then differ by is what the test is about, not two independent tunings)."""
    _, pm = make_pair(depth=256, uv=64, im=32, loss=loss, seed=seed)
    pm.build('cuda')                       # (make_pair already registered the trainables; build() flattens them)
    pm.plan.use_tape = tape
    if like is not None:
        pm.plan.import_tuning(like.plan.export_tuning())
    return pm


def test_forward_replays_match_adapter_launches_and_follow_weight_updates():
    batch = to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=3, seed=70))
    a, b = _model(True), _model(False)
    outs = [[m.call(batch, 'test')[0].clone() for _ in range(4)] for m in (a, b)]
    assert a.plan.tape_replays >= 2 and b.plan.tape_replays == 0      # sights 3 and 4 are replays
    for x in outs[0][1:]:
        assert torch.equal(x, outs[0][0])                              # a replay = the same launches, bit for bit
    close = lambda x, y: float((x - y).norm() / y.norm()) < 1e-5        # (the two models autotune separately: other tiles,
    assert close(outs[0][0], outs[1][0])                               #  other split-K factors -> fp32 re-association only)
    # model b's plan-time autotune grew the shared split-K workspace: every tape recorded before that is invalid and the
    # next sight re-records instead of replaying a dangling pointer
    r = a.plan.tape_replays
    a.call(batch, 'test')
    assert a.plan.tape_replays in (r, r + 1)
    a.call(batch, 'test')
    with torch.no_grad():                                              # an optimizer-style in-place update of the flat bucket
        for m in (a, b):
            m.flat_params.mul_(1.01)
            m.mark_weights_updated()
    before = a.plan.tape_replays
    x, y = a.call(batch, 'test')[0], b.call(batch, 'test')[0]
    assert a.plan.tape_replays == before + 1                          # same tape, refreshed weights
    assert close(x, y) and not close(x, outs[0][0])
    other = to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=3, seed=71))     # new tensors: no replay of the old tape
    assert close(a.call(other, 'test')[0], b.call(other, 'test')[0])
    assert a.plan.tape_replays == before + 1


def test_rendered_texels_are_handed_out_without_a_copy_and_survive_later_replays():
    """Inference: the plan's last launch writes `pred` into a tensor of THAT call (it is kept out of the launch tape and re-issued
    after a replay), so to_vis['pred'] of an earlier call is not overwritten by a later replay of the same tape, and equals what the
    copying path (NLT_PRED_COPY=1 semantics: plan buffer + clone) returns, bit for bit."""
    batch = to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=73))
    other = to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=74))
    a = _model(True)
    first = a.call(batch, 'test')[3]['pred']
    keep = first.clone()
    outs = [a.call(batch, 'test')[3]['pred'] for _ in range(4)]
    assert a.plan.tape_replays >= 2
    assert len({t.data_ptr() for t in outs + [first]}) == 5                      # five calls, five tensors
    assert all(torch.equal(t, keep) for t in outs) and torch.equal(first, keep)
    (bufs,) = a.plan._bufs.values()
    assert all(t.data_ptr() != bufs['pred'].data_ptr() for t in outs)            # none of them is the plan's reusable buffer
    # another batch in between, then the first one again: the earlier results are still intact
    o2 = a.call(other, 'test')[3]['pred']
    again = a.call(batch, 'test')[3]['pred']
    assert torch.equal(again, keep) and torch.equal(outs[0], keep) and not torch.equal(o2, keep)


def test_train_steps_with_and_without_the_tape_agree():
    batch = to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=72))
    res = []
    pm = None
    for tape in (True, False):
        pm = _model(tape, seed=6, loss='barron', like=pm)
        opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
        losses = [float(trainvali.distributed_train_step(pm, batch, opt, 2)[0]) for _ in range(6)]
        res.append((losses, pm.flat_params.detach().clone(), pm.plan.tape_replays))
    (l0, p0, r0), (l1, p1, r1) = res
    assert r0 >= 4 and r1 == 0                                         # forward + backward tapes from step 3 on (re-recorded once more when
                                                                        # the fragment-buffer census retires the trial candidates' buffers)
    np.testing.assert_allclose(l0, l1, rtol=1e-4)
    assert float((p0 - p1).abs().max()) < 1e-4                        # float atomics in the warp scatter, carried through six Adam steps


def test_native_replay_of_the_tape_matches_the_python_replay(monkeypatch):
    """NLT_NATIVE_REPLAY=1 (csrc/tape.hip: nlt_tape_play walks the recorded calls in C, the second-stream event hand-overs
    included): forward and train steps give what the per-call Python replay gives -- the same launches with the same arguments."""
    from nlt_amd import capi as C
    batch = to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=71))
    res = {}
    pm = None
    for native in (False, True):
        monkeypatch.setattr(C, 'NATIVE_REPLAY', native)
        pm = _model(True, seed=6, like=pm)
        fwd = [pm.call(batch, 'test')[0].clone() for _ in range(4)]
        opt = nlt_amd.optim.AdamAMSGrad(pm, 1e-3)
        losses = [float(trainvali.distributed_train_step(pm, batch, opt, 2)[0]) for _ in range(5)]
        torch.cuda.synchronize()
        assert pm.plan.tape_replays >= 4
        res[native] = (fwd, losses, pm.flat_params.detach().clone())
    for a, b in zip(res[False][0], res[True][0]):                                # (the two models autotune separately: fp32 re-association)
        assert float((a - b).norm() / b.norm()) < 1e-5
    assert all(torch.equal(x, res[True][0][0]) for x in res[True][0][1:])       # native replays = the recorded launches, bit for bit
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=1e-4)          # (float atomics in the warp adjoint)
    assert float((res[True][2] - res[False][2]).abs().max()) < 1e-4


@pytest.mark.parametrize('disturb', ['train_forward', 'copy_out', 'timer'])
def test_inference_tapes_survive_a_pass_that_does_not_bring_its_own_output(monkeypatch, disturb):
    """(advisor r04, high) An inference tape recorded with a caller-owned output keeps its last launch outside the tape.  A later
    pass over the same buffers WITHOUT such an output -- a train forward, the copying path, a per-launch survey -- used to clear
    that launch's closure, and the next replay handed out the plan's buffer still holding that other pass's texels."""
    from nlt_amd.engine import OpTimer
    batch = to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=75))
    fresh = to_device_batch(*O.synth_batch(2, 64, 64, 32, 32, 32, 32, k=2, seed=76))
    a = _model(True)
    for _ in range(3):
        a.call(batch, 'test')
    assert a.plan.tape_replays >= 1
    ref = _model(False, like=a)
    if disturb == 'train_forward':
        with torch.enable_grad():
            a.call(batch, 'train')
    elif disturb == 'copy_out':
        monkeypatch.setenv('NLT_PRED_COPY', '1')
        a.call(batch, 'test')
        monkeypatch.setenv('NLT_PRED_COPY', '0')
    else:
        a.plan.timer = OpTimer()
        a.call(batch, 'test')
        a.plan.timer = None
    for t, s in zip(batch, fresh):                                      # new contents at the SAME addresses: the tape key recurs
        if torch.is_tensor(t):
            t.copy_(s)
    want = ref.call(batch, 'test')
    before = a.plan.tape_replays
    got = a.call(batch, 'test')
    close = lambda x, y: float((x - y).norm() / y.norm()) < 1e-6      # (a stale buffer is another batch's texels: O(1) away)
    assert close(got[3]['pred'], want[3]['pred']) and close(got[0], want[0])
    assert a.plan.tape_replays in (before, before + 1)                # (a replay, unless the disturbing pass grew a workspace)
