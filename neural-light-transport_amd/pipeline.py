"""Throughput mode of the render loop: several batches in flight on one GPU.

The reference renders its test batches one after another (nlt/nlt_test.py:78-94: `for batch in datapipe: model.call(...)`), and
so does `Model.call` here: one forward pass is a chain of ~37 launches whose first (the fused front kernel, alone on the chip
and bound by its own issue latency) and last four (two fused expanding blocks, the head, the resampler: HBM-bound, alone) leave
the other resource idle.  Consecutive batches are independent, so a `RenderPipeline` keeps `lanes` of them in flight: each lane is
a copy of the model's RENDER STATE only (plan buffers, launch tapes, folded front weights, HIP streams) over the one set of
weights and packed fragments, and batch i goes to lane i % lanes on that lane's stream.  The front kernel of one batch then runs
beside the HBM-bound tail (or the middle) of another.  Results are bit-identical to `Model.call` (same kernels, same tile
choices, deterministic reductions; tests/test_gpu_pipeline.py).

    pipe = RenderPipeline(model, lanes=3)
    tickets = [pipe.submit(batch, 'test') for batch in batches]     # returns at once: launches queued on the lane's stream
    pred_camspc, _, _, to_vis = tickets[0].result()                 # the caller's stream waits for that batch only
or  outs = pipe.render(datapipe, 'test', on_batch=...)              # the reference's loop, `lanes` batches in flight

Contract for the inputs: a submitted batch is READ on the lane's stream, so its buffers must stay untouched until that batch's
ticket has been waited for (`result()`), or until `lanes` further batches have been submitted (submit makes the caller's stream
wait for the batch issued `lanes` submissions ago): a staging ring (datasets/nlt.py `ring`) needs lanes + 1 slots.
Weights must not change while batches are in flight (inference); after an update the next submit drains every lane first."""
import collections
import copy

import torch

from .engine import RenderPlan

_PLAN_SWITCHES = ('precision', 'fuse_ends', 'front_l2', 'fuse_dec', 'front_v4', 'two_streams', 'use_tape', 'lds_tn128')


def _copy_tuning(dst, src):
    dst.tile_hints, dst.algo_hints = dict(src.tile_hints), dict(src.algo_hints)
    dst.lds_hints, dst.splitk_hints = dict(src.lds_hints), dict(src.splitk_hints)
    dst._drop_tapes()


class RenderTicket:
    """One submitted batch.  `result()` orders the caller's current stream after the batch and returns what Model.call returns."""

    def __init__(self, out, done, tensors):
        self._out, self._done, self._tensors = out, done, tensors

    def result(self):
        if self._done is not None:
            cur = torch.cuda.current_stream(self._tensors[0].device) if self._tensors else torch.cuda.current_stream()
            cur.wait_event(self._done)
            for t in self._tensors:
                t.record_stream(cur)                 # allocated on the lane's stream, consumed on the caller's
            self._done = None
        return self._out


def _tensors_of(out):
    found = []
    for x in out:
        if torch.is_tensor(x):
            found.append(x)
        elif isinstance(x, dict):
            found.extend(v for v in x.values() if torch.is_tensor(v))
    return [t for t in found if t.is_cuda]


class RenderPipeline:
    def __init__(self, model, lanes=3):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        if getattr(model, 'generic', False) and lanes > 1:
            raise NotImplementedError("the layer-by-layer configs render one batch at a time (their scratch is shared)")
        self.model, self.n = model, lanes
        self._lanes = [model] + [None] * (lanes - 1)        # lane 0 IS the model: its plan tunes, the others copy its choices
        self._streams = [None] * lanes
        self._tuned_ref = [None] * lanes
        self._recent = collections.deque()                  # done-events of the last `lanes` submissions
        self._next = 0
        self._weights = None

    # -------------------------------------------------------------- lanes
    def _lane(self, i):
        m = self.model
        if i == 0:
            return m
        lane = self._lanes[i]
        if lane is None:
            lane = copy.copy(m)                             # same nets, flat bucket, pack registry
            lane.plan = RenderPlan(m.net['query'], m.net['obs'], m.use_obs)
            lane._graph = None
            self._lanes[i] = lane
        lane.conv_algo, lane.skip_connect_base = m.conv_algo, m.skip_connect_base
        if self._tuned_ref[i] is not getattr(m.plan, 'tuned', None) or lane.plan.precision != m.plan.precision:
            for a in _PLAN_SWITCHES:
                setattr(lane.plan, a, getattr(m.plan, a))
            _copy_tuning(lane.plan, m.plan)                 # lane 0's plan-time trials decide for every lane
            self._tuned_ref[i] = getattr(m.plan, 'tuned', None)
        lane.plan.autotune = False
        return lane

    def _weights_version(self):
        fp = getattr(self.model, 'flat_params', None)
        return None if fp is None else (fp._version, getattr(self.model, '_epoch', [0])[0])

    # -------------------------------------------------------------- submit / collect
    def submit(self, batch, mode='test', **kw):
        """Queues Model.call(batch, mode, **kw) on the next lane; returns a RenderTicket without waiting for anything."""
        i = self._next % self.n
        probe = next((t for t in batch if torch.is_tensor(t)), None)
        dev = probe.device if probe is not None else getattr(batch[1], 'cvis', torch.empty(0)).device
        if dev.type != 'cuda':                              # host tests: lanes are a launch-scheduling matter, nothing to overlap
            self._next += 1
            return RenderTicket(self._lane(i).call(batch, mode, **kw), None, [])
        wv = self._weights_version()
        fresh = wv != self._weights or (i > 0 and self._lanes[i] is None)
        if fresh:
            # first batch of this lane / the weights changed: whatever is made once and shared (packed fragments, lane 0's
            # plan-time trials) is made on ONE stream with nothing else in flight, and complete before another lane reads it
            torch.cuda.synchronize(dev)
            self._weights = wv
        lane = self._lane(i)
        if self._streams[i] is None:
            self._streams[i] = torch.cuda.Stream(device=dev)
        stream, cur = self._streams[i], torch.cuda.current_stream(dev)
        if len(self._recent) >= self.n:
            cur.wait_event(self._recent.popleft())          # the inputs of the batch `lanes` submissions ago are free again
        stream.wait_stream(cur)                             # this batch was assembled on the caller's stream
        with torch.cuda.stream(stream):
            out = lane.call(batch, mode, **kw)
            done = torch.cuda.Event()
            done.record(stream)
        if fresh:
            torch.cuda.synchronize(dev)
        self._recent.append(done)
        self._next += 1
        return RenderTicket(out, done, _tensors_of(out))

    def render(self, datapipe, mode='test', on_batch=None, **kw):
        """The render loop with `lanes` batches in flight: list of Model.call results in order (or each handed to
        on_batch(i, result) as soon as the pipeline has to wait for it anyway)."""
        outs, pending = [], collections.deque()

        def collect():
            j, t = pending.popleft()
            r = t.result()
            if on_batch is not None:
                on_batch(j, r)
            else:
                outs.append(r)
        for i, batch in enumerate(datapipe):
            pending.append((i, self.submit(batch, mode, **kw)))
            if len(pending) > self.n:
                collect()
        while pending:
            collect()
        return outs
