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

Host side: a forward pass is ~40-60 HIP runtime calls of ~8 us each, i.e. 0.3-0.5 ms of host time per batch -- for the released
shapes (512^2 / 256^2 UV, 4 frames) as long as the GPU needs for the batch, so lanes driven from ONE thread stay host-bound.
With threads=True every lane but the first has its own host thread that replays the lane's launch tape through
nlt_tape_play (one C call per run of launches, interpreter lock released), so the lanes enqueue in parallel.

graphs=True (the released small shapes: 512^2 / 256^2 UV at 4 frames, where even the launch-tape replay is host-bound): every
lane runs its forward on ONE stream and replays it as a hipGraph per set of input addresses (Model.use_graphs; a single-stream
chain replays for ~0.07 ms of host time where the two-stream graph and the launch tape both cost 0.3-0.45 ms), and hands out
copies of the graph's static outputs.  Measured (r03, MI355X, 4 frames per batch): depth 1024 / 256^2 551 -> 1228 Mtexels/s,
depth 256 / 512^2 2247 -> 3832 Mtexels/s at 4 lanes; the 1024^2 headline shape is GPU-bound and stays on eager launches.

Contract for the inputs: a submitted batch is READ on the lane's stream.  Its ticket holds a reference to it until `result()`
and its tensors are `record_stream`ed on the lane's stream, so a loader that hands out fresh tensors per call (tf.data style,
`ring = 0`) may drop them at once -- the caching allocator will not reuse their blocks early.  What the caller must not do is
OVERWRITE a submitted batch in place before its ticket has been waited for or `lanes` further batches have been submitted
(submit makes the caller's stream wait for the batch issued `lanes` submissions ago): a staging ring (datasets/nlt.py `ring`)
needs lanes + 1 slots.
Weights must not change while batches are in flight (inference); after an update the next submit drains every lane first.
Lane 0 is the model itself (except with graphs=True): do not call the model directly while tickets are outstanding."""
import collections
import copy
import queue
import threading

import torch

from . import _capi as C
from .engine import RenderPlan

_PLAN_SWITCHES = ('precision', 'fuse_ends', 'front_l2', 'fuse_dec', 'front_v4', 'two_streams', 'use_tape', 'use_wino', 'use_c32', 'fuse_override', 'alias_obs')


def _copy_tuning(dst, src):
    dst.tile_hints, dst.algo_hints = dict(src.tile_hints), dict(src.algo_hints)
    dst.lds_hints, dst.splitk_hints = dict(src.lds_hints), dict(src.splitk_hints)
    dst.wino_hints = dict(src.wino_hints)
    dst.c32_hints = dict(src.c32_hints)
    dst._drop_tapes()


class RenderTicket:
    """One submitted batch.  `result()` orders the caller's current stream after the batch and returns what Model.call returns."""

    def __init__(self, batch=None):
        self._out = self._done = self._exc = None
        self._tensors = []
        self._batch = batch                          # kept alive until the batch has been waited for: the lane READS it
        self._queued = threading.Event()             # set once the lane's host thread has issued every launch of the batch

    def _set(self, out, done, exc=None):
        self._out, self._done, self._exc = out, done, exc
        self._tensors = _tensors_of(out) if out is not None else []
        self._queued.set()

    def _done_event(self):
        self._queued.wait()
        return self._done

    def result(self):
        self._queued.wait()
        if self._exc is not None:
            raise self._exc
        if self._done is not None:
            cur = torch.cuda.current_stream(self._tensors[0].device) if self._tensors else torch.cuda.current_stream()
            cur.wait_event(self._done)
            for t in self._tensors:
                t.record_stream(cur)                 # allocated on the lane's stream, consumed on the caller's
            self._done = None
        self._batch = None
        return self._out


def _own_copies(out, batch):
    """Model.call's result with every device tensor that is not one of the batch's own buffers cloned."""
    inputs = {id(t) for t in batch if torch.is_tensor(t)}
    cp = lambda x: x.clone() if (torch.is_tensor(x) and x.is_cuda and id(x) not in inputs) else x
    return tuple({k: cp(v) for k, v in x.items()} if isinstance(x, dict) else cp(x) for x in out)


def _batch_tensors(batch):
    """Every CUDA tensor a submitted batch is made of (the 11-tuple's tensors; a store-resident batch's id / map tensors)."""
    found = []
    for x in batch:
        if torch.is_tensor(x):
            found.append(x)
        elif hasattr(x, '__dict__'):
            found.extend(v for v in vars(x).values() if torch.is_tensor(v))
    return [t for t in found if t.is_cuda]


def _tensors_of(out):
    found = []
    for x in out:
        if torch.is_tensor(x):
            found.append(x)
        elif isinstance(x, dict):
            found.extend(v for v in x.values() if torch.is_tensor(v))
    return [t for t in found if t.is_cuda]


class RenderPipeline:
    def __init__(self, model, lanes=3, threads=False, graphs=False):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        if getattr(model, 'generic', False) and lanes > 1:
            raise NotImplementedError("the layer-by-layer configs render one batch at a time (their scratch is shared)")
        self.model, self.n = model, lanes
        self._lanes = [model] + [None] * (lanes - 1)        # lane 0 IS the model: its plan tunes, the others copy its choices
        self._streams = [None] * lanes
        self._tuned_ref = [None] * lanes
        self._switches = [None] * lanes                     # the model plan's switches / hints each lane last copied
        self._hints = [({}, {}, {}, {}, {}, {})] * lanes
        self._recent = collections.deque()                  # tickets of the last `lanes` submissions
        self._graphs = bool(graphs)
        self._threads = bool(threads) and lanes > 1 and not self._graphs      # (stream capture wants the other host threads quiet)
        if self._graphs:
            self._lanes[0] = None                           # every lane a copy: the caller's model keeps its own launch mode
        self._workers = [None] * lanes                      # (thread, job queue) per lane, started on first use
        self._next = 0
        self._weights = None

    # -------------------------------------------------------------- lanes
    def _lane(self, i):
        m = self.model
        if i == 0 and not self._graphs:
            return m
        lane = self._lanes[i]
        new = lane is None
        if new:
            lane = copy.copy(m)                             # same nets, flat bucket, pack registry
            lane.plan = RenderPlan(m.net['query'], m.net['obs'], m.use_obs)
            lane._graph, lane._graphs = None, {}
            lane.use_graphs = self._graphs
            self._lanes[i] = lane
        lane.conv_algo, lane.skip_connect_base = m.conv_algo, m.skip_connect_base
        sw = tuple(getattr(m.plan, a) for a in _PLAN_SWITCHES)
        hints = (m.plan.tile_hints, m.plan.algo_hints, m.plan.lds_hints, m.plan.splitk_hints, m.plan.wino_hints, m.plan.c32_hints)
        if (new or self._tuned_ref[i] is not getattr(m.plan, 'tuned', None) or self._switches[i] != sw
                or any(dict(a) != b for a, b in zip(hints, self._hints[i]))):
            for a in _PLAN_SWITCHES:                        # the model plan's switches and choices, programmatic ones included:
                setattr(lane.plan, a, getattr(m.plan, a))   # a lane issues exactly the launches Model.call would
            _copy_tuning(lane.plan, m.plan)                 # lane 0's plan-time trials decide for every lane
            self._tuned_ref[i] = getattr(m.plan, 'tuned', None)
            self._switches[i] = sw
            self._hints[i] = tuple(dict(a) for a in hints)
        if self._graphs:
            lane.plan.two_streams = False                   # a linear chain: the cheap kind of graph to launch
        lane.plan.autotune = False
        return lane

    def _weights_version(self):
        fp = getattr(self.model, 'flat_params', None)
        return None if fp is None else (fp._version, getattr(self.model, '_epoch', [0])[0])

    # -------------------------------------------------------------- host threads
    def _worker(self, i, dev):
        w = self._workers[i]
        if w is None:
            jobs = queue.SimpleQueue()

            def loop():
                torch.cuda.set_device(dev)
                C.set_thread_native_replay(True)
                while True:
                    job = jobs.get()
                    if job is None:
                        return
                    self._run(*job)
            th = threading.Thread(target=loop, name='nlt-lane-%d' % i, daemon=True)
            th.start()
            w = self._workers[i] = (th, jobs)
        return w

    def _run(self, lane, stream, ready, batch, mode, kw, ticket):
        try:
            with torch.no_grad(), torch.cuda.stream(stream):
                if ready is not None:
                    stream.wait_event(ready)                # the batch was assembled on the caller's stream
                for t in _batch_tensors(batch):
                    t.record_stream(stream)                 # a loader that frees the batch early: its blocks are not handed out
                out = lane.call(batch, mode, **kw)          # again before this lane has read them
                if self._graphs:
                    out = _own_copies(out, batch)           # the graph's static outputs are rewritten by its next replay
                done = torch.cuda.Event()
                done.record(stream)
            ticket._set(out, done)
        except BaseException as e:                          # surfaces in ticket.result()
            ticket._set(None, None, e)

    def drain(self):
        """Host-side wait until every submitted batch has been fully enqueued (not executed)."""
        for t in list(self._recent):
            t._queued.wait()

    def close(self):
        for i, w in enumerate(self._workers):
            if w is not None:
                w[1].put(None)
                w[0].join(timeout=10)
                self._workers[i] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # -------------------------------------------------------------- submit / collect
    def submit(self, batch, mode='test', **kw):
        """Queues Model.call(batch, mode, **kw) on the next lane; returns a RenderTicket without waiting for the GPU."""
        if mode == 'train':
            raise ValueError("RenderPipeline renders ('test' / 'vali'); a train step needs the previous step's weights")
        i = self._next % self.n
        probe = next((t for t in batch if torch.is_tensor(t)), None)
        dev = probe.device if probe is not None else getattr(batch[1], 'cvis', torch.empty(0)).device
        ticket = RenderTicket(batch)
        if dev.type != 'cuda':                              # host tests: lanes are a launch-scheduling matter, nothing to overlap
            self._next += 1
            ticket._set(self._lane(i).call(batch, mode, **kw), None)
            return ticket
        wv = self._weights_version()
        fresh = wv != self._weights or ((i > 0 or self._graphs) and self._lanes[i] is None)
        if fresh:
            # first batch of this lane / the weights changed: whatever is made once and shared (packed fragments, lane 0's
            # plan-time trials) is made on ONE stream with nothing else in flight, and complete before another lane reads it
            self.drain()
            torch.cuda.synchronize(dev)
            self._weights = wv
            if self._graphs and self.model.plan.autotune and not getattr(self.model.plan, 'tuned', None):
                with torch.no_grad():
                    self.model.call(batch, mode, **kw)      # the model's own plan runs its plan-time trials; the lanes copy them
                torch.cuda.synchronize(dev)
        lane = self._lane(i)
        if self._streams[i] is None:
            self._streams[i] = torch.cuda.Stream(device=dev)
        stream, cur = self._streams[i], torch.cuda.current_stream(dev)
        if len(self._recent) >= self.n:
            old = self._recent.popleft()._done_event()      # (host: that batch's launches are all queued -- one job per lane)
            if old is not None:
                cur.wait_event(old)                         # the inputs of the batch `lanes` submissions ago are free again
        if fresh or not self._threads or i == 0:            # lane 0 is driven by the caller's thread
            stream.wait_stream(cur)
            C.set_thread_native_replay(self._threads)
            try:
                self._run(lane, stream, None, batch, mode, kw, ticket)
            finally:
                C.set_thread_native_replay(False)
            if fresh:
                torch.cuda.synchronize(dev)
        else:
            ready = torch.cuda.Event()
            ready.record(cur)
            self._worker(i, dev)[1].put((lane, stream, ready, batch, mode, kw, ticket))
        self._recent.append(ticket)
        self._next += 1
        return ticket

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
        try:
            for i, batch in enumerate(datapipe):
                pending.append((i, self.submit(batch, mode, **kw)))     # (the ticket keeps its batch alive until collected)
                if len(pending) > self.n:
                    collect()
            while pending:
                collect()
        finally:
            if pending:                                     # a failing batch: let the queued ones finish before their inputs go
                self.drain()
                pending.clear()
        return outs
