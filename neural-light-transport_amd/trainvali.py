"""Train / validation steps with the reference's semantics (nlt/trainvali.py:267-325), one process
per GPU: per-example loss -> sum / GLOBAL batch size -> backward -> ONE all-reduce(sum) of the flat
fp32 gradient bucket over RCCL/xGMI -> fused Adam-AMSGrad (bit-identical on every rank) -> scalar
loss all-reduce(sum).  MirroredStrategy's in-process replicas become torch.distributed ranks."""
import os

import torch
import torch.distributed as dist

from . import optim


def make_optimizer(model, config):
    """nlt/trainvali.py:122-127: `Adam(lr, amsgrad=True, clipnorm=mgm)` when mgm > 0.

    What `clipnorm` DOES in the reference's loop is a property of the pinned TensorFlow 2.2.0 (environment.yml:14), not of
    the reference's code: the loop calls `tape.gradient` + `optimizer.apply_gradients` (trainvali.py:279-280), and
    OptimizerV2 2.2 clips only inside `get_gradients` / `_compute_gradients` (the `minimize` path); `apply_gradients` takes
    the gradients as given.  So, as far as can be established without a TF 2.2 install, mgm > 0 changes NOTHING in the
    reference's training.  Default here = that run-time behaviour (no clipping, one warning).  The per-variable
    `tf.clip_by_norm` a reader of the config would expect is built (nlt_clip_by_norm_slots) and OPT-IN: config key
    `mgm_apply = true` (not a reference key) or NLT_APPLY_CLIPNORM=1.  The released configs set mgm = -1 either way."""
    lr = config.getfloat('DEFAULT', 'lr')
    mgm = config.getfloat('DEFAULT', 'mgm')
    apply_clip = (config.getboolean('DEFAULT', 'mgm_apply', fallback=False)
                  or os.environ.get('NLT_APPLY_CLIPNORM', '0') == '1')
    if mgm > 0 and not apply_clip:
        import warnings
        warnings.warn("mgm = %g: TF 2.2's apply_gradients (the reference's train loop) does not apply clipnorm; not clipping. "
                      "Set mgm_apply = true (or NLT_APPLY_CLIPNORM=1) for per-variable tf.clip_by_norm." % mgm)
    return optim.AdamAMSGrad(model, lr, clipnorm=mgm if (mgm > 0 and apply_clip) else None)


CKPT_FORMAT = 'nlt_amd-ckpt-2'


def save_checkpoint(path, model, optimizer, step):
    """`tf.train.Checkpoint(step=, optimizer=, net=)` + `CheckpointManager.save` of the reference train loop
    (nlt/trainvali.py:134-141,197): weights, Adam-AMSGrad slots (m, v, vhat) and iteration count, global step -- enough to
    resume training bit for bit, and what `nlt_test`-style inference restores.  One torch.save file (tensors on CPU).
    Format 2 (r06): every kernel / bias and every optimizer slot is stored PER VARIABLE in canonical order; format 1 stored
    the raw flat bucket, whose slot order depends on the writer's `_flatten` (advisor r05: a file written under another
    NLT_GRAD_RANGES loaded without error and with permuted weights)."""
    cpu = lambda x: ([t.detach().cpu() for t in x] if isinstance(x, (list, tuple)) else (x.detach().cpu() if torch.is_tensor(x) else x))
    cpud = lambda d: {k: cpu(v) for k, v in d.items()}
    torch.save({'format': CKPT_FORMAT, 'step': int(step), 'net': cpud(model.state_dict()),
                'optimizer': cpud(optimizer.state_dict()) if optimizer is not None else None}, path)
    return path


def _convert_ckpt1(ck, model):
    """A format-1 file (raw flat buckets) -> format-2 dicts, cutting the buckets by THIS process's slot table."""
    net = ck['net']
    shapes = [(tuple(c.kernel.shape), tuple(c.bias.shape)) for c in model._conv_layers()]
    if [tuple(map(tuple, x)) for x in net['slots']] != shapes or net['flat_params'].numel() != model.flat_params.numel():
        raise ValueError("checkpoint was written by a different architecture (layer shapes differ)")
    cut = lambda flat: [flat.reshape(-1)[o:o + n].reshape(shp).clone() for o, n, shp in model.legacy_bucket_layout()]
    out = {'net': {'variables': cut(net['flat_params'])}, 'optimizer': None}
    if ck.get('optimizer') is not None:
        o = ck['optimizer']
        out['optimizer'] = {'t': o['t'], 'm': cut(o['m']), 'v': cut(o['v']), 'vhat': cut(o['vhat'])}
    return out


def restore_checkpoint(path, model, optimizer=None, legacy_layout=False):
    """Loads what `save_checkpoint` wrote into a built model (and optimizer); returns the global step.  The optimizer may be
    omitted (inference: nlt/nlt_test.py:92-97 restores the net alone).  A format-1 file (rounds 1-5: raw flat buckets) is
    refused unless `legacy_layout=True` states that it was written with the bucket layout this process uses (same code
    generation, same NLT_GRAD_RANGES) -- the file itself cannot tell."""
    ck = torch.load(path, map_location='cpu', weights_only=True)     # tensors, ints, strings, lists only: no pickle code runs
    fmt = ck.get('format')
    if fmt == 'nlt_amd-ckpt-1':
        if not legacy_layout:
            raise ValueError("%s is a format-1 checkpoint (raw flat parameter bucket): its slot order depends on the writer's "
                             "bucket layout and is not recorded in the file.  Pass legacy_layout=True if it was written by the "
                             "same code generation with the same NLT_GRAD_RANGES, then save it again (format 2)." % path)
        ck = dict(ck, **_convert_ckpt1(ck, model))
    elif fmt != CKPT_FORMAT:
        raise ValueError("%s is not an nlt_amd checkpoint" % path)
    model.load_state_dict(ck['net'])
    if optimizer is not None:
        if ck['optimizer'] is None:
            raise ValueError("%s holds no optimizer state" % path)
        optimizer.load_state_dict(ck['optimizer'])
    return ck['step']


AUTOGRAD_STEP = os.environ.get('NLT_AUTOGRAD_STEP', '0') == '1'   # single-rank step through torch.autograd (the reference-shaped path)


def _world(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def distributed_train_step(model, batch, optimizer, global_bs, group=None, overlap=True):
    """batch = this rank's shard of the global batch.  Returns (loss summed over ranks, to_vis).

    One rank: per-example loss -> sum / global batch -> autograd backward -> fused Adam.
    Several ranks: the gradient sum is THREE all-reduces of fixed contiguous ranges of the flat bucket, cut in the order
    the backward pass finishes them (`Model.bucket_ranges`: expanding blocks | encoder levels D .. 3 | the rest), so every
    rank adds the same numbers in the same order: bit-identical weights.  The first two are issued from inside the backward
    plan as soon as their weight-gradient launches are queued (on their stream) and run over xGMI while the rest of the
    backward occupies the CUs; only the last (0.1 MB of 13.5) follows the backward.  overlap=False issues all three after
    the backward (same result)."""
    assert model.trainable_registered, "Register the trainable layers before using `trainable_variables`"
    world = _world(group)
    if world == 1 and not AUTOGRAD_STEP and not getattr(model, 'generic', False):     # (branch configs run layer by layer through autograd)
        # same arithmetic without a torch.autograd graph around the network (only the loss is differentiated): no AccumulateGrad
        # copy of the 13.5 MB bucket, no expand / fill launches for the sum and the division
        loss, to_vis = model.train_forward_backward(batch, global_bs)
        model.flat_params.grad = model.flat_grads
        optimizer.step(model.flat_grads)
        return loss.clone(), to_vis
    if world == 1:
        pred, gt, loss_kwargs, to_vis = model(batch, mode='train')
        loss_kwargs['keep_batch'] = True
        per_example_loss = model.compute_loss(pred, gt, **loss_kwargs)
        weighted_loss = per_example_loss.sum() / global_bs         # tf.nn.compute_average_loss
        model.flat_params.grad = None
        weighted_loss.backward()
        grad = model.flat_params.grad
        optimizer.step(grad)
        return weighted_loss.detach().clone(), to_vis
    if getattr(model, 'generic', False):
        # the layer-by-layer branch configs (elu / pixel, layer, batch norm / pooling): the same step, the network differentiated
        # through its hand-rolled tape (generic.py) into the same flat bucket, then ONE all-reduce of the whole bucket (no
        # backward plan to overlap with).  Batch norm needs no cross-replica statistics: the reference runs it in inference
        # mode (networks/elements.py ChannelNorm).
        pred, gt, loss_kwargs, to_vis = model(batch, mode='train')
        loss_kwargs['keep_batch'] = True
        loss = model.compute_loss(pred, gt, **loss_kwargs).sum() / global_bs
        model.flat_params.grad = None
        loss.backward()
        grad = model.flat_params.grad
        loss = loss.detach().clone()
        works = [dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group, async_op=True),
                 dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group, async_op=True)]
        for w in works:
            w.wait()
        optimizer.step(grad)
        return loss, to_vis
    grad, r = model.flat_grads, model.bucket_ranges
    works, sent = [], set()

    def reduce_range(i):
        if i not in sent and r[i + 1] > r[i]:
            works.append(dist.all_reduce(grad[r[i]:r[i + 1]], op=dist.ReduceOp.SUM, group=group, async_op=True))
        sent.add(i)
    if overlap:
        model.plan.grad_hook = reduce_range
    try:
        loss, to_vis = model.train_forward_backward(batch, global_bs)    # gradients land in the flat bucket (no autograd copy)
    finally:
        model.plan.grad_hook = None
    for i in range(len(r) - 1):                                         # whatever the backward did not send, in range order
        reduce_range(i)
    loss = loss.clone()
    works.append(dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()                                                    # (CUDA: the current stream waits; the host does not block)
    model.flat_params.grad = grad
    optimizer.step(grad)
    return loss, to_vis


class GraphedTrainStep:
    """`distributed_train_step` with forward + loss + backward captured ONCE as a hipGraph and replayed: the step is
    ~170 small launches, and issuing them one by one from Python costs about as much wall time as the GPU needs to
    run them.  The gradient all-reduce (RCCL) and the Adam-AMSGrad launch (its bias-correction scalars change every
    step) stay eager, after the replay.

    The graph is tied to tensor ADDRESSES: the batch is copied into static buffers owned by this object (a no-op when
    the caller already passes them back), shapes must not change, and the returned loss / to_vis tensors are the
    graph's static outputs -- consume them before the next call.  The first `warmup` calls run eagerly (plan-time
    autotune, workspace allocation); any failure to capture falls back to the eager step for good."""

    TENSORS = (1, 2, 3, 4, 5, 6, 8, 9, 10)       # tensor entries of the 11-tuple batch (nlt/models/nlt.py:91-92)

    def __init__(self, model, optimizer, global_bs, group=None, warmup=2):
        self.model, self.optimizer, self.global_bs, self.group = model, optimizer, global_bs, group
        self.warmup, self.seen = warmup, 0
        self.graph, self.static, self.out, self.failed = None, None, None, None
        self._table = self._reg_version = None       # the registry's descriptor table the graph embeds, and the buffer-set version it was captured at

    def static_batch(self):
        """The graph's own input tensors (None before the capture): a loader that fills THESE and passes them back
        skips the per-step device-to-device copies."""
        return tuple(self.static) if self.graph is not None else None

    def _body(self, batch):
        loss, to_vis = self.model.train_forward_backward(batch, self.global_bs)
        return loss.clone(), to_vis

    def _finish(self, loss):
        grad = self.model.flat_grads
        self.model.flat_params.grad = grad
        if _world(self.group) > 1:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
        self.optimizer.step(grad)
        return loss

    def __call__(self, batch):
        dev_ok = isinstance(batch[1], torch.Tensor) and batch[1].is_cuda and self.model.plan.timer is None
        if self.failed is not None or not dev_ok or self.seen < self.warmup:
            self.seen += 1
            return distributed_train_step(self.model, batch, self.optimizer, self.global_bs, self.group)
        reg = getattr(self.model, 'pack_registry', None)
        if self.graph is not None and reg is not None and reg.version != self._reg_version:
            # the SET of packed buffers changed after the capture (a census retired some, a plan re-tuned: advisor r05): the
            # captured repack launch would walk a descriptor table the registry no longer maintains -- capture again
            self.graph = self.out = None
        if self.graph is None:
            try:
                self.static = list(batch)
                for i in self.TENSORS:
                    self.static[i] = batch[i].clone()
                reg = getattr(self.model, 'pack_registry', None)
                if reg is not None:                      # the refresh of the packed weights must be IN the graph, its
                    reg.prune()                          # (a running census of used fragment buffers ends here, not mid-capture)
                    reg.prepare()                        # descriptor upload must not
                    reg.state = None
                if self.model.plan._front_blob is not None:
                    self.model.plan._front_blob[0] = None    # ... and so must the folded front-kernel weights
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.out = self._body(tuple(self.static))
                self.graph = graph
                if reg is not None:
                    self._table, self._reg_version = reg.table, reg.version   # (keeps the captured table's tensors alive)
            except Exception as e:                       # capture is an optimisation, never a requirement
                self.failed = repr(e)
                self.graph = None
                torch.cuda.synchronize()
                return distributed_train_step(self.model, batch, self.optimizer, self.global_bs, self.group)
        for i in self.TENSORS:
            if batch[i].data_ptr() != self.static[i].data_ptr():
                self.static[i].copy_(batch[i])
        self.graph.replay()
        loss, to_vis = self.out
        return self._finish(loss.clone()), to_vis


def distributed_vali_step(model, batch, global_bs, group=None):
    with torch.no_grad():
        pred, gt, loss_kwargs, to_vis = model(batch, mode='vali')
        loss_kwargs['keep_batch'] = True
        loss = model.compute_loss(pred, gt, **loss_kwargs).sum() / global_bs
        if _world(group) > 1:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group)
    return loss, to_vis
