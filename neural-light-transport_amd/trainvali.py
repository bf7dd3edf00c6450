"""Train / validation steps with the reference's semantics (nlt/trainvali.py:267-325), one process
per GPU: per-example loss -> sum / GLOBAL batch size -> backward -> ONE all-reduce(sum) of the flat
fp32 gradient bucket over RCCL/xGMI -> fused Adam-AMSGrad (bit-identical on every rank) -> scalar
loss all-reduce(sum).  MirroredStrategy's in-process replicas become torch.distributed ranks."""
import torch
import torch.distributed as dist

from . import optim


def make_optimizer(model, config):
    """nlt/trainvali.py:122-127."""
    lr = config.getfloat('DEFAULT', 'lr')
    mgm = config.getfloat('DEFAULT', 'mgm')
    return optim.AdamAMSGrad(model, lr, clipnorm=mgm if mgm > 0 else None)


def _world(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def distributed_train_step(model, batch, optimizer, global_bs, group=None):
    """batch = this rank's shard of the global batch.  Returns (loss summed over ranks, to_vis)."""
    assert model.trainable_registered, "Register the trainable layers before using `trainable_variables`"
    pred, gt, loss_kwargs, to_vis = model(batch, mode='train')
    loss_kwargs['keep_batch'] = True
    per_example_loss = model.compute_loss(pred, gt, **loss_kwargs)
    weighted_loss = per_example_loss.sum() / global_bs             # tf.nn.compute_average_loss
    model.flat_params.grad = None
    weighted_loss.backward()
    grad = model.flat_params.grad
    loss = weighted_loss.detach().clone()
    if _world(group) > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)  # the one data-path collective
        dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group)
    optimizer.step(grad)
    return loss, to_vis


def distributed_vali_step(model, batch, global_bs, group=None):
    with torch.no_grad():
        pred, gt, loss_kwargs, to_vis = model(batch, mode='vali')
        loss_kwargs['keep_batch'] = True
        loss = model.compute_loss(pred, gt, **loss_kwargs).sum() / global_bs
        if _world(group) > 1:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=group)
    return loss, to_vis
