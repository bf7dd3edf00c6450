"""Losses with the reference's call signatures (nlt/losses.py:39-53,90-118) on HIP kernels
(csrc/train_ops.hip, csrc/barron.hip); torch.autograd.Function is glue only."""
import torch

from . import _capi as C


class _L2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        pred, gt = pred.contiguous(), gt.contiguous()
        ctx.save_for_backward(pred, gt)
        return C.l2_loss_forward(pred, gt)

    @staticmethod
    def backward(ctx, gloss):
        pred, gt = ctx.saved_tensors
        return C.l2_loss_backward(pred, gt, gloss.contiguous()), None


class _BarronFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        loss, dunit = C.barron_loss(pred.contiguous(), gt.contiguous(), bool(ctx.needs_input_grad[0]))
        ctx.dunit = dunit
        return loss

    @staticmethod
    def backward(ctx, gloss):
        return C.scale_rows(ctx.dunit, gloss.contiguous()), None


class L2:
    """MeanSquaredError over channels, then mean over H,W: per-example [N] with keep_batch=True
    (what the train step injects, trainvali.py:275), scalar otherwise."""

    def __call__(self, gt, pred, keep_batch=False, weights=None):
        if weights is not None:
            raise NotImplementedError("sample weights")
        per = _L2Fn.apply(pred, gt)
        return per if keep_batch else per.mean()


class Barron:
    """robust_loss AdaptiveImageLossFunction as NLT fixes it: alpha = 1 (Charbonnier), scale = 0.01,
    sYUV, CDF9/7, 5 levels, wavelet_scale_base = 1 -- no trainable variables."""

    def __init__(self, imw, imh):
        self.imw, self.imh = imw, imh

    def __call__(self, gt, pred, keep_batch=False, weights=None):
        if weights is not None:
            raise NotImplementedError("alpha-blended weights")
        assert tuple(pred.shape[1:3]) == (self.imh, self.imw), (tuple(pred.shape), self.imh, self.imw)
        per = _BarronFn.apply(pred, gt)
        return per if keep_batch else per.mean()
