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


class _L2WeightedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, weights):
        pred, gt = pred.contiguous(), gt.contiguous()
        ctx.save_for_backward(pred, gt, weights)
        return C.l2_loss_weighted_forward(pred, gt, weights)

    @staticmethod
    def backward(ctx, gloss):
        pred, gt, weights = ctx.saved_tensors
        return C.l2_loss_weighted_backward(pred, gt, weights, gloss.contiguous()), None, None


class _MulFn(torch.autograd.Function):
    """imgutil.alpha_blend(x, alpha) with no second tensor (nlt/util/img.py:74-89): x * alpha + 0 * (1 - alpha)."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.save_for_backward(alpha)
        return C.mul_forward(x.contiguous(), alpha)

    @staticmethod
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        return C.mul_forward(g.contiguous(), alpha), None


def _sample_weight_map(weights, like):
    """Keras `sample_weight` against the per-texel loss map [N,H,W] (losses_utils.compute_weighted_loss: a trailing axis
    of 1 is squeezed, a missing trailing axis added, the rest must broadcast rank for rank) -> dense float32 [N,H,W]."""
    n, h, w = like.shape[:3]
    wt = torch.as_tensor(weights, dtype=torch.float32, device=like.device)
    if wt.dim() == 4 and wt.shape[-1] == 1:
        wt = wt[..., 0]
    elif wt.dim() == 2:
        wt = wt[..., None]
    if wt.dim() not in (0, 3):
        raise ValueError("sample weights of shape %s do not broadcast to the loss map %s" % (tuple(wt.shape), (n, h, w)))
    return wt.expand(n, h, w).contiguous()


class L2:
    """MeanSquaredError over channels, then mean over H,W: per-example [N] with keep_batch=True
    (what the train step injects, trainvali.py:275), scalar otherwise."""

    def __call__(self, gt, pred, keep_batch=False, weights=None):
        if weights is not None:                 # Keras sample_weight on the [N,H,W] loss map (nlt/losses.py:42-43)
            per = _L2WeightedFn.apply(pred, gt, _sample_weight_map(weights, pred))
            return per if keep_batch else per.mean()
        per = _L2Fn.apply(pred, gt)
        return per if keep_batch else per.mean()


class Barron:
    """robust_loss AdaptiveImageLossFunction as NLT fixes it: alpha = 1 (Charbonnier), scale = 0.01,
    sYUV, CDF9/7, 5 levels, wavelet_scale_base = 1 -- no trainable variables."""

    def __init__(self, imw, imh):
        self.imw, self.imh = imw, imh

    def __call__(self, gt, pred, keep_batch=False, weights=None):
        if weights is not None:                 # nlt/losses.py:107-110: gt and pred alpha-blended against zeros
            alpha = torch.as_tensor(weights, dtype=torch.float32, device=pred.device).expand(pred.shape).contiguous()
            gt, pred = _MulFn.apply(gt, alpha), _MulFn.apply(pred, alpha)
        assert tuple(pred.shape[1:3]) == (self.imh, self.imw), (tuple(pred.shape), self.imh, self.imw)
        per = _BarronFn.apply(pred, gt)
        return per if keep_batch else per.mean()
