"""Keras Adam(amsgrad=True) as TF 2.2 executes it, fused over the model's flat parameter bucket
(replaces tf.keras.optimizers.Adam + apply_gradients, nlt/trainvali.py:122-127,280)."""
import math

import torch

from . import _capi as C


class AdamAMSGrad:
    def __init__(self, model, lr, beta1=0.9, beta2=0.999, eps=1e-7, clipnorm=None):
        # clipnorm (config key mgm > 0, nlt/trainvali.py:122-127): Keras' per-variable tf.clip_by_norm, applied here to the
        # (all-reduced) gradient of every kernel / bias right before the Adam update.  The released configs use mgm = -1.
        self.clipnorm = float(clipnorm) if clipnorm is not None and clipnorm > 0 else None
        self._slots = None
        self.model, self.lr, self.b1, self.b2, self.eps = model, lr, beta1, beta2, eps
        self.t = 0
        z = lambda: torch.zeros_like(model.flat_params, requires_grad=False)
        self.m, self.v, self.vhat = z(), z(), z()

    def step(self, grad=None):
        """grad: flat gradient bucket (defaults to model.flat_params.grad, i.e. what backward() left)."""
        if grad is None:
            grad = self.model.flat_params.grad
        if self.clipnorm is not None:
            if self._slots is None:                      # (offset, count) of every variable inside the flat bucket
                base = self.model.flat_grads.data_ptr()
                rows = [((v.data_ptr() - base) // 4, v.numel()) for c in self.model._conv_layers() for v in (c.dkernel, c.dbias)]
                self._slots = torch.tensor(rows, dtype=torch.int64, device=grad.device)
            C.clip_by_norm_slots(grad, self._slots, self.clipnorm)
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        C.adam_amsgrad_step(self.model.flat_params.detach(), grad, self.m, self.v, self.vhat, lr_t, self.b1, self.b2,
                            self.eps)
        self.model.mark_weights_updated()

    def state_dict(self):
        """Per VARIABLE in canonical order (Model.bucket_to_variables), like the net's own state: the flat bucket's slot order is
        a tuning choice of this process and must not leak into files."""
        m = self.model
        return {'t': self.t, 'm': m.bucket_to_variables(self.m), 'v': m.bucket_to_variables(self.v),
                'vhat': m.bucket_to_variables(self.vhat)}

    def load_state_dict(self, sd):
        m = self.model
        for k in ('m', 'v', 'vhat'):
            if not isinstance(sd.get(k), (list, tuple)):
                raise ValueError("optimizer state '%s' is not a per-variable list (format nlt_amd-ckpt-2)" % k)
            if len(sd[k]) != len(m._slots) or any(tuple(t.shape) != tuple(shp) for t, (_, _, shp) in zip(sd[k], m._slots)):
                raise ValueError("optimizer state '%s' does not match this model's variables" % k)
        self.t = int(sd['t'])
        for k in ('m', 'v', 'vhat'):
            m.variables_to_bucket(list(sd[k]), getattr(self, k))
