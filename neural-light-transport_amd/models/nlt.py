"""NLT model on libnlt_hip.so -- same surface as reference nlt/models/nlt.py:38-205
(`Model(config)`, `.net['query'|'obs']`, `call(batch, mode, obs_override)`, `_call`,
`compute_loss`); tensors are torch CUDA tensors instead of TF eager tensors.
"""
import os
import pickle
import warnings
from glob import glob
from os.path import dirname, exists, join

import numpy as np
import torch

from .. import _capi as C
from .. import losses
from .. import metric
from ..datasets.nlt import ResidentTexels
from ..engine import RenderPlan
from ..networks import convnet
from ..util import vis as V
from .base import Model as BaseModel


def _convs_of(layer):
    """The Conv2D objects of one Network.layers entry (a bare conv or a Sequential, nested ones included)."""
    return [layer] if hasattr(layer, 'set_weights') else layer.all_convs()


class _GenericFn(torch.autograd.Function):
    """autograd glue for the layer-by-layer path (generic.py): same contract as _RenderFn."""

    @staticmethod
    def forward(ctx, flat_params, model, inputs, want_indices):
        base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights = inputs
        out = model._render_generic(base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights, None, want_indices, record=True)
        pred, pred_c, base_c, fg_c, idx, ctx.tape, ctx.out_node = out
        ctx.model, ctx.inputs = model, inputs
        ctx.mark_non_differentiable(pred, base_c, fg_c)
        ctx.set_materialize_grads(False)
        if idx is None:
            idx = torch.empty(0, dtype=torch.int32, device=pred.device)
        ctx.mark_non_differentiable(idx)
        return pred_c, pred, base_c, fg_c, idx

    @staticmethod
    def backward(ctx, d_pred_c, *unused):
        m = ctx.model
        base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights = ctx.inputs
        n, hc, wc, _ = warp.shape
        m.flat_grads.zero_()
        if d_pred_c is not None:
            d_pred_c = d_pred_c.contiguous()
            if (hc, wc) != (m.imh, m.imw):
                d_pred_c = C.resize_bilinear_backward(d_pred_c, hc, wc)
            dpred = torch.empty((n, m.uvh, m.uvw, 3), device=base.device, dtype=torch.float32)
            C.warp_backward(d_pred_c, warp, n, m.uvh, m.uvw, hc, wc, dpred)
            ctx.tape.backward(ctx.out_node, dpred)          # (+ base and the corner zero are constants / masks of pred's warp)
        return m.flat_grads, None, None, None


class _RenderFn(torch.autograd.Function):
    """torch-autograd glue around the hand-written forward/backward plans: `flat_params` is the
    model's single parameter bucket; its gradient is the flat gradient bucket the backward plan
    fills (one tensor -> one RCCL all-reduce, one fused Adam launch)."""

    @staticmethod
    def forward(ctx, flat_params, model, inputs, want_indices):
        base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights = inputs
        out = model._render(base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights, None, want_indices, inference=False)
        ctx.model, ctx.inputs, ctx.generation = model, inputs, model.plan.generation
        pred, pred_c, base_c, fg_c, idx = out
        nd = [t for t in (pred, base_c, fg_c) if t is not None]
        ctx.mark_non_differentiable(*nd)
        ctx.set_materialize_grads(False)
        if idx is None:
            idx = torch.empty(0, dtype=torch.int32, device=pred.device)
        ctx.mark_non_differentiable(idx)
        return pred_c, pred, base_c, fg_c, idx

    @staticmethod
    def backward(ctx, d_pred_c, *unused):
        ctx.model._render_backward(d_pred_c, ctx.inputs, ctx.generation)
        return ctx.model.flat_grads, None, None, None


class Model(BaseModel):
    def __init__(self, config):
        # needed by the Barron loss
        self.imh = config.getint('DEFAULT', 'imh')
        self.imw = config.getint('DEFAULT', 'imw')
        super().__init__(config)
        g = lambda k: config.get('DEFAULT', k)
        net_args = (config.getint('DEFAULT', 'depth0'), config.getint('DEFAULT', 'depth'),
                    config.getint('DEFAULT', 'kernel'), config.getint('DEFAULT', 'stride'))
        net_kwargs = {'norm_type': g('norm'), 'act_type': g('act'), 'pool_type': g('pool')}
        self.net = {'query': convnet.Network(*net_args, **net_kwargs),
                    'obs': convnet.Network(*net_args, **net_kwargs)}
        # the observation network keeps only its contracting (encoder) layers
        self.net['obs'].layers = [x for i, x in enumerate(self.net['obs'].layers)
                                  if self.net['obs'].is_contracting[i]]
        self.uvh = config.getint('DEFAULT', 'uvh')
        self.uvw = config.getint('DEFAULT', 'uvw')
        self.use_obs = config.getboolean('DEFAULT', 'use_obs')
        self.skip_connect_base = config.getboolean('DEFAULT', 'skip_connect_base')
        # config branches the fused plan does not execute (act = elu, a norm, pooling + upconv) run layer by layer (generic.py)
        self.generic = not all(l.is_plain() for net in self.net.values() for l in net.layers if hasattr(l, 'is_plain'))
        self.psnr = metric.PSNR(np.float32)                      # nlt/models/nlt.py:64
        self.plan = RenderPlan(self.net['query'], self.net['obs'], self.use_obs)
        # `precision = bf16` (not a reference key; BASELINE config 5): the middle of the network on bf16 MFMA / bf16 storage
        self.plan.precision = config.get('DEFAULT', 'precision', fallback=self.plan.precision)
        if self.plan.precision not in ('fp32', 'bf16', 'f32x3', 'f32x3_9'):
            raise NotImplementedError("precision = %s" % self.plan.precision)
        self.conv_algo = C.ALGO_AUTO
        # hipGraph replay of the inference forward (opt-in: NLT_GRAPH=1 or model.use_graphs = True).  The ~36 launches
        # of a step cost ~0.6 ms of host time; for small workloads (512^2, k = 1) that is the whole step.
        self.use_graphs = os.environ.get('NLT_GRAPH', '0') == '1'
        self._graph = None              # {'key', 'hits', 'graph', 'out'}: the entry of `_graphs` used last
        self._graphs = {}               # input addresses (+ weights version ...) -> entry; a few staging slots' worth

    def _init_loss(self):
        wloss = []
        for x in self.config.get('DEFAULT', 'loss').split(','):
            loss_name, weight = self._parse_loss_and_weight(x)
            if loss_name == 'l2':
                loss = losses.L2()
            elif loss_name == 'barron':
                loss = losses.Barron(self.imw, self.imh)
            elif loss_name in ('lpips', 'l1', 'ssim'):
                # lpips: the frozen AlexNet blob is not part of the reference tree
                # (.MISSING_LARGE_BLOBS); l1/ssim have no keep_batch and crash the
                # reference's own train step
                raise NotImplementedError(loss_name)
            else:
                raise NotImplementedError(loss_name)
            wloss.append((weight, loss))
        return wloss

    def build(self, device='cuda'):
        """Creates all variables (Keras builds lazily on first call; this does it eagerly)."""
        mult = 2 if self.use_obs else 1
        q, o = self.net['query'], self.net['obs']
        cin_q, cin_o, stack = 5, 3, []
        for i, (layer, c) in enumerate(zip(q.layers, q.is_contracting)):
            if c:
                n_out = layer.build(cin_q, device)
                o.layers[i].build(cin_o, device)
                cin_o = n_out
                cin_q = mult * n_out
                stack.append(cin_q)
            else:
                if stack:
                    cin_q += stack.pop()
                cin_q = layer.build(cin_q, device)
        self._flatten(device)
        return self

    def _conv_layers(self):
        """Every Conv2D of both nets in the flat-bucket order: query layers, then obs layers."""
        out = []
        for name in ('query', 'obs'):
            for layer in self.net[name].layers:
                out += _convs_of(layer)
        return out

    def _flatten(self, device):
        """Moves every kernel / bias into ONE flat fp32 parameter bucket (16-byte aligned slots)
        with a matching flat gradient bucket; layers keep views.  One bucket = one RCCL all-reduce
        and one fused Adam launch per step."""
        convs = self._conv_layers()
        # Slot ORDER inside the bucket = the order in which the backward pass FINISHES the weight gradients (it walks head ->
        # expanding blocks -> encoder, deepest level first), cut into three fixed contiguous ranges (`bucket_ranges`):
        #   0  the query net's expanding blocks                                   (3.3 MB at depth 256)
        #   1  encoder levels D .. M of both paths, deepest first; M = min(3, D)  (10.0 MB: the 64..256-channel levels)
        #   2  everything else: levels M-1 .. 0 and the 1x1 head                  (0.1 MB; the fused training path adds the
        #      head's skip rows and level 1's stride-2 gradients at the very end)
        # Range 0 is all-reduced while the encoder's backward runs, range 1 while the wide shallow levels and the fused front
        # backward run (trainvali.distributed_train_step); only range 2 is exposed behind the backward.
        q, o = self.net['query'], self.net['obs']
        late, mid = [], []
        for layer, is_c in zip(q.layers, q.is_contracting):
            if not is_c and not hasattr(layer, 'set_weights'):
                late += _convs_of(layer)
        D = sum(1 for layer, is_c in zip(q.layers, q.is_contracting) if is_c and not hasattr(layer, 'set_weights'))
        M = min(3, D)
        self.plan.grad_mid_level = M if (D >= 2 and os.environ.get('NLT_GRAD_RANGES', '3') != '2') else 0   # (2: the r01-r04 split, A/B)
        if self.plan.grad_mid_level:
            for l in range(D, M - 1, -1):
                mid += _convs_of(q.layers[l]) + (_convs_of(o.layers[l]) if l < len(o.layers) else [])
        taken = {id(c) for c in late + mid}
        where, off = {}, 0
        ends = []
        for group in (late, mid, [c for c in convs if id(c) not in taken]):
            for c in group:
                for name in ('kernel', 'bias'):
                    t = getattr(c, name)
                    where[(id(c), name)] = (off, t.numel(), tuple(t.shape))
                    off += (t.numel() + 3) // 4 * 4
            ends.append(off)
        self.bucket_ranges = [0] + ends     # range i = [bucket_ranges[i], bucket_ranges[i + 1]) floats; an absent group: empty
        self.bucket_split = ends[0]         # (the leading range alone: what rounds 1-4 overlapped)
        slots = [where[(id(c), name)] for c in convs for name in ('kernel', 'bias')]
        flat = torch.zeros(off, device=device, dtype=torch.float32)
        it = iter(slots)
        for c in convs:
            for name in ('kernel', 'bias'):
                o, n, shp = next(it)
                flat[o:o + n].copy_(getattr(c, name).detach().reshape(-1))
        self.flat_params = flat.requires_grad_(True)
        self.flat_grads = torch.zeros_like(flat)
        self._slots = slots                 # (offset, count, shape) of every variable, in CANONICAL order: `_conv_layers()` x (kernel, bias)
        self.n_params = sum(n for _, n, _ in slots)
        self._epoch = [0]
        from ..networks.elements import PackRegistry
        self.pack_registry = PackRegistry(lambda: (self.flat_params._version, self._epoch[0]))
        it = iter(slots)
        for c in convs:
            for name in ('kernel', 'bias'):
                o, n, shp = next(it)
                setattr(c, name, self.flat_params.detach()[o:o + n].view(shp))
                setattr(c, 'd' + name, self.flat_grads[o:o + n].view(shp))
            c._epoch = self._epoch
            c._packed = {}
            c._registry = self.pack_registry

    def mark_weights_updated(self):
        """Call after writing the flat bucket through a raw pointer (optimizer kernel): packed
        MFMA fragments are re-derived on next use."""
        self._epoch[0] += 1

    def load_weights(self, weights):
        """weights = {'query': [[(kernel, bias), ...] per layer], 'obs': [...]} in Keras layouts
        (NumPy or torch); the exchange format with the oracle and with TF checkpoints."""
        for name in ('query', 'obs'):
            layers = self.net[name].layers
            assert len(layers) == len(weights[name]), (name, len(layers), len(weights[name]))
            for layer, lw in zip(layers, weights[name]):
                convs = _convs_of(layer)
                assert len(convs) == len(lw)
                for ci, (c, (k, b)) in enumerate(zip(convs, lw)):
                    if c.built and (tuple(k.shape) != tuple(c.kernel.shape) or tuple(b.shape) != tuple(c.bias.shape)):
                        raise ValueError("net_%s: layer %d conv %d expects kernel %s / bias %s, got %s / %s"
                                         % (name, layers.index(layer), ci, tuple(c.kernel.shape), tuple(c.bias.shape),
                                            tuple(k.shape), tuple(b.shape)))
                    c.set_weights(k, b)
        return self

    def get_weights(self):
        """The inverse of `load_weights`: {'query': [[(kernel, bias), ...] per layer], 'obs': [...]} as NumPy arrays in
        Keras layouts (what a TF checkpoint of the reference holds under net/net_<name>_layer<i>)."""
        out = {}
        for name in ('query', 'obs'):
            out[name] = []
            for layer in self.net[name].layers:
                out[name].append([(c.kernel.detach().cpu().numpy().copy(), c.bias.detach().cpu().numpy().copy())
                                  for c in _convs_of(layer)])
        return out

    # -- the flat bucket's slot ORDER is a tuning choice (`_flatten`: backward-completion order, NLT_GRAD_RANGES): anything that
    # leaves the process -- checkpoints, optimizer slots -- is exchanged per VARIABLE in canonical order and never as a raw bucket
    def bucket_to_variables(self, flat):
        """A bucket-shaped vector (parameters, Adam m / v / vhat) -> list of tensors, one per variable, canonical order."""
        flat = flat.detach().reshape(-1)
        assert flat.numel() == self.flat_params.numel()
        return [flat[o:o + n].reshape(shp).clone() for o, n, shp in self._slots]

    def variables_to_bucket(self, variables, out):
        """The inverse, into the bucket-shaped tensor `out` (padding floats between slots stay as they are)."""
        if len(variables) != len(self._slots):
            raise ValueError("checkpoint was written by a different architecture (%d variables, this one has %d)" % (len(variables), len(self._slots)))
        for (o, n, shp), v in zip(self._slots, variables):
            if tuple(v.shape) != tuple(shp):
                raise ValueError("checkpoint was written by a different architecture (variable of shape %s where %s is expected)"
                                 % (tuple(v.shape), tuple(shp)))
        with torch.no_grad():
            flat = out.detach().reshape(-1)
            for (o, n, shp), v in zip(self._slots, variables):
                flat[o:o + n].copy_(v.reshape(-1))
        return out

    def state_dict(self):
        """Everything `tf.train.Checkpoint(net=...)` tracks for this model (nlt/trainvali.py:134-141): every kernel / bias as
        its own tensor, in canonical order (query layers, then obs layers; kernel then bias) -- independent of how this
        process happens to lay its flat bucket out."""
        assert getattr(self, 'flat_params', None) is not None, "build() the model first"
        return {'variables': self.bucket_to_variables(self.flat_params)}

    def load_state_dict(self, sd):
        if 'variables' not in sd:
            raise ValueError("not a per-variable state dict (format nlt_amd-ckpt-2)")
        self.variables_to_bucket(list(sd['variables']), self.flat_params)
        self.mark_weights_updated()
        return self

    def legacy_bucket_layout(self):
        """How rounds 1-5 (format nlt_amd-ckpt-1) laid the RAW flat bucket out in a checkpoint: the slot table of the
        CURRENT process.  A ckpt-1 file does not say which layout wrote it; `trainvali.restore_checkpoint` refuses it unless
        the caller vouches for the layout (`legacy_layout=True`: same NLT_GRAD_RANGES, same code generation)."""
        return list(self._slots)

    # ---------------------------------------------------------------- forward
    def _render(self, base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights, obs_override, want_indices, inference=True,
                resident=None):
        if resident is not None:
            n, hc, wc, dev = resident.n, resident.hc, resident.wc, resident.cvis.device
        else:
            (n, hc, wc, _), dev = warp.shape, base.device
        # the plan's last launch writes the rendered texels straight into a tensor of this call when it runs the fused ends (no
        # copy of the plan's reusable buffer afterwards); `_pred_fresh` tells the caller whether that happened
        h_, w_ = (resident.h, resident.w) if resident is not None else base.shape[1:3]
        timing_all = self.plan.timer is not None and getattr(self.plan.timer, 'only', None) is None     # (per-launch survey: plain path)
        fresh = (torch.empty((n, h_, w_, 3), device=dev, dtype=torch.float32)
                 if (dev.type == 'cuda' and not timing_all and os.environ.get('NLT_PRED_COPY', '0') == '0')
                 else None)                                      # (train forwards too since r05: their last launch takes it as well)
        E = lambda: torch.empty((n, hc, wc, 3), device=dev, dtype=torch.float32)
        pred_camspc, base_camspc, fg_camspc = E(), E(), E()
        idx = torch.empty((n, hc, wc, 4), device=dev, dtype=torch.int32) if want_indices else None

        def resample(pred_, pc, bc, fc, ix):
            if resident is not None:
                # base and the uv2cam map are gathered where they live (uint8 diffuse store, fp16 map store): neither the
                # float32 base nor a float32 copy of the map is ever written
                C.warp_forward_store(pred_, resident.diffuse, resident.uv2cam, resident.ids, n, self.uvh, self.uvw, hc, wc, pc, bc, fc, ix)
            else:
                C.warp_forward(pred_, base, warp, n, self.uvh, self.uvw, hc, wc, pc, bc, fc, ix)
        # (The base / foreground gathers do not depend on the network; queueing them on a side stream under the expanding blocks
        # was built and measured in r04 -- 1.294 vs 1.285 ms, nothing -- and removed in r06.)
        pred, _ = self.plan.forward(base, cvis, lvis, nn_rgb, nn_base, obs_weights=obs_weights,
                                    obs_override=obs_override, skip_connect_base=self.skip_connect_base,
                                    algo=self.conv_algo, inference=inference, resident=resident, pred_out=fresh)
        self._pred_fresh = fresh is not None and pred is fresh
        resample(pred, pred_camspc, base_camspc, fg_camspc, idx)
        if (hc, wc) != (self.imh, self.imw):
            fg_camspc = C.resize_bilinear_forward(fg_camspc, self.imh, self.imw)
            base_camspc = C.resize_bilinear_forward(base_camspc, self.imh, self.imw)
            pred_camspc = C.resize_bilinear_forward(pred_camspc, self.imh, self.imw)
        return pred, pred_camspc, base_camspc, fg_camspc, idx

    def _render_generic(self, base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights, obs_override, want_indices, record=False):
        """`_render` on the layer-by-layer path: Model._call on generic layers, then + base, corner zero, warp, resize."""
        from .. import generic
        n, hc, wc, _ = warp.shape
        x = torch.cat((base, cvis, lvis), 3)                                    # nlt.py:95
        k = nn_rgb.shape[1]
        y_obs = [C.sub_forward(nn_rgb[:, i].contiguous(), nn_base[:, i].contiguous()) for i in range(k)]   # nlt.py:96
        node, tape = generic.forward(self, x, y_obs, obs_weights=obs_weights, obs_override=obs_override, record=record)
        pred = torch.empty_like(base)
        C.finish_pred(node.value, base if self.skip_connect_base else None, pred)   # + base, texel (0,0) zeroed (nlt.py:99-110)
        E = lambda: torch.empty((n, hc, wc, 3), device=base.device, dtype=torch.float32)
        pred_camspc, base_camspc, fg_camspc = E(), E(), E()
        idx = torch.empty((n, hc, wc, 4), device=base.device, dtype=torch.int32) if want_indices else None
        C.warp_forward(pred, base, warp, n, self.uvh, self.uvw, hc, wc, pred_camspc, base_camspc, fg_camspc, idx)
        if (hc, wc) != (self.imh, self.imw):
            fg_camspc = C.resize_bilinear_forward(fg_camspc, self.imh, self.imw)
            base_camspc = C.resize_bilinear_forward(base_camspc, self.imh, self.imw)
            pred_camspc = C.resize_bilinear_forward(pred_camspc, self.imh, self.imw)
        return (pred, pred_camspc, base_camspc, fg_camspc, idx, tape, node) if record else (pred, pred_camspc, base_camspc, fg_camspc, idx)

    def _render_backward(self, d_pred_c, inputs, generation=None):
        """Fills the flat gradient bucket from dL/d(pred_camspc): resize / warp (TFA resampler) adjoints, then the
        hand-ordered backward plan over the activations the last inference=False forward left behind."""
        base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights = inputs
        n, hc, wc, _ = warp.shape
        self.flat_grads.zero_()
        if d_pred_c is not None:
            d_pred_c = d_pred_c.contiguous()
            if (hc, wc) != (self.imh, self.imw):
                d_pred_c = C.resize_bilinear_backward(d_pred_c, hc, wc)
            dpred = getattr(self, '_dpred', None)            # persistent: the backward plan's launch tape points at it
            if dpred is None or dpred.shape[0] != n or dpred.device != base.device:
                dpred = self._dpred = torch.empty((n, self.uvh, self.uvw, 3), device=base.device, dtype=torch.float32)
            C.warp_backward(d_pred_c, warp, n, self.uvh, self.uvw, hc, wc, dpred)
            self.plan.backward(dpred, base, cvis, lvis, nn_rgb, nn_base, obs_weights, generation=generation)

    def train_forward_backward(self, batch, global_bs):
        """One train step's forward + loss + backward WITHOUT the torch.autograd tape around the network (only the loss
        is differentiated, with autograd.grad -- no AccumulateGrad node, so the whole thing can be captured in a
        hipGraph): returns (loss summed over this rank's examples / global_bs, to_vis) and leaves the gradients in
        `flat_grads`.  Same arithmetic as `call(batch, 'train')` + `compute_loss` + `.backward()`."""
        id_, base, cvis, lvis, warp, rgb, rgb_camspc, nn_id, nn_base, nn_rgb, nn_rgb_camspc = batch
        if isinstance(base, ResidentTexels):                     # load_batch(resident=True): training reads the float buffers
            m = base.materialize()
            base, cvis, lvis, rgb, nn_base, nn_rgb = (m[x] for x in ('base', 'cvis', 'lvis', 'rgb', 'nn_base', 'nn_rgb'))
            warp = m['warp'] if warp is None else warp
        if nn_rgb.dim() == 4:
            nn_rgb, nn_base = nn_rgb.unsqueeze(1), nn_base.unsqueeze(1)
        nn_rgb, nn_base = nn_rgb.contiguous(), nn_base.contiguous()
        with torch.no_grad():
            pred, pred_camspc, base_camspc, fg_camspc, _ = self._render(base, cvis, lvis, warp, nn_rgb, nn_base, None, None,
                                                                        False, inference=False)
            # (the fused train forward wrote `pred` into a tensor of this call: no 50 MB copy of the plan's buffer at the step's end)
            pred_vis = pred if self._pred_fresh else pred.clone()
            gen = self.plan.generation
            plain_l2 = len(self.wloss) == 1 and self.wloss[0][0] == 1 and type(self.wloss[0][1]).__name__ == 'L2'
            if plain_l2:
                # gt = rgb * fg, the per-example means, their sum / global batch and the gradient in one launch (12 otherwise)
                loss, gt_camspc, d_pred_c = C.l2_train_loss(pred_camspc, rgb_camspc, fg_camspc, global_bs)
            else:
                gt_camspc = C.mul_forward(rgb_camspc, fg_camspc)
            plain_barron = (not plain_l2 and len(self.wloss) == 1 and self.wloss[0][0] == 1
                            and type(self.wloss[0][1]).__name__ == 'Barron')
            if plain_barron:
                # the released loss (`loss = barron`): value and gradient from the one C call that computes both, the
                # sum / global batch applied to the per-example gradient rows -- no autograd graph around it (host time:
                # the released 512^2 training shape is enqueue-bound)
                per, dunit = C.barron_loss(pred_camspc.contiguous(), gt_camspc, True)
                loss = per.sum() / global_bs
                key = (per.shape[0], float(global_bs), str(per.device))
                inv = getattr(self, '_inv_gbs', None)
                if inv is None or inv[0] != key:
                    inv = self._inv_gbs = (key, torch.full((per.shape[0],), 1.0 / global_bs, device=per.device))
                d_pred_c = C.scale_rows(dunit, inv[1])
        if not plain_l2 and not plain_barron:
            leaf = pred_camspc.detach().requires_grad_(True)
            with torch.enable_grad():
                loss = self.compute_loss(leaf, gt_camspc, keep_batch=True).sum() / global_bs
                (d_pred_c,) = torch.autograd.grad(loss, leaf)
        with torch.no_grad():
            self._render_backward(d_pred_c, (base, cvis, lvis, warp, nn_rgb, nn_base, None), gen)
            to_vis = {'id': id_, 'nn_id': nn_id, 'base_camspc': base_camspc, 'pred': pred_vis, 'pred_camspc': pred_camspc,
                      'nn_camspc': nn_rgb_camspc, 'gt': rgb, 'gt_camspc': gt_camspc}
        return loss.detach(), to_vis

    def _render_maybe_graphed(self, base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights, obs_override, want_indices):
        """`_render` (+ the copy of pred) either launched kernel by kernel or, with use_graphs, replayed as one
        hipGraph.  A graph is tied to the ADDRESSES of its inputs: it is captured the second time the same input
        tensors (and weights version) come back and replayed from then on; up to 8 such graphs are kept (the slots of a
        staging ring).  Replayed outputs are the graph's own static tensors: consume them before the next call that
        replays the same graph."""
        args = (base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights, obs_override, want_indices)
        eager = lambda: self._render(*args) + (None,)
        ok = (self.use_graphs and base.is_cuda and self.plan.timer is None and obs_override is None and obs_weights is None
              and getattr(self, 'flat_params', None) is not None)
        if not ok:
            out = eager()
            return out[:5] + (out[0] if getattr(self, '_pred_fresh', False) else out[0].clone(),)
        key = (tuple(t.data_ptr() for t in (base, cvis, lvis, warp, nn_rgb, nn_base)), tuple(base.shape), tuple(warp.shape),
               nn_rgb.shape[1], want_indices, self._epoch[0], self.flat_params._version, self.plan.fuse_ends, self.conv_algo)
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 8:                           # ever-changing input addresses: forget the oldest
                self._graphs.pop(next(iter(self._graphs)))
            self._graph = self._graphs[key] = {'key': key, 'hits': 0, 'graph': None, 'out': None}
            out = eager()                                        # first sight: eager (also runs the plan-time autotune)
            return out[:5] + (out[0] if getattr(self, '_pred_fresh', False) else out[0].clone(),)
        self._graph = g
        if g['graph'] is None:                                   # second sight: capture (the capture itself does not execute)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                o = self._render(*args)
                g['out'] = o + (o[0].clone(),)
            g['graph'] = graph
        g['graph'].replay()
        g['hits'] += 1
        return g['out']

    def call(self, batch, mode, obs_override=None, obs_weights=None, want_indices=False):
        self._validate_mode(mode)
        id_, base, cvis, lvis, warp, rgb, rgb_camspc, nn_id, nn_base, nn_rgb, nn_rgb_camspc = batch
        differentiable = (mode == 'train' and torch.is_grad_enabled() and obs_override is None
                          and getattr(self, 'flat_params', None) is not None)
        resident = None
        if isinstance(base, ResidentTexels):
            # texel buffers still in the uint8 store (Dataset.load_batch(resident=True)): the fused front kernel reads
            # them there when this call is an inference forward it can take; anything else gets the float tensors
            act0 = self.net['query'].layers[1].convs()[0][1] if (not self.generic and len(self.net['query'].layers) > 2) else None
            if (not self.generic and not differentiable and obs_weights is None and act0 is not None
                    and getattr(self, 'flat_params', None) is not None
                    and (self.plan.resident_ok(base.n, base.k, base.h, base.w, act0.alpha) if obs_override is None
                         else self.plan.resident_override_ok(base, obs_override, act0.alpha))):
                resident = base
                base = cvis = lvis = nn_rgb = nn_base = None
            else:
                m = base.materialize()
                base, cvis, lvis, rgb, nn_base, nn_rgb = (m[x] for x in ('base', 'cvis', 'lvis', 'rgb', 'nn_base', 'nn_rgb'))
                warp = m['warp'] if warp is None else warp
        if resident is None:
            if nn_rgb.dim() == 4:           # the reference's single neighbour
                nn_rgb, nn_base = nn_rgb.unsqueeze(1), nn_base.unsqueeze(1)
            nn_rgb, nn_base = nn_rgb.contiguous(), nn_base.contiguous()
        if self.generic:
            if differentiable:
                pred_camspc, pred, base_camspc, fg_camspc, idx = _GenericFn.apply(
                    self.flat_params, self, (base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights), want_indices)
                pred_copy = pred
            else:
                pred, pred_camspc, base_camspc, fg_camspc, idx = self._render_generic(
                    base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights, obs_override, want_indices)
                pred_copy = pred
            differentiable = False                                   # (`pred` is a fresh tensor on this path: no clone below)
        elif resident is not None:
            pred, pred_camspc, base_camspc, fg_camspc, idx = self._render(None, None, None, warp, None, None, None, obs_override,
                                                                          want_indices, resident=resident)
            pred_copy = pred if getattr(self, '_pred_fresh', False) else pred.clone()
            if mode != 'test' and rgb is None:
                rgb = resident.materialize()['rgb']
        elif differentiable:
            pred_camspc, pred, base_camspc, fg_camspc, idx = _RenderFn.apply(
                self.flat_params, self, (base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights), want_indices)
        else:
            pred, pred_camspc, base_camspc, fg_camspc, idx, pred_copy = self._render_maybe_graphed(
                base, cvis, lvis, warp, nn_rgb, nn_base, obs_weights, obs_override, want_indices)
        # `pred` lives in the plan's reusable buffer: hand out a copy
        to_vis = {'id': id_, 'nn_id': nn_id, 'base_camspc': base_camspc,
                  'pred': pred_copy if not differentiable else pred.clone(),
                  'pred_camspc': pred_camspc, 'nn_camspc': nn_rgb_camspc}
        if want_indices:
            to_vis['uv_indices'] = idx
        if mode in ('train', 'vali'):
            gt_camspc = C.mul_forward(rgb_camspc, fg_camspc)     # imgutil.alpha_blend(rgb_camspc, fg_camspc)
            to_vis['gt'] = rgb
            to_vis['gt_camspc'] = gt_camspc
            return pred_camspc, gt_camspc, {}, to_vis
        return pred_camspc, None, None, to_vis

    def _call(self, query_x, obs_xs, obs_weights=None, obs_override=None):
        """Layer-by-layer form with the reference's exact structure (materialised concats),
        built on the generic layer objects; `call` uses the fused RenderPlan instead."""
        q, o = self.net['query'], self.net['obs']
        stack, query_y = [], None
        for i, (layer_q, is_c) in enumerate(zip(q.layers, q.is_contracting)):
            if is_c:
                obs_ys = [o.layers[i](x) for x in obs_xs]
                obs_xs = obs_ys
                query_y = layer_q(query_x)
                if self.use_obs:
                    if obs_override is not None:
                        obs_agg = obs_override[i].expand(query_y.shape[0], -1, -1, -1)
                    else:
                        n, h, w, c = obs_ys[0].shape
                        stacked = torch.stack(obs_ys, 1).contiguous()
                        obs_agg = torch.empty_like(obs_ys[0])
                        C.obs_mean_forward(stacked, obs_weights, n, len(obs_ys), h * w, c, obs_agg, c)
                    query_x = torch.cat((query_y, obs_agg), -1)
                else:
                    query_x = query_y
                stack.append(query_x)
            else:
                if stack:
                    query_x = torch.cat((query_x, stack.pop()), -1)
                query_y = layer_q(query_x)
                query_x = query_y
        return query_y

    def compute_loss(self, pred, gt, **kwargs):
        loss = 0
        for weight, loss_func in self.wloss:
            loss = loss + weight * loss_func(gt, pred, **kwargs)
        return loss

    def vis_batch(self, data_dict, outdir, mode, dump_raw_to=None,
                  text_loc_ratio=0.05, text_size_ratio=0.05, text_color=(1, 1, 1)):
        """nlt/models/nlt.py:207-272: per sample `<i>_{base,pred,nn[,gt]}.png` (clip to [0,1], linear -> sRGB when
        `linear_space`, x255 truncated), the two labelled animated PNGs, `<i>_metadata.json` (ids; PSNR of prediction and
        diffuse base against the ground truth outside test mode -- on the CLIPPED LINEAR values, as the reference), and
        optionally the raw dict as a pickle (arrays as NumPy: there is no tf.Tensor to pickle here).
        One device -> host copy per map; the arithmetic that decides the bytes is float32 NumPy like the reference's;
        the PSNR sums run on the device when the maps live there (`metric.PSNR`, float64)."""
        is_linear = self.config.getboolean('DEFAULT', 'linear_space')
        self._validate_mode(mode)
        if data_dict.get('id') is None or data_dict.get('nn_id') is None:
            raise ValueError("vis_batch needs the samples' ids (`id`, `nn_id` of the batch tuple): this batch carries none")
        ids = [V.to_str(x) for x in data_dict['id']]
        nn_ids = [V.to_str(x) for x in data_dict['nn_id']]
        keys = ('base_camspc', 'pred_camspc', 'nn_camspc') + (() if mode == 'test' else ('gt_camspc',))
        host = {k: np.clip(V.to_numpy(data_dict[k]), 0, 1) for k in keys}
        bases, preds, nns, gts = host['base_camspc'], host['pred_camspc'], host['nn_camspc'], host.get('gt_camspc')
        # the clipped maps where the PSNR kernel can read them without a second trip over PCIe
        dev = {k: (data_dict[k].detach().clamp(0, 1) if torch.is_tensor(data_dict[k]) and data_dict[k].is_cuda else None) for k in keys}
        for i in range(len(ids)):
            base, pred, nn = bases[i], preds[i], nns[i]
            gt = None if gts is None else gts[i]
            if is_linear:
                base, pred, nn = V.linear2srgb(base), V.linear2srgb(pred), V.linear2srgb(nn)
                gt = None if gt is None else V.linear2srgb(gt)
            imgs = {'base': V.write_arr(base, join(outdir, '%d_base.png' % i)),
                    'pred': V.write_arr(pred, join(outdir, '%d_pred.png' % i))}
            V.write_arr(nn, join(outdir, '%d_nn.png' % i))
            imgs['gt'] = None if gt is None else V.write_arr(gt, join(outdir, '%d_gt.png' % i))
            hw = base.shape[:2]
            label_loc = (int(text_loc_ratio * hw[1]), int(text_loc_ratio * hw[0]))
            font_size = int(text_size_ratio * hw[0])
            V.make_apng((imgs['base'], imgs['pred']), labels=('Diffuse Base', 'Prediction'), label_top_left_xy=label_loc,
                        font_size=font_size, font_color=text_color, outpath=join(outdir, '%d_base-vs-pred.apng' % i))
            if imgs['gt'] is not None:
                V.make_apng((imgs['gt'], imgs['pred']), labels=('Ground Truth', 'Prediction'), label_top_left_xy=label_loc,
                            font_size=font_size, font_color=text_color, outpath=join(outdir, '%d_gt-vs-pred.apng' % i))
        for i, id_ in enumerate(ids):
            metadata = {'id': id_, 'nn_id': nn_ids[i]}
            if gts is not None:
                gt, pred, base = (dev[k][i] if dev[k] is not None else host[k][i] for k in ('gt_camspc', 'pred_camspc', 'base_camspc'))
                metadata['pred_psnr'] = self.psnr(gt, pred)
                metadata['base_psnr'] = self.psnr(gt, base)
            V.write_json(metadata, join(outdir, '%d_metadata.json' % i))
        if dump_raw_to is not None:
            raw = {k: (V.to_numpy(v) if hasattr(v, 'detach') else v) for k, v in data_dict.items()}
            os.makedirs(dirname(dump_raw_to) or '.', exist_ok=True)
            with open(dump_raw_to, 'wb') as h:
                pickle.dump(raw, h)

    def compile_batch_vis(self, batch_vis_dirs, outpref, mode, fps=6, file_explorer=''):
        """nlt/models/nlt.py:274-285: train / vali -> `<outpref>.html` (one table row per sample), test -> the frames in id
        order as `<outpref>.mp4` where an encoder exists and `<outpref>.apng` + `.frames.json` always.  Returns the link the
        reference logs to TensorBoard (`file_explorer` + path; the reference hard-codes its lab's file server there)."""
        self._validate_mode(mode)
        if mode in ('train', 'vali'):
            outpath = outpref + '.html'
            self._compile_into_webpage(batch_vis_dirs, outpath, title="NLT (%s)" % mode)
        else:
            outpath = outpref + '.mp4'
            written = self._compile_into_video(batch_vis_dirs, outpath, fps=fps)
            if outpath not in written:          # no matplotlib / ffmpeg here (advisor r05): link the file that exists
                outpath = outpref + '.apng'
        return file_explorer + outpath

    @staticmethod
    def _compile_into_webpage(batch_dirs, out_html, title=None):
        page = V.Page()
        if title is not None:
            page.add_header(title)
        n_rows = 0
        for batch_dir in batch_dirs:
            for metadata_path in sorted(glob(join(batch_dir, '?_metadata.json'))):
                pref = metadata_path[:-len('metadata.json')]
                page.add_row([str(V.read_json(metadata_path)), pref + 'base-vs-pred.apng', pref + 'gt-vs-pred.apng', pref + 'nn.png'],
                             ['text', 'image', 'image', 'image'],
                             captions=["Metadata", "Prediction vs. Diffuse Base", "Prediction vs. Ground Truth", "Nearest Neighbor"])
                n_rows += 1
        assert n_rows > 0, "No row"
        page.save(out_html)

    @staticmethod
    def _compile_into_video(batch_dirs, out_mp4, fps=12):
        from ..datasets.nlt import read_png
        frames, paths = {}, {}
        for batch_dir in batch_dirs:
            for metadata_path in glob(join(batch_dir, '?_metadata.json')):
                pred_path = metadata_path[:-len('metadata.json')] + 'pred.png'
                if not exists(pred_path):
                    warnings.warn("Skipping because of missing file:\n\t%s" % pred_path)
                    continue
                id_ = V.read_json(metadata_path)['id']
                frames[id_], paths[id_] = read_png(pred_path), pred_path
        order = sorted(frames)
        written = V.write_frames([frames[k] for k in order], out_mp4, fps=fps)
        stem = out_mp4[:-len('.mp4')]
        V.write_json({'fps': fps, 'ids': order, 'frames': [paths[k] for k in order], 'written': written}, stem + '.frames.json')
        return written
