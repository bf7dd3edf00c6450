"""NLT model on libnlt_hip.so -- same surface as reference nlt/models/nlt.py:38-205
(`Model(config)`, `.net['query'|'obs']`, `call(batch, mode, obs_override)`, `_call`,
`compute_loss`); tensors are torch CUDA tensors instead of TF eager tensors.
"""
import torch

from .. import _capi as C
from .. import losses
from ..engine import RenderPlan
from ..networks import convnet
from .base import Model as BaseModel


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
        self.plan = RenderPlan(self.net['query'], self.net['obs'], self.use_obs)
        self.conv_algo = C.ALGO_AUTO

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
        return self

    def load_weights(self, weights):
        """weights = {'query': [[(kernel, bias), ...] per layer], 'obs': [...]} in Keras layouts
        (NumPy or torch); the exchange format with the oracle and with TF checkpoints."""
        for name in ('query', 'obs'):
            layers = self.net[name].layers
            assert len(layers) == len(weights[name]), (name, len(layers), len(weights[name]))
            for layer, lw in zip(layers, weights[name]):
                convs = [layer] if hasattr(layer, 'set_weights') else [c for c, _ in layer.convs()]
                assert len(convs) == len(lw)
                for c, (k, b) in zip(convs, lw):
                    c.set_weights(k, b)
        return self

    # ---------------------------------------------------------------- forward
    def call(self, batch, mode, obs_override=None, obs_weights=None, want_indices=False):
        self._validate_mode(mode)
        id_, base, cvis, lvis, warp, rgb, rgb_camspc, nn_id, nn_base, nn_rgb, nn_rgb_camspc = batch
        if nn_rgb.dim() == 4:           # the reference's single neighbour
            nn_rgb, nn_base = nn_rgb.unsqueeze(1), nn_base.unsqueeze(1)
        n, hc, wc, _ = warp.shape
        pred, _ = self.plan.forward(base, cvis, lvis, nn_rgb.contiguous(), nn_base.contiguous(),
                                    obs_weights=obs_weights, obs_override=obs_override,
                                    skip_connect_base=self.skip_connect_base, algo=self.conv_algo)
        E = lambda: torch.empty((n, hc, wc, 3), device=base.device, dtype=torch.float32)
        pred_camspc, base_camspc, fg_camspc = E(), E(), E()
        idx = torch.empty((n, hc, wc, 4), device=base.device, dtype=torch.int32) if want_indices else None
        C.warp_forward(pred, base, warp, n, self.uvh, self.uvw, hc, wc, pred_camspc, base_camspc, fg_camspc, idx)
        if (hc, wc) != (self.imh, self.imw):
            fg_camspc = C.resize_bilinear_forward(fg_camspc, self.imh, self.imw)
            base_camspc = C.resize_bilinear_forward(base_camspc, self.imh, self.imw)
            pred_camspc = C.resize_bilinear_forward(pred_camspc, self.imh, self.imw)
        to_vis = {'id': id_, 'nn_id': nn_id, 'base_camspc': base_camspc, 'pred': pred,
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
