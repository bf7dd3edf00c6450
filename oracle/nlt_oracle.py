"""CPU oracle of the NLT model: channel schedule, two-path U-Net, warp, losses, train step.

ORACLE = test infrastructure (see oracle/__init__.py); never imported by the product.
Follows, in order: nlt/util/net.py:18-56 -> nlt/networks/convnet.py:30-90 ->
nlt/networks/elements.py:26-39,69-78 -> nlt/models/nlt.py:39-64,89-199 ->
nlt/util/img.py:74-120,179-185 -> nlt/losses.py:39-53,90-118 -> nlt/trainvali.py:272-281.
"""
import math

import numpy as np
import torch

from . import tf_ops as T
from . import barron as B


# ----------------------------------------------------------------------------
# nlt/util/net.py:18-56
# ----------------------------------------------------------------------------
def gen_feat_n(min_n, max_n, final_n=3):
    assert max_n >= min_n and max_n >= final_n
    n_ch = [2 ** i for i in range(int(np.log2(min_n)) + 1, int(np.log2(max_n)) + 1)]
    if not n_ch or n_ch[0] != min_n:
        n_ch = [min_n] + n_ch
    if not n_ch or n_ch[-1] != max_n:
        n_ch.append(max_n)
    n_ch += n_ch[::-1]
    n_ch += [2 ** i for i in range(int(np.log2(n_ch[-1])) - 1, int(np.log2(final_n)), -1)]
    while n_ch and n_ch[-1] < final_n:
        n_ch.pop()
    n_ch.append(final_n)
    return n_ch


# ----------------------------------------------------------------------------
# nlt/networks/convnet.py:30-90 (released branch: norm=None, pool=None)
# ----------------------------------------------------------------------------
def build_layers(depth0, depth, kernel=2, stride=2):
    """Returns (layers, is_contracting, spatsize_changes); a layer is a dict
    {'kind': 'conv1x1'|'down'|'up', 'n': out_channels, 'act': bool}."""
    n_feat = gen_feat_n(depth0, depth)
    layers, is_c, ssc = [], [], []
    layers.append({'kind': 'conv1x1', 'n': n_feat[0], 'act': False})   # :44
    is_c.append(True); ssc.append(1)
    prev_n = 0
    for n in n_feat[:-1]:
        if n >= prev_n:                                                # :49-64
            layers.append({'kind': 'down', 'n': n, 'act': True, 'k': kernel, 's': stride})
            is_c.append(True); ssc.append(1 / stride)
        else:                                                          # :66-81
            layers.append({'kind': 'up', 'n': n, 'act': True, 'k': kernel, 's': stride})
            is_c.append(False); ssc.append(stride)
        prev_n = n
    layers.append({'kind': 'conv1x1', 'n': n_feat[-1], 'act': False})  # :85
    is_c.append(False); ssc.append(1)
    assert np.cumprod(ssc)[-1] == 1, "Resolution doesn't return to the original value"  # :88-90
    return layers, is_c, ssc


def _layer_in_channels(layers, is_c, cin_query, cin_obs, use_obs=True):
    """Channel count entering each query/obs layer under Model._call's concat
    rules (nlt/models/nlt.py:141-199), including the bottleneck self-concat."""
    q_in, o_in = [], []
    qc, oc = cin_query, cin_obs
    stack = []
    for L, c in zip(layers, is_c):
        if c:
            q_in.append(qc); o_in.append(oc)
            oc = L['n']
            qc = L['n'] + (L['n'] if use_obs else 0)
            stack.append(qc)
        else:
            if stack:
                qc = qc + stack.pop()
            q_in.append(qc); o_in.append(None)
            qc = L['n']
    return q_in, o_in


def init_weights(layers, in_channels, rng, bias_range=0.1, contracting_only=None, pool=False, norm=None):
    """Keras-layout weights.  conv: (kh,kw,Cin,Cout); deconv: (kh,kw,Cout,Cin).
    Kernels glorot-uniform (Keras default); biases U(-bias_range,bias_range) instead of
    Keras' zeros so the bias path is exercised.  norm = 'layer' / 'batch': a (gamma, beta) pair follows each
    conv of a down / up block (Keras variable order of the Sequential, convnet.py:50-59,67-76); gamma U(0.5, 1.5) and beta
    U(-bias_range, bias_range) instead of Keras' ones / zeros, for the same reason."""
    ws = []
    for i, (L, cin) in enumerate(zip(layers, in_channels)):
        if contracting_only is not None and not contracting_only[i]:
            break
        n = L['n']

        def bias():
            return rng.uniform(-bias_range, bias_range, size=(n,)).astype(np.float32)

        def with_norms(lw, first=0):
            if norm not in ('layer', 'batch'):
                return lw
            out = []
            for j, cw in enumerate(lw):
                out.append(cw)
                if j >= first:
                    out.append((rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32), bias()))
            return out
        if L['kind'] == 'conv1x1':
            ws.append([(T.glorot_uniform(rng, (1, 1, cin, n)), bias())])
        elif L['kind'] == 'down':
            k = L['k']
            ws.append(with_norms([(T.glorot_uniform(rng, (k, k, cin, n)), bias()),
                                  (T.glorot_uniform(rng, (k, k, n, n)), bias())]))
        elif pool:                                               # upconv's Conv2D(n, 2) first (elements.py:42-48), then the deconvs
            k = L['k']
            ws.append(with_norms([(T.glorot_uniform(rng, (2, 2, cin, n)), bias()),
                                  (T.glorot_uniform(rng, (k, k, n, n)), bias()),
                                  (T.glorot_uniform(rng, (k, k, n, n)), bias())], first=1))   # (upconv's conv has no norm)
        else:
            k = L['k']
            ws.append(with_norms([(T.glorot_uniform(rng, (k, k, n, cin)), bias()),
                                  (T.glorot_uniform(rng, (k, k, n, n)), bias())]))
    return ws


ACT_ALPHA = {'leakyrelu': T.LRELU_ALPHA, 'relu': 0.0, 'elu': 1.0}     # elements.py:69-75: LeakyReLU(0.3) / ReLU(0) / ELU(alpha=1)


def pixel_norm(x, eps=1.0e-8):
    """elements.py:103-121."""
    return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps)


NORM_EPS = 1.0e-3                                                # elements.py:53,56: epsilon=0.001 for both


def layer_norm(x, gamma, beta, eps=NORM_EPS):
    """tf.keras.layers.LayerNormalization(epsilon=0.001, center=True, scale=True), default axis = -1 (elements.py:55-56):
    per texel over its channels, biased variance; TF 2.2 normalises first ((x - mean) * rsqrt(var + eps), its fused path
    with scale 1 / offset 0) and applies gamma / beta afterwards."""
    m = x.mean(-1, keepdim=True)
    v = ((x - m) ** 2).mean(-1, keepdim=True)
    return (x - m) * torch.rsqrt(v + eps) * gamma + beta


def batch_norm_inference(x, gamma, beta, moving_mean=0.0, moving_var=1.0, eps=NORM_EPS):
    """tf.keras.layers.BatchNormalization(momentum=0.99, epsilon=0.001) (elements.py:52-53) AS CALLED by the reference:
    `layer(x)` without `training=` (networks/seq.py:36-41; models/nlt.py:154-195) => Keras inference mode in every mode
    of the loop: (x - moving_mean) * rsqrt(moving_variance + eps) * gamma + beta, and since only training-mode calls
    update them, the moving statistics stay at their initial 0 / 1."""
    return (x - moving_mean) * torch.rsqrt(torch.as_tensor(moving_var + eps, dtype=x.dtype)) * gamma + beta


def pool2x2(x, kind):
    """tf.keras.layers.Max/AveragePooling2D(pool_size=2, strides=2, padding='same') on even sizes (elements.py:81-94)."""
    f = torch.nn.functional.max_pool2d if kind == 'max' else torch.nn.functional.avg_pool2d
    return f(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def round_bf16(t):
    """Round to bfloat16 (nearest even) and back: the value a bf16-STORED tensor holds."""
    return t.to(torch.bfloat16).to(t.dtype)


def apply_layer_bf16(L, w, x, alpha, keep_fp32_out=False):
    """A down / up block of the bf16 region (BASELINE config 5): operands rounded to bf16 (inputs as stored, weights as
    packed), fp32 accumulation, fp32 bias + LeakyReLU, the result stored as bf16 -- except the region's last conv, whose
    output stays fp32 (keep_fp32_out).  Mirrors csrc/conv_bf16.hip layer for layer."""
    rb = round_bf16
    if L['kind'] == 'down':
        y = rb(T.leaky_relu(T.conv2d_same(rb(x), rb(w[0][0]), w[0][1], L['s']), alpha))
        return rb(T.leaky_relu(T.conv2d_same(y, rb(w[1][0]), w[1][1], 1), alpha))
    y = rb(T.leaky_relu(T.conv2d_transpose_same(rb(x), rb(w[0][0]), w[0][1], L['s']), alpha))
    z = T.leaky_relu(T.conv2d_transpose_same(y, rb(w[1][0]), w[1][1], 1), alpha)
    return z if keep_fp32_out else rb(z)


def apply_layer(L, w, x, alpha=T.LRELU_ALPHA, norm=None, pool=None, act='leakyrelu', masks=None):
    """One entry of Network.layers (convnet.py:44,50-59,67-76,85); alpha = negative slope (lrelu / relu) or ELU's alpha;
    norm in (None, 'pixel', 'layer', 'batch'); pool in (None, 'max', 'avg') -- with pooling the expanding blocks start with `upconv`.
    masks (test instrument, LeakyReLU / ReLU only): one boolean tensor per activation of the block, True where the
    activation takes its positive branch -- `where(mask, v, alpha * v)` instead of deciding on v's own sign (see
    OracleModel.act_masks)."""
    if norm in ('layer', 'batch') and L['kind'] != 'conv1x1':
        # w = [conv, (gamma, beta), conv, (gamma, beta)] (upconv: one more conv in front): split into convs and norm pairs
        convs, norms = [], []
        rest = list(w)
        if pool and L['kind'] == 'up':
            convs.append(rest.pop(0))
        while rest:
            convs.append(rest.pop(0))
            norms.append(rest.pop(0))
        w = convs
        nit = iter(norms)
        fn = layer_norm if norm == 'layer' else batch_norm_inference

        def nrm(v):
            g_, b_ = next(nit)
            return fn(v, g_, b_)
    else:
        nrm = (lambda v: pixel_norm(v)) if norm == 'pixel' else (lambda v: v)
    a = (lambda v: torch.nn.functional.elu(v, alpha)) if act == 'elu' else (lambda v: T.leaky_relu(v, alpha))
    if masks is not None:
        assert act != 'elu'
        it = iter(masks)
        a = lambda v: torch.where(next(it), v, alpha * v)
    if L['kind'] == 'conv1x1':
        return T.conv2d_same(x, w[0][0], w[0][1], 1)
    if L['kind'] == 'down':
        y = a(nrm(T.conv2d_same(x, w[0][0], w[0][1], L['s'])))
        y = a(nrm(T.conv2d_same(y, w[1][0], w[1][1], 1)))
        return pool2x2(y, pool) if pool else y
    if pool:                                                     # upconv: bilinear x2 + Conv2D(n, 2, 'same')
        x = T.conv2d_same(T.resize_bilinear(x, 2 * x.shape[1], 2 * x.shape[2]), w[0][0], w[0][1], 1)
        w = w[1:]
    y = a(nrm(T.conv2d_transpose_same(x, w[0][0], w[0][1], L['s'])))
    return a(nrm(T.conv2d_transpose_same(y, w[1][0], w[1][1], 1)))


class OracleModel:
    """nlt/models/nlt.py Model, restated.  Weights are torch-CPU leaf tensors."""

    def __init__(self, depth0=16, depth=256, kernel=2, stride=2, uvh=512, uvw=512, imh=512,
                 imw=512, use_obs=True, skip_connect_base=True, loss='l2', seed=0, dtype=torch.float32, act='leakyrelu',
                 norm=None, pool=None):
        self.layers, self.is_contracting, _ = build_layers(depth0, depth, kernel, stride)
        self.alpha = ACT_ALPHA[act]                              # config key `act` (dragon_specular.ini:61)
        self.act, self.norm, self.pool = act, norm, pool         # config keys `norm`, `pool` (dragon_specular.ini:60,62)
        # precision = 'bf16' (BASELINE config 5): layers [3, n_layers - 4] -- encoder levels >= 3 and the expanding blocks
        # with >= 64 input channels -- on bf16 operands / bf16 storage, fp32 accumulation (csrc/conv_bf16.hip)
        self.bf16_layers = None
        # Test instrument for gradient parity (not reference behaviour): {('q', layer) | ('o', layer, obs index): (mask of the
        # block's first activation, mask of its second)} -- the branch every LeakyReLU takes is DICTATED (the sign pattern of
        # the implementation under test) instead of decided by this model's own pre-activations.  LeakyReLU's derivative is
        # discontinuous: two correct evaluations that round a pre-activation to opposite sides of zero get gradients that
        # differ by a whole texel's contribution.  With the masks shared, what remains is accumulation error only.
        self.act_masks = None
        self.uvh, self.uvw, self.imh, self.imw = uvh, uvw, imh, imw
        self.use_obs, self.skip_connect_base = use_obs, skip_connect_base
        self.loss_spec = loss
        q_in, o_in = _layer_in_channels(self.layers, self.is_contracting, 5, 3, use_obs)
        rng = np.random.default_rng(seed)
        wq = init_weights(self.layers, q_in, rng, pool=bool(pool), norm=norm)
        wo = init_weights(self.layers, o_in, rng, contracting_only=self.is_contracting, pool=bool(pool), norm=norm)
        tt = lambda a: torch.tensor(a, dtype=dtype, requires_grad=True)
        self.wq = [[(tt(k), tt(b)) for k, b in lw] for lw in wq]
        self.wo = [[(tt(k), tt(b)) for k, b in lw] for lw in wo]   # obs net keeps contracting layers only (nlt.py:57-59)

    # -- flat parameter order shared with the product: query layers, then obs layers; kernel then bias
    def parameters(self):
        out = []
        for net in (self.wq, self.wo):
            for lw in net:
                for k, b in lw:
                    out += [k, b]
        return out

    def numpy_weights(self):
        conv = lambda net: [[(k.detach().numpy().copy(), b.detach().numpy().copy()) for k, b in lw] for lw in net]
        return {'query': conv(self.wq), 'obs': conv(self.wo)}

    # -- nlt/models/nlt.py:141-199
    def set_precision(self, precision):
        n = len(self.layers)
        self.bf16_layers = (3, n - 4) if precision == 'bf16' else None     # depth 256: layers 3 .. 10 of 14
        return self

    def _layer(self, i, L, w, x, path=('q',)):
        r = self.bf16_layers
        if r is not None and r[0] <= i <= r[1]:
            return apply_layer_bf16(L, w, x, self.alpha, keep_fp32_out=(i == r[1]))
        masks = None
        if self.act_masks is not None and L['kind'] != 'conv1x1':
            masks = self.act_masks[(path[0], i) + tuple(path[1:])]
        return apply_layer(L, w, x, self.alpha, self.norm, self.pool, self.act, masks=masks)

    def _call(self, query_x, obs_xs, obs_weights=None, obs_override=None, return_feats=False, layer_outputs=None):
        """layer_outputs: a list that receives the query path's output of every layer (tests of intermediate maps)."""
        feats = []
        if obs_weights is not None:
            obs_weights = obs_weights.reshape(obs_weights.shape[0], 1, 1, 1, -1)
        stack = []
        query_y = None
        for i, (L, c) in enumerate(zip(self.layers, self.is_contracting)):
            if c:
                obs_ys = [self._layer(i, L, self.wo[i], x, ('o', j)) for j, x in enumerate(obs_xs)]   # :154-155
                obs_agg = torch.stack(obs_ys, -1)                               # :161
                if obs_weights is not None:
                    obs_agg = obs_weights * obs_agg                             # :162-163
                obs_agg = obs_agg.mean(-1)                                      # :164
                if self.bf16_layers is not None and self.bf16_layers[0] <= i <= self.bf16_layers[1]:
                    obs_agg = round_bf16(obs_agg)                               # the mean is stored in the bf16 fm[l]
                obs_xs = obs_ys                                                 # :166
                query_y = self._layer(i, L, self.wq[i], query_x)                # :168
                if layer_outputs is not None:
                    layer_outputs.append(query_y)
                if self.use_obs:
                    if obs_override is not None:
                        obs_agg = obs_override[i]                               # :172-173
                    query_x = torch.cat((query_y, obs_agg), -1)                 # :174
                else:
                    query_x = query_y
                stack.append(query_x)                                           # :180
                feats.append(obs_agg)
            else:
                if stack:
                    query_x = torch.cat((query_x, stack.pop()), -1)             # :184-190
                query_y = self._layer(i, L, self.wq[i], query_x)                # :195
                if layer_outputs is not None:
                    layer_outputs.append(query_y)
                query_x = query_y
        return (query_y, feats) if return_feats else query_y

    # -- nlt/models/nlt.py:89-139
    def call(self, batch, mode='train', obs_override=None, nn_list=None):
        """batch = the reference 11-tuple (ids may be None).  `nn_list`, if given, is a
        list of (nn_base, nn_rgb) pairs and replaces the single neighbour of the
        reference (k>1 observation maps; _call already takes a list, nlt.py:154-164)."""
        if mode not in ('train', 'vali', 'test'):
            raise ValueError(mode)
        _, base, cvis, lvis, warp, rgb, rgb_camspc, _, nn_base, nn_rgb, _ = batch
        x = torch.cat((base, cvis, lvis), 3)                                    # :95
        if nn_list is None:
            y_obs = [nn_rgb - nn_base]                                          # :96
        else:
            y_obs = [r - b for b, r in nn_list]
        pred = self._call(x, y_obs, obs_override=obs_override)
        if self.skip_connect_base:
            pred = pred + base                                                  # :101-102
        warp_px = torch.stack((warp[..., 0] * self.uvw, warp[..., 1] * self.uvh), 3)  # :104-106
        fg = T.set_left_top_corner(torch.ones_like(pred), 0)                    # :107-108
        base0 = T.set_left_top_corner(base, 0)
        pred = T.set_left_top_corner(pred, 0)
        fg_c = T.resize_bilinear(T.resampler(fg, warp_px), self.imh, self.imw)  # :112-120
        base_c = T.resize_bilinear(T.resampler(base0, warp_px), self.imh, self.imw)
        pred_c = T.resize_bilinear(T.resampler(pred, warp_px), self.imh, self.imw)
        to_vis = {'base_camspc': base_c, 'pred': pred, 'pred_camspc': pred_c, 'fg_camspc': fg_c,
                  'warp_px': warp_px}
        if mode in ('train', 'vali'):
            gt_c = T.alpha_blend(rgb_camspc, fg_c)                              # :132-133
            to_vis['gt_camspc'] = gt_c
            return pred_c, gt_c, {}, to_vis
        return pred_c, None, None, to_vis

    # -- nlt/models/base.py:63-77 + nlt/models/nlt.py:66-87,201-205
    def compute_loss(self, pred, gt, keep_batch=True):
        total = 0
        for term in self.loss_spec.split(','):
            name, weight = parse_loss_and_weight(term)
            if name == 'l2':
                val = l2_loss(gt, pred, keep_batch)
            elif name == 'barron':
                val = B.barron_loss(gt, pred, keep_batch)
            else:
                raise NotImplementedError(name)
            total = total + weight * val
        return total


def parse_loss_and_weight(s):
    """nlt/models/base.py:63-77: longest float prefix is the weight."""
    for i in range(len(s), -1, -1):
        try:
            weight = float(s[:i])
        except ValueError:
            continue
        return s[i:], weight
    return s, 1.


def l2_loss(gt, pred, keep_batch=False, weights=None):
    """nlt/losses.py:39-53: MeanSquaredError(reduction='none') = mean over C, then
    mean over H,W (per example) or over everything.  weights = Keras `sample_weight` (losses.py:42-43): multiplies the
    [N,H,W] per-texel loss map (losses_utils.compute_weighted_loss: a trailing axis of 1 squeezed, a missing trailing
    axis added, then an ordinary broadcast); reduction='none' does not renormalise by the weight sum."""
    loss = ((gt - pred) ** 2).mean(-1)
    if weights is not None:
        wt = torch.as_tensor(weights, dtype=loss.dtype)
        if wt.dim() == loss.dim() + 1 and wt.shape[-1] == 1:
            wt = wt[..., 0]
        elif wt.dim() == loss.dim() - 1:
            wt = wt[..., None]
        loss = loss * wt
    return loss.mean(dim=(1, 2)) if keep_batch else loss.mean()


# ----------------------------------------------------------------------------
# Keras Adam(amsgrad=True), TF 2.2 OptimizerV2     nlt/trainvali.py:122-127,280
# ----------------------------------------------------------------------------
class KerasAdamAMSGrad:
    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        self.params = params
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.vhat = [torch.zeros_like(p) for p in params]

    def step(self, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        with torch.no_grad():
            for p, g, m, v, vh in zip(self.params, grads, self.m, self.v, self.vhat):
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                torch.maximum(vh, v, out=vh)
                p.sub_(lr_t * m / (vh.sqrt() + self.eps))


def clip_by_norm(t, clip):
    """tf.clip_by_norm (what Keras `clipnorm` applies per variable, nlt/trainvali.py:122-127): t * clip / max(||t||, clip),
    multiply first, then divide; the norm of an all-zero tensor is taken as 0."""
    l2 = (t * t).sum()
    norm = torch.sqrt(l2) if float(l2) > 0 else l2
    return (t * clip) / torch.maximum(norm, torch.as_tensor(clip, dtype=t.dtype))


def train_step(model, opt, batch, global_bs, nn_list=None, clipnorm=None):
    """nlt/trainvali.py:272-281 on one replica: per-example loss -> sum/global_bs ->
    grads (-> per-variable clipnorm when mgm > 0) -> Adam-AMSGrad.  Returns (weighted_loss, grads as applied)."""
    pred, gt, kw, _ = model.call(batch, 'train', nn_list=nn_list)
    per_example = model.compute_loss(pred, gt, keep_batch=True)
    weighted = per_example.sum() / global_bs                 # tf.nn.compute_average_loss
    params = model.parameters()
    grads = torch.autograd.grad(weighted, params)
    if clipnorm is not None and clipnorm > 0:
        grads = tuple(clip_by_norm(g, clipnorm) for g in grads)
    opt.step(grads)
    return weighted.detach(), grads


# ----------------------------------------------------------------------------
# synthetic batches
# ----------------------------------------------------------------------------
def synth_batch(n, uvh, uvw, hc, wc, imh, imw, k=1, seed=0, identity_warp=False, fg_frac=0.7,
                quantize_vis=False):
    """Seeded stand-in for datasets/nlt.py:_load_data outputs.  Returns (batch11, nn_list)."""
    rng = np.random.default_rng(seed)
    U = lambda *s: rng.random(s, dtype=np.float32)
    base, rgb = U(n, uvh, uvw, 3), U(n, uvh, uvw, 3)
    cvis, lvis = U(n, uvh, uvw, 1), U(n, uvh, uvw, 1)
    if quantize_vis:
        cvis = np.floor(255 * cvis) / np.float32(255); lvis = np.floor(255 * lvis) / np.float32(255)
    rgb_c = U(n, imh, imw, 3)
    if identity_warp:
        jj, ii = np.meshgrid(np.arange(wc, dtype=np.float32), np.arange(hc, dtype=np.float32))
        warp = np.stack((jj / np.float32(wc), ii / np.float32(hc)), -1)[None].repeat(n, 0)
        warp = warp.astype(np.float32)
    else:
        warp = U(n, hc, wc, 2).astype(np.float16).astype(np.float32)   # save_float16_npy (data_gen/util.py:67-70)
        bg = rng.random((n, hc, wc)) >= fg_frac
        warp[bg] = 0                                                   # render.py:155
    nn_list = [(U(n, uvh, uvw, 3), U(n, uvh, uvw, 3)) for _ in range(k)]
    nn_rgb_c = U(n, imh, imw, 3)
    tt = torch.from_numpy
    batch = (None, tt(base), tt(cvis), tt(lvis), tt(warp), tt(rgb), tt(rgb_c), None,
             tt(nn_list[0][0]), tt(nn_list[0][1]), tt(nn_rgb_c))
    nn_t = [(tt(b), tt(r)) for b, r in nn_list]
    return batch, nn_t
