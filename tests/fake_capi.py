"""TEST-ONLY stand-in for the Python-level `nlt_amd._capi` adapters, executing each C-ABI op
with the oracle's torch-CPU primitives on CPU tensors.  It honours the pointer/stride calling
convention (channel slices of wider tensors), so the host orchestration (engine buffers,
virtual concat, slices, labels) can be verified without a GPU.  Never used by the product."""
import torch

from nlt_amd import _capi as C
from oracle import tf_ops as T
from oracle import nlt_oracle as O

_MODES = {C.CONV1X1: (1, False), C.CONV_K2S2: (2, False), C.CONV_K2S1: (1, False),
          C.DECONV_K2S2: (2, True), C.DECONV_K2S1: (1, True)}


def _view(t, n, h, w, c, ld):
    return torch.as_strided(t, (n, h, w, c), (h * w * ld, w * ld, ld, 1))


def conv_forward(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, w_keras, w_packed, bias, cout, out, ldo,
                 act=True, alpha=0.3, algo=0, tile_hint=0, mask_src=None, ldm=0, accumulate=False):
    x = _view(src0, n, h, w, c0, ld0)
    if c1:
        x = torch.cat((x, _view(src1, n, h, w, c1, ld1)), -1)
    s, tr = _MODES[mode]
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    y = f(x, w_keras, bias[:cout], s)
    oh, ow = y.shape[1:3]
    o = _view(out, n, oh, ow, cout, ldo)
    if accumulate:
        y = y + o
    if mask_src is not None:
        y = y * torch.where(_view(mask_src, n, oh, ow, cout, ldm) > 0, 1.0, alpha)
    elif act:
        y = T.leaky_relu(y, alpha)
    o.copy_(y)


def conv_backward_data(adj_mode, dpre, cpre, ldp, n, h, w, w_packed, zero_bias, cout, out, ldo, mask_src=None, ldm=0,
                       mask_alpha=0.3, accumulate=False, tile_hint=0, ksplit=1, split=None, w_keras=None):
    assert ksplit == 1, "split-K is a GPU-only plan choice; the CPU emulation never selects it"
    x = _view(dpre, n, h, w, cpre, ldp)
    s, tr = _MODES[adj_mode]
    y = (T.conv2d_transpose_same if tr else T.conv2d_same)(x, w_keras, zero_bias[:cout], s)
    oh, ow = y.shape[1:3]
    o = _view(out, n, oh, ow, cout, ldo)
    if accumulate:
        y = y + o
    nq = cout
    if split is not None:                                       # observation half of dfm[l] -> finished dobs (k = 1)
        c, sy, sd, sa, sp = split
        nq = c
        d = sd.view(n, oh, ow, c)
        yo = y[..., c:] + (d if sp else 0)
        d.copy_(yo * torch.where(sy.view(n, oh, ow, c) > 0, 1.0, sa))
    yq = y[..., :nq]
    if mask_src is not None:
        yq = yq * torch.where(_view(mask_src, n, oh, ow, nq, ldm) > 0, 1.0, mask_alpha)
    o[..., :nq].copy_(yq)


def pack_conv_weights(mode, w_keras, c0, c1, cout):
    return torch.zeros(1)


def repack_table(entries, device):
    return (None, len(entries), 0)


def repack_weights(table, n_desc, total_blocks):
    pass                                   # the emulated kernels read the Keras arrays themselves


def stem_forward(base, cvis, lvis, nn_rgb, nn_base, obs_weights, n, k, h, w, c, wq, bq, wo, bo, fm0, obs0):
    x = torch.cat((base, cvis, lvis), -1)
    fm0[..., :c] = x @ wq[0, 0] + bq
    o = (nn_rgb - nn_base) @ wo[0, 0] + bo
    obs0.copy_(o)
    if obs_weights is not None:
        o = o * obs_weights[:, :, None, None, None]
    fm0[..., c:] = o.mean(1)


def obs_mean_forward(obs, obs_weights, n, k, hw, c, out, ldo):
    o = obs.reshape(n, k, hw, c)
    if obs_weights is not None:
        o = o * obs_weights[:, :, None, None]
    torch.as_strided(out, (n, hw, c), (hw * ldo, ldo, 1)).copy_(o.mean(1))


def head_forward(dec, ldd, cd, skip, lds, cs, w_keras, bias, base, n, h, w, pred):
    x = _view(dec, n, h, w, cd, ldd)
    if cs:
        x = torch.cat((x, _view(skip, n, h, w, cs, lds)), -1)
    y = x @ w_keras[0, 0] + bias
    if base is not None:
        y = y + base
    y[:, 0, 0, :] = 0
    pred.copy_(y)


def warp_forward(pred, base, warp, n, uvh, uvw, hc, wc, pred_cam, base_cam, fg_cam, idx_out=None):
    wpx = warp * torch.tensor([uvw, uvh], dtype=torch.float32)
    if pred_cam is not None:
        pred_cam.copy_(T.resampler(pred, wpx))
    if base_cam is not None:
        base_cam.copy_(T.resampler(T.set_left_top_corner(base, 0), wpx))
    if fg_cam is not None:
        fg_cam.copy_(T.resampler(T.set_left_top_corner(torch.ones_like(pred), 0), wpx))
    if idx_out is not None:
        fx, fy, inside = T.resampler_indices(wpx.numpy(), uvh, uvw)
        idx_out[..., 0] = torch.from_numpy(fx); idx_out[..., 1] = torch.from_numpy(fy)
        idx_out[..., 2] = torch.from_numpy(inside.astype('int32')); idx_out[..., 3] = 0


def resize_bilinear_forward(x, oh, ow):
    return T.resize_bilinear(x, oh, ow).contiguous()


def mul_forward(a, b):
    return a * b



# ------------------------------------------------------------------ train-step ops (TEST-ONLY emulation)
from oracle import barron as _B


def conv_backward_weights(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp, cout, dw, db, algo=0):
    x = _view(src0, n, h, w, c0, ld0)
    if c1:
        x = torch.cat((x, _view(src1, n, h, w, c1, ld1)), -1)
    s, tr = _MODES[mode]
    f = T.conv2d_transpose_same if tr else T.conv2d_same
    with torch.enable_grad():
        wz = torch.zeros_like(dw, requires_grad=True)
        bz = torch.zeros(cout, requires_grad=True)
        y = f(x.detach(), wz, bz, s)
        g = _view(dpre, n, y.shape[1], y.shape[2], cout, ldp)
        gw, gb = torch.autograd.grad(y, (wz, bz), g)
    dw += gw
    if db is not None:
        db += gb


def lrelu_backward(g, ldg, y, ldy, c, texels, alpha, out, ldo):
    gv = torch.as_strided(g, (texels, c), (ldg, 1))
    yv = torch.as_strided(y, (texels, c), (ldy, 1))
    torch.as_strided(out, (texels, c), (ldo, 1)).copy_(gv * torch.where(yv > 0, 1.0, alpha))


def obs_mean_backward(dmean, ldm, obs_y, obs_weights, dobs_partial, n, k, hw, c, alpha, dpre_obs):
    dm = torch.as_strided(dmean, (n, hw, c), (hw * ldm, ldm, 1)).unsqueeze(1) / k
    g = dm.expand(n, k, hw, c)
    if obs_weights is not None:
        g = g * obs_weights[:, :, None, None]
    if dobs_partial is not None:
        g = g + dobs_partial.reshape(n, k, hw, c)
    if obs_y is not None:
        g = g * torch.where(obs_y.reshape(n, k, hw, c) > 0, 1.0, alpha)
    dpre_obs.copy_(g.reshape(dpre_obs.shape))


def l2_train_loss(pred, rgb, fg, global_bs):
    gt = rgb * fg
    per = pred[0].numel()
    d = pred - gt
    loss = ((d * d).reshape(pred.shape[0], -1).sum(1) / per).sum() / global_bs
    return loss, gt, 2.0 * d / per / global_bs


def level_split_backward(dfm, fm_y, ld, obs_y, obs_weights, dobs_partial, n, k, hw, c, alpha_q, alpha_o, dpre_obs):
    obs_mean_backward(dfm.view(-1)[c:], ld, obs_y, obs_weights, dobs_partial, n, k, hw, c, alpha_o, dpre_obs)
    lrelu_backward(dfm, ld, fm_y, ld, c, n * hw, alpha_q, dfm, ld)


def stem_backward(base, cvis, lvis, nn_rgb, nn_base, obs_weights, n, k, h, w, c, dfm0, dobs0, dwq, dbq, dwo, dbo):
    x = torch.cat((base, cvis, lvis), -1).reshape(-1, 5)
    gq = dfm0[..., :c].reshape(-1, c)
    dwq += (x.t() @ gq).reshape(dwq.shape)
    dbq += gq.sum(0)
    gm = (dfm0[..., c:] / k).unsqueeze(1).expand(n, k, h, w, c)
    if obs_weights is not None:
        gm = gm * obs_weights[:, :, None, None, None]
    g = gm + (dobs0 if dobs0 is not None else 0)
    d = (nn_rgb - nn_base).reshape(-1, 3)
    dwo += (d.t() @ g.reshape(-1, c)).reshape(dwo.shape)
    dbo += g.reshape(-1, c).sum(0)


def head_backward(dec, ldd, cd, skip, lds, cs, w_keras, dpred, n, h, w, d_dec, ldgd, d_skip, ldgs, dw, db):
    x = _view(dec, n, h, w, cd, ldd)
    if cs:
        x = torch.cat((x, _view(skip, n, h, w, cs, lds)), -1)
    g = dpred.clone()
    g[:, 0, 0, :] = 0
    dx = g @ w_keras[0, 0, :cd + cs].t()
    _view(d_dec, n, h, w, cd, ldgd).copy_(dx[..., :cd])
    if cs:
        _view(d_skip, n, h, w, cs, ldgs).copy_(dx[..., cd:])
    dw.view(-1, 3)[:cd + cs] += x.reshape(-1, cd + cs).t() @ g.reshape(-1, 3)      # cs = 0: the first cd rows only
    db += g.reshape(-1, 3).sum(0)


def warp_backward(dpred_cam, warp, n, uvh, uvw, hc, wc, dpred):
    wpx = warp * torch.tensor([uvw, uvh], dtype=torch.float32)
    with torch.enable_grad():
        data = torch.zeros((n, uvh, uvw, 3), requires_grad=True)
        out = T.resampler(T.set_left_top_corner(data, 0), wpx)
        (g,) = torch.autograd.grad(out, data, dpred_cam)
    dpred.copy_(g)


def resize_bilinear_backward(dout, h, w):
    n, oh, ow, c = dout.shape
    with torch.enable_grad():
        x = torch.zeros((n, h, w, c), requires_grad=True)
        (g,) = torch.autograd.grad(T.resize_bilinear(x, oh, ow), x, dout)
    return g


def l2_loss_forward(pred, gt):
    return ((gt - pred) ** 2).mean(dim=(1, 2, 3))


def l2_loss_backward(pred, gt, gloss):
    return gloss.view(-1, 1, 1, 1) * 2 * (pred - gt) / pred[0].numel()


def barron_loss(pred, gt, want_grad):
    p = pred.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        loss = _B.barron_loss(gt, p, keep_batch=True)
        dunit = torch.autograd.grad(loss.sum(), p)[0] if want_grad else None
    return loss.detach(), dunit


def scale_rows(x, scale):
    return x * scale.view(-1, *([1] * (x.dim() - 1)))


def clip_by_norm_slots(grad, slots, clipnorm):
    for off, cnt in slots.tolist():
        g = grad[off:off + cnt]
        g.copy_(O.clip_by_norm(g.clone(), clipnorm))


def adam_amsgrad_step(param, grad, m, v, vhat, lr_t, beta1, beta2, eps):
    with torch.no_grad():
        m.mul_(beta1).add_(grad, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        torch.maximum(vhat, v, out=vhat)
        param.sub_(lr_t * m / (vhat.sqrt() + eps))


_TRAIN = ('conv_backward_weights', 'lrelu_backward', 'obs_mean_backward', 'level_split_backward', 'l2_train_loss', 'stem_backward', 'head_backward',
          'warp_backward', 'resize_bilinear_backward', 'l2_loss_forward', 'l2_loss_backward', 'barron_loss',
          'scale_rows', 'adam_amsgrad_step', 'clip_by_norm_slots')


# ------------------------------------------------------------------ texel-buffer assembly (TEST-ONLY emulation)
from oracle import buffers as _BU
import numpy as _np


def _t(a):
    return torch.from_numpy(_np.ascontiguousarray(a))


def cosine_map(locs, normals, valid, occluded, src_loc, want_float=True, want_u8=True):
    shp = tuple(valid.shape)
    ref = _BU.cosine_map(src_loc, locs.numpy().reshape(shp + (3,)), normals.numpy().reshape(shp + (3,)), valid.numpy(),
                         None if occluded is None else occluded.numpy())
    return (_t(ref) if want_float else None), (_t(_BU.quantize_unit(ref)) if want_u8 else None)


def albedo(rgb_frames):
    return _t(_BU.albedo_from_frames(rgb_frames.numpy()))


def diffuse_base(albedo_, lvis):
    return _t(_np.stack([_BU.diffuse_base(albedo_.numpy(), l) for l in lvis.numpy()]))


def remap_bilinear(src, mapping, force_kbg=True):
    f = _BU.remap_u8 if src.dtype == torch.uint8 else _BU.remap_f32
    return _t(f(src.numpy(), mapping.numpy(), force_kbg))


def uv_index_map(uvs, values, h, w, max_l1=4, fill=0.0, want_index=False):
    out, idx = _BU.uv_index_map(uvs.numpy(), values.numpy(), (h, w), max_l1, fill, return_index=True)
    return (_t(out), _t(idx)) if want_index else _t(out)


def knn_indices(ref_pos, cand_pos, k=1):
    return _t(_BU.knn_indices(ref_pos.numpy(), cand_pos.numpy(), k))


def warp_forward_store(pred, diffuse_store, uv2cam_store, ids, n, uvh, uvw, hc, wc, pred_cam, base_cam, fg_cam, idx_out=None):
    base = gather_frames_u8(diffuse_store, ids)
    warp_forward(pred, base, uv2cam_store[ids.long()].float(), n, uvh, uvw, hc, wc, pred_cam, base_cam, fg_cam, idx_out)


def gather_frames_u8(store, ids, out=None):
    i = ids.numpy()
    res = (store.numpy()[_np.maximum(i, 0)] / 255.0).astype(_np.float32)
    res[i < 0] = 0
    if out is not None:
        out.copy_(_t(res))
        return out
    return _t(res)


def assemble_batch(diffuse_store, rgb_store, cvis_store, lvis_store, ids, nn_ids, test_mode=False, out=None):
    st = {'diffuse': diffuse_store.numpy(), 'rgb': rgb_store.numpy(), 'cvis': cvis_store.numpy(), 'lvis': lvis_store.numpy()}
    nn = _np.zeros((ids.numel(), 0), _np.int32) if nn_ids is None else nn_ids.numpy()
    b = {k: _t(v) for k, v in _BU.assemble_batch(st, ids.numpy(), nn, 'test' if test_mode else 'train').items()}
    if out is not None:
        for k, v in b.items():
            if v is not None:
                out[k].copy_(v)
        return out
    return b


def resize_cv_linear(src, oh, ow, out=None):
    a = src.numpy()
    norm = {torch.uint8: 255.0, torch.int32: 65535.0, torch.float32: 1.0}[src.dtype]
    res = _np.stack([_BU.cv_resize_linear(f.astype(_np.float64) / norm, oh, ow) for f in a]).astype(_np.float32)
    res = torch.from_numpy(res)
    if out is not None:
        out.copy_(res)
        return out
    return res


_BUFFERS = ('resize_cv_linear', 'cosine_map', 'albedo', 'diffuse_base', 'remap_bilinear', 'uv_index_map', 'knn_indices', 'gather_frames_u8', 'warp_forward_store',
            'assemble_batch')


# ------------------------------------------------------------------ fused inference ends (TEST-ONLY emulation)
def front_pack_weights(wq0, bq0, wo0, bo0, wqa, bqa, wqb, bqb, woa, boa, wob, bob, wh, bh, out=None):
    return dict(wq0=wq0, bq0=bq0, wo0=wo0, bo0=bo0, wqa=wqa, bqa=bqa, wqb=wqb, bqb=bqb, woa=woa, boa=boa,
                wob=wob, bob=bob, wh=wh, bh=bh)


def front_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, P, add_base, alpha, fm1, obs1, skip3):
    """Layer-by-layer (UNfolded) evaluation of what the front kernel computes."""
    lr = lambda x: T.leaky_relu(x, alpha)
    q0 = torch.cat((base, cvis, lvis), -1) @ P['wq0'][0, 0] + P['bq0']
    o0 = (nn_rgb - nn_base) @ P['wo0'][0, 0] + P['bo0']                  # [n,k,h,w,16]
    fm0 = torch.cat((q0, o0.mean(1)), -1)
    q1 = lr(T.conv2d_same(lr(T.conv2d_same(fm0, P['wqa'], P['bqa'], 2)), P['wqb'], P['bqb'], 1))
    o1 = torch.stack([lr(T.conv2d_same(lr(T.conv2d_same(o0[:, i], P['woa'], P['boa'], 2)), P['wob'], P['bob'], 1))
                      for i in range(k)], 1)
    fm1.copy_(torch.cat((q1, o1.mean(1)), -1))
    obs1.copy_(o1)
    s = fm0 @ P['wh'][0, 0, 4:, :] + P['bh']
    skip3.copy_(s + base if add_base else s)


def back_forward(x, fm1, skip3, n, h2, w2, w_s2, b_s2, w_s1, b_s1, w_head, alpha, pred):
    lr = lambda t: T.leaky_relu(t, alpha)
    d = lr(T.conv2d_transpose_same(lr(T.conv2d_transpose_same(torch.cat((x, fm1), -1), w_s2, b_s2, 2)), w_s1, b_s1, 1))
    y = d @ w_head.reshape(-1, 3)[:4] + skip3
    y[:, 0, 0, :] = 0
    pred.copy_(y)


def front_forward_train(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, P, add_base, alpha, fm1, obs1, skip3, qtmp1, otmp1):
    front_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, P, add_base, alpha, fm1, obs1, skip3)
    lr = lambda x: T.leaky_relu(x, alpha)
    q0 = torch.cat((base, cvis, lvis), -1) @ P['wq0'][0, 0] + P['bq0']
    o0 = (nn_rgb - nn_base) @ P['wo0'][0, 0] + P['bo0']
    qtmp1.copy_(lr(T.conv2d_same(torch.cat((q0, o0.mean(1)), -1), P['wqa'], P['bqa'], 2)))
    otmp1.copy_(torch.stack([lr(T.conv2d_same(o0[:, i], P['woa'], P['boa'], 2)) for i in range(k)], 1))


def front4_forward_train(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, P, P2, add_base, alpha, fm1, skip3, qtmp2, otmp2,
                         obs1, qtmp1, otmp1):
    front_forward_train(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, P, add_base, alpha, fm1, obs1, skip3, qtmp1, otmp1)
    lr = lambda x: T.leaky_relu(x, alpha)
    qtmp2.copy_(lr(T.conv2d_same(fm1, P2['wq'], P2['bq'], 2)))
    otmp2.copy_(torch.stack([lr(T.conv2d_same(obs1[:, i], P2['wo'], P2['bo'], 2)) for i in range(k)], 1))


def back_forward_train(x, fm1, skip3, n, h2, w2, w_s2, b_s2, w_s1, b_s1, w_head, alpha, pred, u, v):
    lr = lambda t: T.leaky_relu(t, alpha)
    u.copy_(lr(T.conv2d_transpose_same(torch.cat((x, fm1), -1), w_s2, b_s2, 2)))
    v.copy_(lr(T.conv2d_transpose_same(u, w_s1, b_s1, 1)))
    y = v @ w_head.reshape(-1, 3)[:4] + skip3
    y[:, 0, 0, :] = 0
    pred.copy_(y)


def front_backward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, dy1q, dy1o, dpred, weights, grads):
    """Autograd through the (unfolded) linear part: L0, the PRE-activation stride-2 convs of level 1, the head's skip rows."""
    g = dpred.clone()
    g[:, 0, 0, :] = 0
    with torch.enable_grad():
        wq0, bq0, wo0, bo0, wqa, woa, wh = [t.detach().clone().requires_grad_(True) for t in weights]
        q0 = torch.cat((base, cvis, lvis), -1) @ wq0[0, 0] + bq0
        o0 = (nn_rgb - nn_base) @ wo0[0, 0] + bo0
        fm0 = torch.cat((q0, o0.mean(1)), -1)
        zero = torch.zeros(16)
        s = (T.conv2d_same(fm0, wqa, zero, 2) * dy1q).sum() + (fm0 @ wh[0, 0, 4:, :] * g).sum()
        for i in range(k):
            s = s + (T.conv2d_same(o0[:, i], woa, zero, 2) * dy1o.reshape(n, k, h // 2, w // 2, 16)[:, i]).sum()
        gr = torch.autograd.grad(s, (wq0, bq0, wo0, bo0, wqa, woa, wh))
    dwq0, dbq0, dwo0, dbo0, dwqa, dbqa, dwoa, dboa, dwh = grads
    for dst, src in ((dwq0, gr[0]), (dbq0, gr[1]), (dwo0, gr[2]), (dbo0, gr[3]), (dwqa, gr[4]), (dwoa, gr[5]), (dwh, gr[6])):
        dst += src.reshape(dst.shape)
    dbqa += dy1q.reshape(-1, 16).sum(0)
    dboa += dy1o.reshape(-1, 16).sum(0)


def back_backward(x, fm1, u, v, dpred, n, h2, w2, w_s2, w_s1, w_head, alpha, dx, dfm1, dw_s2, db_s2, dw_s1, db_s1, dw_head, db_head):
    g = dpred.clone()
    g[:, 0, 0, :] = 0
    with torch.enable_grad():
        xin = torch.cat((x, fm1), -1).detach().clone().requires_grad_(True)
        ws2, ws1, wh = [t.detach().clone().requires_grad_(True) for t in (w_s2, w_s1, w_head)]
        b2, b1 = torch.zeros(4, requires_grad=True), torch.zeros(4, requires_grad=True)
        # values = the saved activations (the biases are not passed in), gradients = through the linear maps and the
        # LeakyReLU slopes those activations imply
        slope = lambda a: torch.where(a > 0, torch.ones_like(a), torch.full_like(a, alpha))
        zu = T.conv2d_transpose_same(xin, ws2, b2, 2) * slope(u)
        uu = u.detach() + (zu - zu.detach())
        zv = T.conv2d_transpose_same(uu, ws1, b1, 1) * slope(v)
        vv = v.detach() + (zv - zv.detach())
        s = ((vv @ wh[0, 0, :4]) * g).sum()
        gr = torch.autograd.grad(s, (xin, ws2, b2, ws1, b1, wh))
    dx.copy_(gr[0][..., :8] * slope(x)); dfm1.copy_(gr[0][..., 8:])          # dx: w.r.t. the producer's pre-activation
    dw_s2 += gr[1]; db_s2 += gr[2]; dw_s1 += gr[3]; db_s1 += gr[4]; dw_head += gr[5]
    db_head += g.reshape(-1, 3).sum(0)


_FUSED = ('front_pack_weights', 'front_forward', 'back_forward', 'front_forward_train', 'back_forward_train', 'front_backward',
          'back_backward')


# ------------------------------------------------------------------ LDS-tiled encoder convs (TEST-ONLY emulation)
def pack_conv_tile_weights(mode, w_keras, cin, cout, tn):
    assert cin % 16 == 0 and cout % tn == 0 and (tn in (32, 64) or (tn == 128 and mode == C.CONV_K2S2))
    return w_keras


def conv_tile_forward(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, tn, out, ldo, mean_out, ldm,
                      act=True, alpha=0.3):
    nf = frames * kobs
    x = _view(src, nf, h, w, cin, ld)
    y = T.conv2d_same(x, packed, bias[:cout], 2 if mode == C.CONV_K2S2 else 1)
    if act:
        y = T.leaky_relu(y, alpha)
    oh, ow = y.shape[1:3]
    if out is not None:
        _view(out, nf, oh, ow, cout, ldo).copy_(y)
    if mean_out is not None:
        _view(mean_out, frames, oh, ow, cout, ldm).copy_(y.reshape(frames, kobs, oh, ow, cout).mean(1))


# ------------------------------------------------------------------ Winograd stride-1 k2 convs (TEST-ONLY emulation)
def pack_conv_wino_weights(mode, w_keras, cin, cout, tn, full=None, lo=0):
    assert mode in (C.CONV_K2S1, C.DECONV_K2S1) and cin % 8 == 0 and cout % tn == 0 and tn in (32, 64)
    if full is None:
        return w_keras
    return w_keras[:, :, lo:lo + cout, :] if mode == C.DECONV_K2S1 else w_keras[..., lo:lo + cout]     # adjoint family: a slice


def conv_wino_forward(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, tn, out, ldo, mean_out, ldm, act=True, alpha=0.3):
    assert (kobs == 1 and mean_out is None) or mode == C.CONV_K2S1
    nf = frames * kobs
    x = _view(src, nf, h, w, cin, ld)
    y = (T.conv2d_transpose_same if mode == C.DECONV_K2S1 else T.conv2d_same)(x, packed, bias[:cout], 1)
    if act:
        y = T.leaky_relu(y, alpha)
    if out is not None:
        _view(out, nf, h, w, cout, ldo).copy_(y)
    if mean_out is not None:
        _view(mean_out, frames, h, w, cout, ldm).copy_(y.reshape(frames, kobs, h, w, cout).mean(1))


def conv_wino_backward_data(adj_mode, dpre, cpre, ldp, n, h, w, packed, cout, tn, out, ldo, mask_src=None, ldm=0, mask_alpha=0.3,
                            accumulate=False):
    x = _view(dpre, n, h, w, cpre, ldp)
    y = (T.conv2d_transpose_same if adj_mode == C.DECONV_K2S1 else T.conv2d_same)(x, packed, torch.zeros(cout), 1)
    o = _view(out, n, h, w, cout, ldo)
    if accumulate:
        y = y + o
    if mask_src is not None:
        y = y * torch.where(_view(mask_src, n, h, w, cout, ldm) > 0, 1.0, mask_alpha)
    o.copy_(y)


def conv_c32_supported(mode, cin, cout):
    return mode == C.CONV_K2S1 and cin in (16, 32) and cout == 32


def conv_c32_forward(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, out, ldo, mean_out, ldm, act=True, alpha=0.3):
    assert conv_c32_supported(mode, cin, cout)
    conv_tile_forward(mode, src, ld, cin, frames, kobs, h, w, packed, bias, cout, 32, out, ldo, mean_out, ldm, act, alpha)


_TILE = ('pack_conv_tile_weights', 'conv_tile_forward', 'pack_conv_wino_weights', 'conv_wino_forward', 'conv_wino_backward_data',
         'conv_c32_supported', 'conv_c32_forward')


def conv_forward_splitk(mode, ksplit, src0, c0, ld0, src1, c1, ld1, n, h, w, w_packed, bias, cout, out, ldo,
                        act=True, alpha=0.3, tile_hint=0, mask_src=None, ldm=0, accumulate=False):
    raise AssertionError("split-K is a GPU-only plan choice; the CPU emulation never selects it")


def conv_backward_weights_tiled(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp, cout, dw, db):
    conv_backward_weights(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp, cout, dw, db)


def conv_backward_weights_narrow(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp, cout, dw, db):
    conv_backward_weights(mode, src0, c0, ld0, src1, c1, ld1, n, h, w, dpre, ldp, cout, dw, db)


def front_pack_l2_weights(wq, bq, wo, bo, out=None):
    return dict(wq=wq, bq=bq, wo=wo, bo=bo)


def front2_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, P, P2, add_base, alpha, fm1, skip3, qtmp2, otmp2):
    obs1 = torch.empty((n, k, h // 2, w // 2, 16))
    front_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, P, add_base, alpha, fm1, obs1, skip3)
    lr = lambda x: T.leaky_relu(x, alpha)
    qtmp2.copy_(lr(T.conv2d_same(fm1, P2['wq'], P2['bq'], 2)))
    otmp2.copy_(torch.stack([lr(T.conv2d_same(obs1[:, i], P2['wo'], P2['bo'], 2)) for i in range(k)], 1))


def act_forward(x, kind, alpha):
    return torch.nn.functional.elu(x, alpha) if kind == 1 else T.leaky_relu(x, alpha)


def act_backward(g, y, kind, alpha):
    return g * torch.where(y > 0, torch.ones_like(y), (y + alpha) if kind == 1 else torch.full_like(y, alpha))


def pixelnorm_forward(x, eps=1e-8):
    return O.pixel_norm(x, eps)


def pixelnorm_backward(g, x, eps=1e-8):
    with torch.enable_grad():
        xx = x.detach().clone().requires_grad_(True)
        return torch.autograd.grad(O.pixel_norm(xx, eps), xx, g)[0]


def _norm_fn(kind, x, gamma, beta, mean, var, eps):
    if kind == 0:
        return O.layer_norm(x, gamma, beta, eps)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def norm_forward(kind, x, gamma, beta, mean, var, eps):
    return _norm_fn(kind, x, gamma, beta, mean, var, eps)


def norm_backward(kind, g, x, gamma, mean, var, eps, dgamma, dbeta):
    with torch.enable_grad():
        xx = x.detach().clone().requires_grad_(True)
        gg = gamma.detach().clone().requires_grad_(True)
        bb = torch.zeros_like(gg).requires_grad_(True)
        dx, dg, db = torch.autograd.grad(_norm_fn(kind, xx, gg, bb, mean, var, eps), (xx, gg, bb), g)
    dgamma += dg
    dbeta += db
    return dx


def pool2x2_forward(x, kind):
    return O.pool2x2(x, 'max' if kind == 0 else 'avg').contiguous()


def pool2x2_backward(g, x, kind):
    with torch.enable_grad():
        xx = x.detach().clone().requires_grad_(True)
        return torch.autograd.grad(O.pool2x2(xx, 'max' if kind == 0 else 'avg'), xx, g)[0]


def resize_bilinear_backward_(dout, h, w):
    return resize_bilinear_backward(dout, h, w)


def sub_forward(a, b):
    return a - b


def finish_pred(y, base, pred):
    v = y + base if base is not None else y.clone()
    pred.copy_(T.set_left_top_corner(v, 0))


def dec_block_forward(x, cx, skip, cs, n, h, w, w_s2, b_s2, w_s1, b_s1, c, alpha, out):
    lr = lambda v: T.leaky_relu(v, alpha)
    u = lr(T.conv2d_transpose_same(torch.cat((x, skip), 3), w_s2, b_s2, 2))
    out.copy_(lr(T.conv2d_transpose_same(u, w_s1, b_s1, 1)))


def front4_supported(*tensors):
    return True


def front4_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, P, P2, add_base, alpha, fm1, skip3, qtmp2, otmp2,
                   waves_per_simd=0):
    front2_forward(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, P, P2, add_base, alpha, fm1, skip3, qtmp2, otmp2)


def front4_forward_u8(diffuse_store, rgb_store, cvis_store, lvis_store, ids, nn_ids, n, k, h, w, P, P2, add_base, alpha,
                      fm1, skip3, qtmp2, otmp2, waves_per_simd=0):
    b = assemble_batch(diffuse_store, rgb_store, cvis_store, lvis_store, ids, nn_ids)
    front2_forward(b['base'], b['cvis'], b['lvis'], b['nn_rgb'], b['nn_base'], n, k, h, w, P, P2, add_base, alpha, fm1, skip3,
                   qtmp2, otmp2)


def conv_forward_map(mode, ksplit, src0, c0, ld0, src1, c1, ld1, n, h, w, w_packed, bias, cout, out, ldo, bias_map,
                     act=True, alpha=0.3, tile_hint=0, w_keras=None):
    x = _view(src0, n, h, w, c0, ld0)
    if c1:
        x = torch.cat((x, _view(src1, n, h, w, c1, ld1)), -1)
    s, tr = _MODES[mode]
    y = (T.conv2d_transpose_same if tr else T.conv2d_same)(x, w_keras, bias[:cout], s) + bias_map
    if act:
        y = T.leaky_relu(y, alpha)
    _view(out, n, y.shape[1], y.shape[2], cout, ldo).copy_(y)


def front_ovr_forward(base, cvis, lvis, n, h, w, P, P2, p1, s0, p2, add_base, alpha, q1, ldq, skip3, qtmp2):
    """Layer-by-layer evaluation of csrc/front_ovr.hip: the query rows of L0 -> L1 -> L2's stride-2 conv, the given maps'
    share (and every bias upstream of an activation) arriving through p1 / s0 / p2."""
    lr = lambda x: T.leaky_relu(x, alpha)
    z = lambda c: torch.zeros(c)
    q0 = torch.cat((base, cvis, lvis), -1) @ P['wq0'][0, 0]                  # (L0's bias lives in the maps)
    y1 = lr(T.conv2d_same(q0, P['wqa'][:, :, :16, :].contiguous(), z(16), 2) + p1)
    v = lr(T.conv2d_same(y1, P['wqb'], P['bqb'], 1))
    _view(q1, n, h // 2, w // 2, 16, ldq).copy_(v)
    qtmp2.copy_(lr(T.conv2d_same(v, P2['wq'][:, :, :16, :].contiguous(), z(32), 2) + p2))
    sk = q0 @ P['wh'][0, 0, 4:20, :] + s0[..., :3]
    skip3.copy_(sk + base if add_base else sk)


def front_ovr_forward_u8(diffuse_store, cvis_store, lvis_store, ids, n, h, w, P, P2, p1, s0, p2, add_base, alpha, q1, ldq, skip3, qtmp2):
    idx = ids.long()
    f = lambda st: st[idx].double().div(255.0).float()                                # datasets/nlt.py `_load_data`: uint8 image / 255
    front_ovr_forward(f(diffuse_store), f(cvis_store).unsqueeze(-1), f(lvis_store).unsqueeze(-1), n, h, w, P, P2, p1, s0, p2,
                      add_base, alpha, q1, ldq, skip3, qtmp2)


def dec_block_forward_map(x, skip, lds, n, h, w, w_s2q, w_s1, b_s1, c, alpha, bias_map, out):
    lr = lambda v: T.leaky_relu(v, alpha)
    xin = torch.cat((x, _view(skip, n, h, w, 4 * c, lds)), 3)
    u = lr(T.conv2d_transpose_same(xin, w_s2q, torch.zeros(c), 2) + bias_map)
    out.copy_(lr(T.conv2d_transpose_same(u, w_s1, b_s1, 1)))


def back_forward_map(x, q1, ldq, skip3, n, h2, w2, w_s2q, w_s1, b_s1, w_head, alpha, bias_map, pred):
    lr = lambda t: T.leaky_relu(t, alpha)
    xin = torch.cat((x, _view(q1, n, h2, w2, 16, ldq)), 3)
    d = lr(T.conv2d_transpose_same(lr(T.conv2d_transpose_same(xin, w_s2q, torch.zeros(4), 2) + bias_map), w_s1, b_s1, 1))
    y = d @ w_head.reshape(-1, 3)[:4] + skip3
    y[:, 0, 0, :] = 0
    pred.copy_(y)


_FUSED = _FUSED + ('conv_forward_map', 'front_ovr_forward', 'front_ovr_forward_u8', 'dec_block_forward_map', 'back_forward_map', 'front_pack_l2_weights', 'front2_forward', 'front4_supported', 'front4_forward', 'front4_forward_u8', 'front4_forward_train', 'dec_block_forward', 'act_forward', 'act_backward',
                  'pixelnorm_forward', 'pixelnorm_backward', 'norm_forward', 'norm_backward', 'pool2x2_forward', 'pool2x2_backward', 'sub_forward', 'finish_pred')


_FORWARD = ('conv_forward', 'conv_backward_data', 'pack_conv_weights', 'repack_table', 'repack_weights', 'stem_forward', 'obs_mean_forward', 'head_forward', 'warp_forward',
            'resize_bilinear_forward', 'mul_forward')


def install(monkeypatch):
    """Replaces every Python-level C-ABI adapter of nlt_amd._capi by its CPU emulation above."""
    for name in _FORWARD + _TRAIN + _BUFFERS + _FUSED + _TILE + ('conv_forward_splitk', 'conv_backward_weights_tiled', 'conv_backward_weights_narrow'):
        monkeypatch.setattr(C, name, globals()[name])
