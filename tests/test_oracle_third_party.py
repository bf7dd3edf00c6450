"""CPU: the oracle's restatements of the third-party semantics (TF 2.2 / TF-Addons 0.10, neither installable here) checked
against implementations that were NOT written from the same notes:

  * tfa.image.resampler (oracle/tf_ops.resampler*, nlt/models/nlt.py:112-114)  vs  torch.nn.functional.grid_sample(bilinear,
    padding_mode='zeros', align_corners=True) -- pixel-unit coordinates, implicit one-texel zero border, zero outside;
  * tf.image.resize (TF2 default; nlt/util/img.py:115) restated as a NumPy loop  vs  F.interpolate(bilinear, align_corners=False);
  * Conv2D / Conv2DTranspose 'same' (nlt/networks/elements.py:26-39) vs scipy.signal.correlate2d / convolve2d on the padded image
    (the pad side -- bottom/right for k2s1, top/left for its transpose -- is TF's documented rule pad_before = pad_total // 2);
  * Keras Adam(amsgrad=True): for a CONSTANT gradient g the update has the closed form theta_t = theta_0 - sum_i lr_i sign(g)
    |g| / (|g| sqrt(1 - b2^i)... ) -- evaluated independently in float64 from the OptimizerV2 formulas.

And, when tests/golden/tf_ops.npz exists (made by tests/golden/make_tf_golden.py on a machine with tensorflow==2.2.0 and
tensorflow-addons==0.10.0), the same oracle functions against TensorFlow's own outputs -- skipped otherwise."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nlt_oracle as O
from oracle import tf_ops as T

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'tf_ops.npz')


def _grid_sample(data, warp_xy):
    """[N,H,W,C] sampled at pixel-unit (x, y) through torch's own bilinear sampler."""
    n, h, w, c = data.shape
    gx = warp_xy[..., 0] * (2.0 / (w - 1)) - 1.0              # align_corners=True: -1 -> pixel 0, +1 -> pixel w - 1
    gy = warp_xy[..., 1] * (2.0 / (h - 1)) - 1.0
    out = F.grid_sample(data.permute(0, 3, 1, 2), torch.stack((gx, gy), -1), mode='bilinear', padding_mode='zeros',
                        align_corners=True)
    return out.permute(0, 2, 3, 1)


@pytest.mark.parametrize('h,w', [(9, 13), (16, 16), (5, 31)])
def test_resampler_equals_torch_grid_sample_including_the_borders(h, w):
    rng = np.random.default_rng(h * w)
    data = torch.tensor(rng.standard_normal((2, h, w, 3)), dtype=torch.float64)
    # interior points, points within one texel of every edge (partial taps), points beyond it (zero), exact integers
    xy = rng.uniform(-2.0, 1.0, (2, 40, 17, 2)) * [1.0, 1.0] + rng.uniform(0, 1, (2, 40, 17, 2)) * [w + 1.0, h + 1.0]
    xy[0, :4, :4] = np.round(xy[0, :4, :4])
    xy[0, 5, :, 0] = -0.5; xy[0, 6, :, 0] = w - 0.5; xy[0, 7, :, 1] = -0.25; xy[0, 8, :, 1] = h - 0.75
    xy[1, 0, 0] = (0.0, 0.0); xy[1, 0, 1] = (w - 1.0, h - 1.0)
    warp = torch.tensor(xy, dtype=torch.float64)
    ours = T.resampler(data, warp)
    ref = _grid_sample(data, warp)
    # the samplers agree wherever TFA's `inside` predicate holds and a point does not sit within rounding of -1 / w / h
    np.testing.assert_allclose(ours.numpy(), ref.numpy(), atol=1e-12)
    naive = T.resampler_naive(data.float().numpy(), warp.float().numpy())
    np.testing.assert_allclose(naive, _grid_sample(data.float(), warp.float()).numpy(), atol=2e-5)


def test_resampler_gradient_equals_grid_samples():
    rng = np.random.default_rng(4)
    data = torch.tensor(rng.standard_normal((1, 7, 9, 3)), dtype=torch.float64, requires_grad=True)
    warp = torch.tensor(rng.uniform(-1.5, 9.5, (1, 11, 6, 2)), dtype=torch.float64)
    g = torch.tensor(rng.standard_normal((1, 11, 6, 3)), dtype=torch.float64)
    (ga,) = torch.autograd.grad((T.resampler(data, warp) * g).sum(), data)
    (gb,) = torch.autograd.grad((_grid_sample(data, warp) * g).sum(), data)
    np.testing.assert_allclose(ga.numpy(), gb.numpy(), atol=1e-12)


@pytest.mark.parametrize('h,w,oh,ow', [(8, 8, 16, 16), (16, 12, 5, 7), (6, 10, 6, 25), (64, 4, 7, 4)])
def test_resize_naive_equals_torch_interpolate(h, w, oh, ow):
    x = np.random.default_rng(oh).standard_normal((2, h, w, 3)).astype(np.float32)
    ref = F.interpolate(torch.tensor(x).permute(0, 3, 1, 2), size=(oh, ow), mode='bilinear', align_corners=False, antialias=False)
    np.testing.assert_allclose(T.resize_bilinear_naive(x, oh, ow), ref.permute(0, 2, 3, 1).numpy(), atol=1e-5)


@pytest.mark.parametrize('k,s', [(1, 1), (2, 1), (2, 2)])
def test_conv_same_equals_scipy_correlate_on_the_tf_padded_image(k, s):
    from scipy.signal import correlate2d
    rng = np.random.default_rng(k + s)
    h, w, cin, cout = 8, 10, 3, 2
    x = rng.standard_normal((1, h, w, cin)); wk = rng.standard_normal((k, k, cin, cout)); b = rng.standard_normal(cout)
    oh, ow = -(-h // s), -(-w // s)
    ph, pw = max((oh - 1) * s + k - h, 0), max((ow - 1) * s + k - w, 0)            # TF 'SAME': pad_before = total // 2, rest after
    xp = np.pad(x[0], ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
    ref = np.stack([sum(correlate2d(xp[..., c], wk[..., c, o], mode='valid') for c in range(cin))[::s, ::s] + b[o]
                    for o in range(cout)], -1)
    got = T.conv2d_same(torch.tensor(x), torch.tensor(wk), torch.tensor(b), s)[0].numpy()
    np.testing.assert_allclose(got, ref, atol=1e-12)


@pytest.mark.parametrize('s', [1, 2])
def test_conv_transpose_same_equals_scipy_full_convolution_cropped(s):
    """Conv2DTranspose = the gradient of the forward conv: a FULL convolution of the zero-stuffed input with the taps, cropped
    by the forward conv's padding (k2s1: pad (0,1) -> keep rows / cols [0, h); k2s2: no padding)."""
    from scipy.signal import convolve2d
    rng = np.random.default_rng(10 + s)
    h, w, cin, cout = 6, 7, 3, 2
    x = rng.standard_normal((1, h, w, cin)); wk = rng.standard_normal((2, 2, cout, cin)); b = rng.standard_normal(cout)
    up = np.zeros((h * s - (s - 1), w * s - (s - 1), cin)); up[::s, ::s] = x[0]
    ref = np.stack([sum(convolve2d(up[..., c], wk[:, :, o, c], mode='full') for c in range(cin))[:h * s, :w * s] + b[o]
                    for o in range(cout)], -1)
    got = T.conv2d_transpose_same(torch.tensor(x), torch.tensor(wk), torch.tensor(b), s)[0].numpy()
    np.testing.assert_allclose(got, ref, atol=1e-12)


def test_keras_adam_amsgrad_constant_gradient_closed_form():
    """TF 2.2 OptimizerV2 Adam(amsgrad=True), constant gradient g: m_t = g (1 - b1^t), v_t = g^2 (1 - b2^t) = vhat_t (monotone),
    theta_t = theta_{t-1} - lr sqrt(1 - b2^t) / (1 - b1^t) * m_t / (sqrt(vhat_t) + eps) -- epsilon OUTSIDE the bias correction."""
    g, lr, b1, b2, eps = 0.37, 1e-3, 0.9, 0.999, 1e-7
    p = torch.tensor([1.0, -2.0], dtype=torch.float64)
    opt = O.KerasAdamAMSGrad([p], lr)
    theta = np.array([1.0, -2.0])
    for t in range(1, 6):
        opt.step([torch.tensor([g, -g], dtype=torch.float64)])
        m, v = g * (1 - b1 ** t), g * g * (1 - b2 ** t)
        step = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (math.sqrt(v) + eps)
        theta = theta - np.array([step, -step])
        np.testing.assert_allclose(p.numpy(), theta, rtol=1e-12)
    # torch's own Adam puts epsilon elsewhere (sqrt(vhat) / sqrt(1 - b2^t) + eps): with eps comparable to sqrt(v) the two differ
    q = torch.tensor([1.0], dtype=torch.float64, requires_grad=True)
    topt = torch.optim.Adam([q], lr=lr, betas=(b1, b2), eps=1e-3, amsgrad=True)
    r = torch.tensor([1.0], dtype=torch.float64)
    kopt = O.KerasAdamAMSGrad([r], lr, eps=1e-3)
    for _ in range(3):
        q.grad = torch.tensor([1e-3], dtype=torch.float64); topt.step()
        kopt.step([torch.tensor([1e-3], dtype=torch.float64)])
    assert abs(float(q.detach()) - float(r)) > 1e-5


# ------------------------------------------------------------------ TensorFlow's own outputs, when somebody has made them
needs_golden = pytest.mark.skipif(not os.path.exists(GOLDEN), reason="tests/golden/tf_ops.npz absent (needs tensorflow==2.2.0 + "
                                  "tensorflow-addons==0.10.0: tests/golden/make_tf_golden.py)")


@needs_golden
def test_oracle_ops_against_tensorflow_golden():
    g = np.load(GOLDEN)
    t = lambda a: torch.tensor(np.asarray(a))
    for name in [k[:-2] for k in g.files if k.startswith('conv_') and k.endswith('_y')]:
        k_, s_ = int(name.split('_')[1][1]), int(name.split('_')[2][1])
        tr = name.startswith('conv_t')
        x, w, b, y = g[name + '_x'], g[name + '_w'], g[name + '_b'], g[name + '_y']
        f = T.conv2d_transpose_same if tr else T.conv2d_same
        np.testing.assert_allclose(f(t(x), t(w), t(b), s_).numpy(), y, atol=2e-5 * np.abs(y).max(), err_msg=name)
    np.testing.assert_array_equal(T.leaky_relu(t(g['lrelu_x']), 0.3).numpy(), g['lrelu_y'])
    np.testing.assert_allclose(T.resampler_naive(g['resampler_data'], g['resampler_warp']), g['resampler_out'], atol=1e-6)
    np.testing.assert_allclose(T.resize_bilinear_naive(g['resize_x'], *g['resize_y'].shape[1:3]), g['resize_y'], atol=1e-6)
    p = [t(g['adam_p0']).clone()]
    opt = O.KerasAdamAMSGrad(p, float(g['adam_lr']))
    for i in range(3):
        opt.step([t(g['adam_g%d' % i])])
        np.testing.assert_allclose(p[0].numpy(), g['adam_p%d' % (i + 1)], rtol=2e-6, atol=1e-9)
    # clipnorm through apply_gradients (the reference's loop, nlt/trainvali.py:279-280): whatever TF 2.2 did is recorded
    clipped = O.clip_by_norm(t(g['clip_g']), float(g['clip_norm'])).numpy()
    applied_like_clipped = np.allclose(g['clip_p1_applied'], g['clip_p1_if_clipped'], rtol=1e-6, atol=1e-9)
    assert applied_like_clipped == bool(g['clip_apply_gradients_clips']), "DESIGN.md section 8 states what TF 2.2 does here"
    np.testing.assert_allclose(clipped, g['clip_by_norm_out'], rtol=2e-6)
