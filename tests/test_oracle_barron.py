"""Pins the oracle's Barron-loss pieces to the reference's own golden data
(third_party/robust_loss/*_test.py).  CPU only."""
import os
import numpy as np
import pytest
import torch
import scipy.special

from oracle import barron as B

G = os.path.join(os.path.dirname(__file__), 'golden')


def test_pad_reflecting_golden1():      # wavelet_test.py:89-104
    n, p0, p1 = 8, 17, 13
    ref = np.concatenate((np.arange(3, 0, -1), np.arange(n), np.arange(n - 2, 0, -1), np.arange(n),
                          np.arange(n - 2, 0, -1), np.arange(7)))
    np.testing.assert_array_equal(B.pad_reflecting(np.arange(n), p0, p1, 0), ref)


def test_pad_reflecting_golden2():      # wavelet_test.py:106-119
    n, p0, p1 = 11, 15, 7
    ref = np.concatenate((np.arange(5, n), np.arange(n - 2, 0, -1), np.arange(n), np.arange(n - 2, 2, -1)))
    np.testing.assert_array_equal(B.pad_reflecting(np.arange(n), p0, p1, 0), ref)


def test_pad_one_reflection_matches_numpy():    # wavelet_test.py:47-68 (tf.pad REFLECT == np.pad reflect)
    rng = np.random.default_rng(0)
    for _ in range(8):
        n = int(rng.integers(2, 10))
        x = rng.uniform(size=(n, n, n))
        pb, pa = int(rng.integers(0, n)), int(rng.integers(0, n))
        ax = int(rng.integers(0, 3))
        pads = [(0, 0)] * 3
        pads[ax] = (pb, pa)
        np.testing.assert_array_equal(B.pad_reflecting(x, pb, pa, ax), np.pad(x, pads, mode='reflect'))


def test_analysis_lowpass_normalised():  # wavelet_test.py:121-128
    assert abs(np.sum(B.ANALYSIS_LO[:, None] * B.ANALYSIS_LO[None, :]) - 2.) < 1e-10
    assert len(B.ANALYSIS_LO) == 9 and len(B.ANALYSIS_HI) == 7


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_construct_matches_golden(dtype):       # wavelet_test.py:167-171 (atol 1e-5)
    d = np.load(os.path.join(G, 'wavelet_golden.npz'))
    im = torch.tensor(np.float32(d['I_color'])).to(dtype)
    pyr = B.construct(im, 5)
    for lvl in range(5):
        for b in range(3):
            ref = d['band_%d_%d' % (lvl, b)]
            got = pyr[lvl][b].numpy()
            assert got.shape == ref.shape
            np.testing.assert_allclose(got, ref, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(pyr[-1].numpy(), d['resid'], atol=1e-5, rtol=1e-5)


def test_flatten_shape_and_volume():    # wavelet_test.py:130-144 (|det J| = 1 for power-of-two sizes)
    im = torch.rand(1, 4, 4, dtype=torch.float64)
    fun = lambda z: B.flatten(B.construct(z, 2)).reshape(-1)
    J = torch.autograd.functional.jacobian(fun, im).reshape(16, 16)
    assert abs(abs(np.linalg.det(J.numpy())) - 1.) < 1e-5
    d = np.load(os.path.join(G, 'wavelet_golden.npz'))
    flat = B.flatten(B.construct(torch.tensor(d['I_color']), 5))
    assert tuple(flat.shape) == d['I_color'].shape


def test_partition_spline_known_values():       # distribution_test.py:86-106
    s = np.load(os.path.join(G, 'partition_spline.npz'))
    lz = B.log_base_partition_function
    assert abs(lz(np.inf, s) - 0.70526025442) < 1e-7
    assert abs(lz(0., s) - np.log(np.pi * np.sqrt(2))) < 1e-7        # distribution.py:72-73
    assert abs(lz(2., s) - np.log(np.sqrt(2 * np.pi))) < 1e-7        # distribution.py:74-75
    # alpha = 1: Z = 2 e K_1(1) (Charbonnier normaliser); the constant the product kernels use
    z1 = np.log(2 * np.e * scipy.special.k1(1.0))
    assert abs(lz(1., s) - z1) < 1e-6
    assert abs(B.LOG_Z_ALPHA1 - lz(1., s)) < 5e-9


def test_lossfun_closed_forms():        # general_test.py:245-257 and neighbours
    x = np.arange(-20, 20, 0.1)
    np.testing.assert_allclose(B.lossfun(x, 1., 1.7), np.sqrt((x / 1.7) ** 2 + 1) - 1, rtol=1e-12)
    np.testing.assert_allclose(B.lossfun(x, 2., 1.7), 0.5 * (x / 1.7) ** 2, rtol=1e-12)
    np.testing.assert_allclose(B.lossfun(x, 0., 1.7), np.log(0.5 * (x / 1.7) ** 2 + 1), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(B.lossfun(x, -2., 1.7), 2 * (x / 1.7) ** 2 / ((x / 1.7) ** 2 + 4), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(B.lossfun(x, -np.inf, 1.7), 1 - np.exp(-0.5 * (x / 1.7) ** 2), rtol=1e-9, atol=1e-15)
    # the fixed-alpha form the oracle/product use == general form at alpha=1, c=0.01
    w = np.linspace(-1, 1, 101)
    got = B.charbonnier_nll(torch.tensor(w), log_z=0.0).numpy() - np.log(0.01)
    np.testing.assert_allclose(got, B.lossfun(w, 1., 0.01), rtol=1e-12, atol=1e-12)


def test_syuv_volume_preserving():      # util_test.py:132-139
    m = B.RGB_TO_YUV * B.VOLUME_PRESERVING_YUV_SCALE
    assert abs(abs(np.linalg.det(m)) - 1.) < 1e-5


def test_barron_loss_runs_and_is_per_example():
    torch.manual_seed(0)
    gt, pred = torch.rand(2, 64, 64, 3), torch.rand(2, 64, 64, 3)
    per = B.barron_loss(gt, pred, keep_batch=True)
    assert per.shape == (2,)
    assert abs(per.mean().item() - B.barron_loss(gt, pred).item()) < 1e-5
    # zero residual -> every coefficient 0 -> rho=0 -> loss = log(c)+logZ(1)
    z = B.barron_loss(gt, gt).item()
    assert abs(z - (np.log(0.01) + B.LOG_Z_ALPHA1)) < 1e-6


def test_log_partition_fractions_match_numerical_integration():    # distribution_test.py:95-106, 156-166
    """The reference checks the spline against a Meijer-G closed form (mpmath) at alpha = n/11, n = 0..22; here the same
    grid is checked against the DEFINITION  Z(alpha) = integral exp(-rho(x, alpha, 1)) dx  by quadrature -- i.e. the
    restated spline interpolation + the shipped partition_spline.npz give a pdf that integrates to one."""
    import scipy.integrate
    s = np.load(os.path.join(G, 'partition_spline.npz'))
    for n in range(0, 23):
        alpha = n / 11.0
        z, _ = scipy.integrate.quad(lambda x: np.exp(-float(B.lossfun(x, alpha, 1.0))), -np.inf, np.inf, epsabs=1e-12, epsrel=1e-12,
                                    limit=400)
        assert abs(float(B.log_base_partition_function(alpha, s)) - np.log(z)) < 2e-6, (alpha, np.log(z))


def test_lossfun_general_properties():          # general_test.py:104-131, 184-195
    rng = np.random.default_rng(0)
    x = rng.normal(size=4000) * 4
    alpha = rng.uniform(-4, 6, size=4000)
    scale = np.exp(rng.normal(size=4000))
    loss = B.lossfun(x, alpha, scale)
    assert np.all(np.isfinite(loss)) and np.all(loss >= 0)
    small = B.lossfun(rng.normal(size=4000) * 1e-6, alpha, scale)
    assert np.all(np.abs(small) < 1e-5)                                             # near zero at the origin
    mask = np.abs(x) < 0.5 * scale
    np.testing.assert_allclose(loss[mask], 0.5 * (x[mask] / scale[mask]) ** 2, rtol=1e-5, atol=1e-2)   # quadratic bowl near 0
    mult = np.maximum(0.2, np.exp(rng.normal(size=4000)))
    np.testing.assert_allclose(B.lossfun(mult * x, alpha, mult * scale), loss, rtol=1e-9, atol=1e-12)  # scale invariance
    # monotone in |x|: the derivative has the sign of x
    d = (B.lossfun(x + 1e-6, alpha, scale) - B.lossfun(x - 1e-6, alpha, scale)) / 2e-6
    assert np.all(d[np.abs(x) > 1e-3] * np.sign(x[np.abs(x) > 1e-3]) > 0)
