"""CPU: the gather form of the wavelet pass's adjoint (csrc/barron.hip: dwt_T_gather) against the scatter form it replaced
(dwt_axis_T_kernel: every output coefficient adds tap * gradient at reflect(position)).  For every axis length the launcher sends
to the gather form (n >= 6) each input element must collect exactly the (tap, coefficient) pairs the scatter form sends it."""
import numpy as np
import pytest

KLO = np.array([+0.037828455507, -0.023849465020, -0.110624404418, +0.377402855613, +0.852698679009,
                +0.377402855613, -0.110624404418, -0.023849465020, +0.037828455507])
KHI = np.array([+0.064538882629, -0.040689417609, -0.418092273222, +0.788485616406,
                -0.418092273222, -0.040689417609, +0.064538882629])


def reflect(j, n):                                                  # barron.hip: reflect()
    period = max(1, 2 * (n - 1))
    jm = j % period
    return min(2 * (n - 1) - jm, jm)


def n_lo(n): return (n - 1) // 2 + 1
def n_hi(n): return (n - 2) // 2 + 1 if n >= 2 else 0


def scatter(glo, ghi, n):
    x = np.zeros(n)
    for i in range(n_lo(n)):
        for t in range(9):
            x[reflect(2 * i + t - 4, n)] += KLO[t] * glo[i]
    for i in range(n_hi(n)):
        for t in range(7):
            x[reflect(2 * i + 1 + t - 3, n)] += KHI[t] * ghi[i]
    return x


def gather(glo, ghi, i, n):                                         # barron.hip: dwt_T_gather()
    nl, nh = n_lo(n), n_hi(n)
    s = 0.0
    for v in range(3):
        if (v == 1 and i == 0) or (v == 2 and i == n - 1):
            continue
        p = i if v == 0 else (-i if v == 1 else 2 * (n - 1) - i)
        if p < -4 or p > n + 3:
            continue
        j0, j1 = max((p - 4 + 1) >> 1, 0), min((p + 4) >> 1, nl - 1)
        for j in range(j0, j1 + 1):
            s += KLO[p - 2 * j + 4] * glo[j]
        j0, j1 = max((p - 4 + 1) >> 1, 0), min((p + 2) >> 1, nh - 1)
        for j in range(j0, j1 + 1):
            s += KHI[p - 2 * j + 2] * ghi[j]
    return s


@pytest.mark.parametrize('n', list(range(6, 40)) + [63, 64, 65, 127, 128, 200, 511, 512])
def test_gather_form_collects_what_the_scatter_form_sends(n):
    rng = np.random.default_rng(n)
    glo, ghi = rng.standard_normal(n_lo(n)), rng.standard_normal(n_hi(n))
    ref = scatter(glo, ghi, n)
    got = np.array([gather(glo, ghi, i, n) for i in range(n)])
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    # and by one-hot gradients: exactly the same (tap, coefficient) pairs, not just the same sums
    for j in (0, 1, n_lo(n) - 1):
        e = np.zeros(n_lo(n)); e[j] = 1.0
        z = np.zeros(n_hi(n))
        np.testing.assert_allclose([gather(e, z, i, n) for i in range(n)], scatter(e, z, n), atol=1e-15)
    for j in (0, n_hi(n) - 1):
        e = np.zeros(n_hi(n)); e[j] = 1.0
        z = np.zeros(n_lo(n))
        np.testing.assert_allclose([gather(z, e, i, n) for i in range(n)], scatter(z, e, n), atol=1e-15)
