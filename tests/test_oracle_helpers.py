"""oracle restatements of the ZCV-facing spectrum helpers (analysis/power_spectrum.py:303-660) against the outputs of the
shimmed reference (tests/golden/power_helpers.npz, oracle/make_golden.py helpers)"""
import numpy as np
import pytest
from conftest import load_golden

from oracle import oracle


@pytest.fixture(scope='module')
def g():
    return load_golden('power_helpers')


@pytest.mark.parametrize('n', [16, 21])
def test_helpers_against_reference(g, n):
    L = float(g['meta.L'])
    p3d, ke, xi, re = g[f'n{n}.p3d'], g[f'n{n}.kedges'], g[f'n{n}.xi'], g[f'n{n}.redges']
    m, c = oracle.bin_kppi(n, L, ke, np.pi * n / L * 1.01, 5, p3d)
    np.testing.assert_array_equal(c, g[f'n{n}.kppi.counts'])
    np.testing.assert_allclose(m, g[f'n{n}.kppi.mean'], rtol=2e-5)
    m, c = oracle.bin_kppi(n, L, re, L / 2 * 1.01, 4, xi, fourier=False)
    np.testing.assert_array_equal(c, g[f'n{n}.rppi.counts'])
    np.testing.assert_allclose(m, g[f'n{n}.rppi.mean'], rtol=2e-4, atol=2e-5)
    bp, npo = oracle.project_3d_to_poles(ke, p3d, L, [0, 2, 4])
    np.testing.assert_array_equal(npo, g[f'n{n}.p2poles.N'])
    np.testing.assert_allclose(bp, g[f'n{n}.p2poles.poles'], rtol=2e-5, atol=1e-5 * np.abs(g[f'n{n}.p2poles.poles']).max())
    rb, xp, nr = oracle.pk_to_xi(p3d.copy(), L, re, poles=[0, 2, 4])
    np.testing.assert_allclose(rb, g[f'n{n}.pk2xi.r'])
    np.testing.assert_array_equal(nr, g[f'n{n}.pk2xi.N'])
    np.testing.assert_allclose(xp, g[f'n{n}.pk2xi.poles'], rtol=2e-4, atol=2e-5 * np.abs(g[f'n{n}.pk2xi.poles']).max())
    np.testing.assert_allclose(oracle.expand_poles_to_3d(g[f'n{n}.expand.k_ell'], g[f'n{n}.expand.P_ell'], n, L, [0, 2, 4]),
                               g[f'n{n}.expand.Pk'], rtol=2e-5, atol=2e-4 * np.abs(g[f'n{n}.expand.Pk']).max())
    np.testing.assert_allclose(oracle.get_smoothing(n, L, 7.5), g[f'n{n}.smoothing'], rtol=2e-6)
    np.testing.assert_allclose(oracle.get_delta_mu2(g[f'n{n}.delta'], n), g[f'n{n}.delta_mu2'], rtol=2e-6, atol=1e-7)
