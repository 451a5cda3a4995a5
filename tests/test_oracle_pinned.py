"""Pins that round 1 left open, CPU side: the oracle's `compute_ngal` restatement and the
product's pair-count WRAPPER arithmetic against golden vectors the reference itself produced in the build container
(oracle/make_golden.py ngal / pairs: abacus_hod.py:861-1179 and tpcf_corrfunc.py:17-372 run under the shim, Corrfunc's
counters replaced by the oracle's brute-force counter)."""
import json
from types import SimpleNamespace

import numpy as np
import pytest
from conftest import load_golden


def ngal_ball(g):
    """object with the attributes compute_ngal reads: edges + the weighted histograms of AbacusHOD.__init__ (:200-251)"""
    nb = int(g['nbin'])
    ball = SimpleNamespace(z_mock=float(g['z']))
    ball.logMbins = np.linspace(np.log10(np.min(g['hmass'])), np.log10(np.max(g['hmass'])), nb + 1)
    ball.deltacbins = ball.fenvbins = ball.shearbins = np.linspace(-0.5, 0.5, nb + 1)
    cols = [np.log10(g['hmass']), g['hdeltac'], g['hfenv'], g['hshear']]
    ball.halo_mass_func, _ = np.histogramdd(np.vstack(cols[:3]).T, bins=[ball.logMbins, ball.deltacbins, ball.fenvbins],
                                            weights=g['hmultis'])
    ball.halo_mass_func_wshear, _ = np.histogramdd(np.vstack(cols).T, bins=[ball.logMbins, ball.deltacbins, ball.fenvbins,
                                                                            ball.shearbins], weights=g['hmultis'])
    return ball


def test_compute_ngal_oracle_vs_reference():
    from oracle import oracle
    g = load_golden('ngal')
    ball = ngal_ball(g)
    cases = json.loads(str(g['cases_json']))
    assert set(cases) == {'defaults', 'ab_zpivot', 'elg_evolving_conformity_defaults'}
    for name, tracers in cases.items():
        ngal, fsat = oracle.compute_ngal_numpy(ball, tracers)
        for t in tracers:
            np.testing.assert_allclose(ngal[t], float(g[f'{name}.{t}.ngal']), rtol=1e-12, err_msg=f'{name} {t}')
            np.testing.assert_allclose(fsat[t], float(g[f'{name}.{t}.fsat']), rtol=1e-12, err_msg=f'{name} {t}')


def brute_counters():
    """Corrfunc-convention DDrppi / DDsmu backed by the oracle's brute-force counter (what the golden was made with)"""
    from oracle import oracle as O

    def res(n):
        out = np.zeros(len(n), dtype=[('npairs', 'u8')])
        out['npairs'] = n
        return out

    def DDrppi(autocorr, nthreads, binfile=None, pimax=None, X1=None, Y1=None, Z1=None, X2=None, Y2=None, Z2=None,
               boxsize=None, **kw):
        s = (None, None, None) if autocorr else (X2, Y2, Z2)
        return res(O.paircount_brute('rppi', X1, Y1, Z1, float(boxsize), binfile, *s, pimax=float(pimax),
                                     npibins=int(pimax), nthread=4))

    def DDsmu(autocorr, nthreads, binfile=None, mu_max=None, nmu_bins=None, X1=None, Y1=None, Z1=None, X2=None, Y2=None,
              Z2=None, boxsize=None, **kw):
        s = (None, None, None) if autocorr else (X2, Y2, Z2)
        return res(O.paircount_brute('smu', X1, Y1, Z1, float(boxsize), binfile, *s, mu_max=float(mu_max),
                                     nmubins=int(nmu_bins), nthread=4))
    return DDrppi, DDsmu


def check_pair_wrappers(T):
    """calc_xirppi_fast / calc_wp_fast / calc_multipole_fast / tpcf_multipole of module T against the reference's
    outputs: bit for bit (same counts, same dtype and operation order in RR and xi)"""
    g = load_golden('pair_wrappers')
    a, b, L = g['a'], g['b'], float(g['L'])
    pimax, pbs, nmu = int(g['pimax']), int(g['pi_bin_size']), int(g['nbins_mu'])
    for tag, second in (('auto', {}), ('cross', dict(x2=b[:, 0], y2=b[:, 1], z2=b[:, 2]))):
        xi = T.calc_xirppi_fast(a[:, 0], a[:, 1], a[:, 2], g['rpbins'], pimax, pbs, L, 4, **second)
        wp = T.calc_wp_fast(a[:, 0], a[:, 1], a[:, 2], g['rpbins'], pimax, L, 4, **second)
        mp = T.calc_multipole_fast(a[:, 0], a[:, 1], a[:, 2], g['sbins'], L, 4, nbins_mu=nmu, orders=[0, 2, 4], **second)
        for got, key in ((xi, 'xirppi'), (wp, 'wp'), (mp, 'multipole')):
            want = g[f'{tag}.{key}']
            assert got.dtype == want.dtype and got.shape == want.shape, (tag, key)
            np.testing.assert_array_equal(got, want, err_msg=f'{tag}.{key}')
    for ell in (0, 1, 2, 4):
        np.testing.assert_array_equal(T.tpcf_multipole(g['tpcf.xi'], g['tpcf.mu_bins'], order=ell), g[f'tpcf.l{ell}'])


def test_pair_wrapper_arithmetic_vs_reference(monkeypatch):
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    DDrppi, DDsmu = brute_counters()
    monkeypatch.setattr(T, 'DDrppi', DDrppi)
    monkeypatch.setattr(T, 'DDsmu', DDsmu)
    check_pair_wrappers(T)


def test_pair_wrapper_argument_errors():
    """ValueError rules of tpcf_corrfunc.py:112-121,304-305 (raised before any counting)"""
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    x = np.zeros(4)
    bins = np.linspace(0.1, 1, 4)
    with pytest.raises(ValueError, match='pimax needs to be an integer'):
        T.calc_xirppi_fast(x, x, x, bins, 30.0, 5, 100.0, 1)
    with pytest.raises(ValueError, match='pi_bin_size needs to be an integer'):
        T.calc_xirppi_fast(x, x, x, bins, 30, 5.0, 100.0, 1)
    with pytest.raises(ValueError, match='integer divisor'):
        T.calc_xirppi_fast(x, x, x, bins, 30, 7, 100.0, 1)
    with pytest.raises(ValueError, match='pimax needs to be an integer'):
        T.calc_wp_fast(x, x, x, bins, 30.0, 100.0, 1)


def test_cell_list_counter_equals_brute_force():
    """oracle.paircount_cells (the CPU baseline of bench.py's pair leg) against the brute-force counter, the checker of the
    HIP kernels: identical integers in all three modes, auto and cross, coordinates in [0, L) and in [-L/2, L/2)"""
    from oracle import oracle
    rng = np.random.default_rng(1)
    box = 200.0
    bins = np.logspace(-1, np.log10(30), 14)
    for centered in (False, True):
        p = rng.random((4000, 3)) * box - (box / 2 if centered else 0)
        q = rng.random((3000, 3)) * box - (box / 2 if centered else 0)
        for mode, kw in (('r', {}), ('rppi', dict(pimax=30.0, npibins=30)), ('smu', dict(mu_max=1.0, nmubins=20))):
            a = oracle.paircount_brute(mode, p[:, 0], p[:, 1], p[:, 2], box, bins, nthread=4, **kw)
            b = oracle.paircount_cells(mode, p[:, 0], p[:, 1], p[:, 2], box, bins, nthread=4, **kw)
            np.testing.assert_array_equal(a, b)
            a = oracle.paircount_brute(mode, p[:, 0], p[:, 1], p[:, 2], box, bins, q[:, 0], q[:, 1], q[:, 2], nthread=4, **kw)
            b = oracle.paircount_cells(mode, p[:, 0], p[:, 1], p[:, 2], box, bins, q[:, 0], q[:, 1], q[:, 2], nthread=4, **kw)
            np.testing.assert_array_equal(a, b)
