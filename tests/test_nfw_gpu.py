"""NFW satellites (`gen_gal_cat(nfw=True)`, reference gen_sats_nfw hod/GRAND_HOD.py:417-822).  The reference draws
from unseeded per-thread generators, so parity is STATISTICAL: satellite numbers against the Poisson means of the
oracle's restatement, radii against the NFW_draw table, directions, velocities; the deterministic parts (centrals,
host mass / id, the RSD relation) are checked exactly."""
import warnings

import numpy as np
import pytest
from scipy import stats

from abacusutils_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


def nfw_table(n=200000, cmax=12.0, seed=3):
    """draws of r/r_s from an NFW mass profile truncated at cmax (what abacusutils ships as NFW_draw)"""
    rng = np.random.default_rng(seed)
    x = np.linspace(0, cmax, 20001)
    mcum = np.log1p(x) - x / (1 + x)
    return np.interp(rng.random(n) * mcum[-1], mcum, x)


@pytest.fixture(scope='module')
def setup():
    from abacusutils_amd.hod import GRAND_HOD as G
    hd, pd, params = synth.synth_hod_inputs(400000, 1000, seed=21, lbox=1000.0)
    rng = np.random.default_rng(4)
    hd['hc'] = rng.uniform(3.0, 9.0, len(hd['hmass']))
    hd['hrvir'] = 0.3 * (hd['hmass'] / 1e13) ** (1 / 3)
    tracers = {'LRG': dict(synth.LRG_PARAMS, f_sigv=0.8), 'ELG': dict(synth.ELG_PARAMS, f_sigv=1.1),
               'QSO': dict(synth.QSO_PARAMS, f_sigv=0.5)}
    return G, hd, pd, params, tracers, nfw_table()


def run(G, hd, pd, params, tracers, draw, rsd, seed):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return G.gen_gal_cat(hd, pd, tracers, params, rsd=rsd, nfw=seed, NFW_draw=draw)


def test_nfw_statistics(setup):
    G, hd, pd, params, tracers, draw = setup
    cat = run(G, hd, pd, params, tracers, draw, False, 12345)
    plain = G.gen_gal_cat(hd, pd, tracers, params, rsd=False)           # particle path: same centrals
    st = G.StagedCatalog(hd, pd)
    st.populate(G.marshal_params(tracers, params, False, False))
    keep_cent, _ = st.fetch_keep()
    st.free()
    lam = oracle.nfw_expected_counts(hd, tracers, params, keep_cent)
    order = np.argsort(hd['hid'])
    for tr in tracers:
        c = cat[tr]
        nc = c['Ncent']
        assert nc == plain[tr]['Ncent']
        for k in ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass', 'id'):
            np.testing.assert_array_equal(c[k][:nc], plain[tr][k][:nc])   # centrals: the deterministic path
        nsat = len(c['x']) - nc
        tot = lam[tr].sum()
        assert tot > 500
        assert abs(nsat - tot) < 5 * np.sqrt(tot), (tr, nsat, tot)         # Poisson total
        h = order[np.searchsorted(hd['hid'][order], c['id'][nc:])]          # host halo of every satellite
        np.testing.assert_array_equal(hd['hid'][h], c['id'][nc:])
        np.testing.assert_array_equal(hd['hmass'][h], c['mass'][nc:])
        # per-halo numbers: mean and variance of (N - lambda) / sqrt(lambda) over halos with lambda > 0.05
        n_per = np.bincount(h, minlength=len(hd['hmass']))
        sel = lam[tr] > 0.05
        pull = (n_per[sel] - lam[tr][sel]) / np.sqrt(lam[tr][sel])
        assert abs(pull.mean()) < 5 / np.sqrt(sel.sum()) and abs(pull.var() - 1) < 0.1
        assert n_per[lam[tr] == 0].sum() == 0
        # radii: eta * c = NFW_draw[k] with NFW_draw[k] <= c
        d = np.stack([c[k][nc:] for k in 'xyz'], 1) - hd['hpos'][h]
        r = np.linalg.norm(d, axis=1)
        tval = r / hd['hrvir'][h] * hd['hc'][h]
        assert np.all(tval <= hd['hc'][h] * (1 + 1e-12))
        cbin = (hd['hc'][h] > 5.9) & (hd['hc'][h] < 6.1)                    # one concentration: plain truncated table
        if cbin.sum() > 300:
            assert stats.ks_2samp(tval[cbin], draw[draw <= 6.0]).pvalue > 1e-4
        u = d / r[:, None]                                                  # isotropy
        assert np.abs(u.mean(0)).max() < 5 / np.sqrt(3 * nsat)
        assert stats.kstest(u[:, 2], 'uniform', args=(-1, 2)).pvalue > 1e-4
        sig = hd['hsigma3d'][h] * 0.577 * tracers[tr]['f_sigv']             # velocities (:511-516)
        zv = (np.stack([c[k][nc:] for k in ('vx', 'vy', 'vz')], 1) - hd['hvel'][h]) / sig[:, None]
        assert stats.kstest(zv.ravel(), 'norm').pvalue > 1e-4
        assert np.abs(np.corrcoef(zv.T) - np.eye(3)).max() < 0.05


def test_nfw_determinism_rsd_and_errors(setup):
    G, hd, pd, params, tracers, draw = setup
    a = run(G, hd, pd, params, tracers, draw, False, 777)
    b = run(G, hd, pd, params, tracers, draw, False, 777)
    c = run(G, hd, pd, params, tracers, draw, True, 777)
    d = run(G, hd, pd, params, tracers, draw, False, 778)
    L = params['Lbox']
    for tr in tracers:
        for k in a[tr]:
            np.testing.assert_array_equal(a[tr][k], b[tr][k])
        nc = a[tr]['Ncent']
        assert len(d[tr]['x']) != len(a[tr]['x']) or not np.array_equal(d[tr]['x'], a[tr]['x'])
        np.testing.assert_array_equal(c[tr]['x'][nc:], a[tr]['x'][nc:])
        zr = a[tr]['z'][nc:] + a[tr]['vz'][nc:] / params['velz2kms']
        np.testing.assert_allclose(c[tr]['z'][nc:], zr - np.floor(zr / L) * L, rtol=0, atol=1e-9)   # Python modulo (:787-789)
        assert c[tr]['z'][nc:].min() >= 0 and c[tr]['z'][nc:].max() < L
    np.random.seed(5)
    e = run(G, hd, pd, params, tracers, draw, False, True)                 # nfw=True: key from NumPy's global generator
    np.random.seed(5)
    f = run(G, hd, pd, params, tracers, draw, False, True)
    np.testing.assert_array_equal(e['LRG']['x'], f['LRG']['x'])
    with pytest.raises(ValueError):
        G.gen_gal_cat(hd, pd, tracers, params, nfw=True)                   # no NFW_draw
    hd2 = {k: v for k, v in hd.items() if k != 'hrvir'}
    with pytest.raises(KeyError):
        G.gen_gal_cat(hd2, pd, tracers, params, nfw=True, NFW_draw=draw)
    lc = dict(params, origin=np.array([-990.0, -990.0, -990.0]))
    from abacusutils_amd._lib import AbacusHipError
    with pytest.raises(AbacusHipError):
        G.gen_gal_cat(hd, pd, tracers, lc, nfw=True, NFW_draw=draw)        # no light cones on this path (:551)
