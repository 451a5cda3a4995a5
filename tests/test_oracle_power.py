"""CPU oracle power-spectrum chain vs golden vectors captured from the reference's calc_power /
calc_pk_from_deltak / get_W_compensated (oracle/make_golden.py).  No GPU needed.

The reference's own test inputs (tests/data_power/test_pos.npz) are a missing blob, so the nbodykit
goldens there are unusable; its invariants (tests/test_power.py:58-61) are kept.
"""
import numpy as np
import pytest
from conftest import assert_spectrum_close, load_golden

from abacusutils_amd import synth
from oracle import oracle

L, N, NMESH = 500.0, 20000, 32


def _pos():
    return synth.synth_positions(N, L, seed=300, clustered=True)


RTOL = 1e-5   # the north_star tolerance, against what the REFERENCE returned (measured: power <= 9.1e-7, k_avg <= 1.8e-6,
              # multipoles <= 5e-7 of the largest value in all 16 mode x accumulator combinations and the five extra cases)


def _check(tab, g, name, rtol=RTOL):
    for k in ('power', 'k_avg', 'poles'):
        if f'{name}.{k}' in g:
            assert_spectrum_close(tab[k], g[f'{name}.{k}'], rtol=rtol, err_msg=f'{name}.{k}')
    for k in ('N_mode', 'N_mode_poles'):
        if f'{name}.{k}' in g:
            np.testing.assert_array_equal(np.asarray(tab[k]), g[f'{name}.{k}'], err_msg=f'{name}.{k}')
    np.testing.assert_allclose(tab['k_mid'], g[f'{name}.k_mid'], rtol=1e-15)


@pytest.mark.parametrize('paste', ['TSC', 'CIC'])
@pytest.mark.parametrize('comp', [False, True])
@pytest.mark.parametrize('inter', [False, True])
@pytest.mark.parametrize('accum64', [False, True])
def test_calc_power_modes(paste, comp, inter, accum64):
    g = load_golden('power_cases')
    name = f'{paste}_c{int(comp)}_i{int(inter)}'
    tab = oracle.calc_power(_pos(), L, kbins=12, mubins=4, k_max=np.pi * NMESH / L + 1e-6, paste=paste, nmesh=NMESH,
                            compensated=comp, interlaced=inter, poles=[0, 2, 4], nthread=1 if not accum64 else 4,
                            accum64=accum64)
    # float32-accumulator mode follows the reference's nthread=1 order: agreement is at the float32 rounding
    # level (powf/x*x and f32 promotion differences of the shim, see test_oracle_tsc); float64 accumulators differ
    # from the reference's own float32 sums by its accumulation error.
    _check(tab, g, name)
    # invariant of tests/test_power.py:58-61: monopole == mode-weighted mean of the wedges
    p, nm = np.asarray(tab['power'], dtype='f8'), np.asarray(tab['N_mode'], dtype='f8')
    with np.errstate(invalid='ignore'):
        mono = np.nansum(p * nm, axis=1) / nm.sum(axis=1)
    ok = nm.sum(axis=1) > 0
    np.testing.assert_allclose(np.asarray(tab['poles'])[ok, 0], mono[ok], rtol=1e-5 if not accum64 else 1e-6)


def test_weights_squeeze():
    g = load_golden('power_cases')
    rng = np.random.default_rng(5)
    w = (0.5 + rng.random(N, dtype='f4')).astype('f4')
    tab = oracle.calc_power(_pos(), L, kbins=10, mubins=None, paste='TSC', nmesh=NMESH, compensated=True,
                            interlaced=False, w=w, poles=[0, 2])
    assert tab['power'].ndim == 1 and 'mu_mid' not in tab
    _check(tab, g, 'TSC_weights_squeeze')


def test_cross_logk_defaults_odd():
    g = load_golden('power_cases')
    pos2 = synth.synth_positions(N // 2, L, seed=301, clustered=True)
    tab = oracle.calc_power(_pos(), L, kbins=9, mubins=3, paste='TSC', nmesh=NMESH, compensated=True, interlaced=True,
                            pos2=pos2, poles=[0, 2, 4])
    _check(tab, g, 'TSC_cross')   # cross power passes through zero: the floor of assert_spectrum_close
    tab = oracle.calc_power(_pos(), L, kbins=8, mubins=2, logk=True, paste='TSC', nmesh=NMESH, compensated=False,
                            interlaced=False)
    _check(tab, g, 'TSC_logk')
    tab = oracle.calc_power(_pos(), L, paste='TSC', nmesh=24, compensated=True, interlaced=True)
    _check(tab, g, 'TSC_defaults_n24')
    tab = oracle.calc_power(_pos(), L, kbins=7, mubins=2, paste='TSC', nmesh=27, compensated=True, interlaced=True,
                            poles=[0, 2])
    _check(tab, g, 'TSC_odd27')


def test_pk_from_deltak():
    """bin_kmu in the reference's float32/nthread=1 order on a given spectrum: no FFT, no deposit in between"""
    g = load_golden('power_cases')
    n = 20
    ke, me = oracle.get_k_mu_edges(L, np.pi * n / L, 6, 3, False)
    r = oracle.calc_pk_from_deltak(g['deltak.f1'], L, ke, me, field2_fft=g['deltak.f2'],
                                   poles=np.array([0, 2, 4, 6]))
    for k in ('N_mode', 'N_mode_poles'):
        np.testing.assert_array_equal(r[k], g[f'deltak.cross.{k}'])
    for k in ('power', 'binned_poles', 'k_avg'):
        # cross power of two random fields: signed terms cancel, so the floor is set by max|value|
        want = g[f'deltak.cross.{k}']
        np.testing.assert_allclose(r[k], want, rtol=3e-6, atol=2e-6 * np.abs(want).max(), err_msg=k)
    r = oracle.calc_pk_from_deltak(g['deltak.f1'], L, ke, me, squeeze_mu_axis=False)
    np.testing.assert_array_equal(r['N_mode'], g['deltak.auto.N_mode'])
    np.testing.assert_allclose(r['power'], g['deltak.auto.power'], rtol=3e-6)
    np.testing.assert_allclose(r['k_avg'], g['deltak.auto.k_avg'], rtol=3e-6)
    assert r['binned_poles'].shape == (0, 6)
    # float64 accumulators agree with the float32 reference order to float32 accumulation error
    r64 = oracle.calc_pk_from_deltak(g['deltak.f1'], L, ke, me, squeeze_mu_axis=False, accum64=True, nthread=3)
    np.testing.assert_array_equal(r64['N_mode'], r['N_mode'])
    np.testing.assert_allclose(r64['power'], r['power'], rtol=1e-5)


@pytest.mark.parametrize('paste', ['TSC', 'CIC'])
@pytest.mark.parametrize('inter', [False, True])
def test_window(paste, inter):
    g = load_golden('power_cases')
    np.testing.assert_array_equal(oracle.get_W_compensated(L, 32, paste, inter), g[f'W.{paste}_i{int(inter)}'])
    with pytest.raises(ValueError):
        oracle.get_W_compensated(L, 32, 'NGP', inter)


def test_shot_noise_known_answer():
    """uniform randoms: P(k) -> L^3/N (SURVEY.md 8d C3 known answer)"""
    pos = synth.synth_positions(200000, 1000.0, seed=300)
    tab = oracle.calc_power(pos, 1000.0, kbins=8, paste='TSC', nmesh=64, compensated=True, interlaced=True,
                            nthread=4, accum64=True)
    assert abs(np.mean(tab['power'][2:]) / (1000.0**3 / 200000) - 1) < 0.02


@pytest.mark.parametrize('paste', ['TSC', 'CIC'])
@pytest.mark.parametrize('comp,inter', [(False, False), (True, False), (False, True), (True, True)])
def test_analytic_known_answer_matches_calc_power(paste, comp, inter):
    """the closed-form evaluator behind the full-size known-answer tests (oracle.pk_of_particles_analytic: cloud
    transforms of a handful of particles as separable terms, no mesh) against the oracle's calc_power - itself pinned to
    the reference by the golden vectors - on a mesh small enough to build"""
    from oracle import oracle
    L, nmesh = 300.0, 48
    rng = np.random.default_rng(17)
    pos = (rng.random((6, 3)) * L).astype(np.float32)
    pos[0] = (np.floor(pos[0] / (L / nmesh)) + 0.5) * (L / nmesh)   # one particle exactly half-way between cell centres
    w = (rng.random(6) + 0.5).astype(np.float32)
    kw = dict(kbins=12, mubins=3, k_max=np.pi * nmesh / L, paste=paste, nmesh=nmesh, compensated=comp, interlaced=inter,
              poles=[0, 2, 4])
    ref = oracle.calc_power(pos.copy(), L, w=w, nthread=2, accum64=True, **kw)
    kedges = np.concatenate([ref['k_min'], ref['k_max'][-1:]])
    muedges = np.linspace(0, 1, 4)
    got = oracle.pk_of_particles_analytic(pos, w, L, nmesh, kedges, muedges, [0, 2, 4], paste=paste, compensated=comp,
                                          interlaced=inter, nthread=2)
    np.testing.assert_array_equal(got['N_mode'], ref['N_mode'])
    scale = np.abs(ref['power']).max()
    np.testing.assert_allclose(got['power'], ref['power'], rtol=5e-6, atol=5e-6 * scale)
    np.testing.assert_allclose(got['poles'], ref['poles'], rtol=5e-6, atol=5e-6 * scale)
    np.testing.assert_allclose(got['k_avg'], ref['k_avg'], rtol=1e-6)


@pytest.mark.parametrize('n', [16, 21])
def test_exported_pieces_against_reference_goldens(n):
    """bin_kmu, get_raw_power, shift_field_fft, get_interlaced_field_fft of the reference (tests/golden/power_exports.npz,
    oracle/make_golden.py exports) restated by the oracle"""
    g = load_golden('power_exports')
    Lb = float(g['meta.L'])
    f1, f2 = g[f'n{n}.f1'], g[f'n{n}.f2']
    np.testing.assert_allclose(oracle.get_raw_power(f1), g[f'n{n}.raw_auto'], rtol=1e-6)
    np.testing.assert_allclose(oracle.get_raw_power(f1, f2), g[f'n{n}.raw_cross'], rtol=1e-5, atol=1e-6)
    res = oracle.bin_kmu(n, Lb, g[f'n{n}.kedges'], g[f'n{n}.muedges'], g[f'n{n}.raw_auto'], poles=np.array([0, 2, 4]))
    for name, a in zip(('power', 'N_mode', 'poles', 'N_mode_poles', 'k_avg'), res):
        if name.startswith('N_'):
            np.testing.assert_array_equal(a, g[f'n{n}.kmu.{name}'])
        else:
            assert_spectrum_close(a, g[f'n{n}.kmu.{name}'], rtol=1e-5, err_msg=name)
    res = oracle.bin_kmu(n, Lb, g[f'n{n}.redges'], np.array([0.0, 0.5, 1.0]), g[f'n{n}.xi'], poles=np.array([0, 2]), fourier=False)
    for name, a in zip(('power', 'N_mode', 'poles', 'N_mode_poles', 'k_avg'), res):
        if name.startswith('N_'):
            np.testing.assert_array_equal(a, g[f'n{n}.rmu.{name}'])
        else:
            assert_spectrum_close(a, g[f'n{n}.rmu.{name}'], rtol=1e-5, err_msg=name)
    a = f1.copy()
    oracle.shift_field_fft(a, f2, n, Lb, Lb / n)
    assert np.abs(a - g[f'n{n}.shifted']).max() <= 2e-6 * np.abs(g[f'n{n}.shifted']).max()
    pos = synth.synth_positions(3000, Lb, seed=70 + n, clustered=True)
    for key, paste, w in ((f'n{n}.il_tsc', 'TSC', None), (f'n{n}.il_cic_w', 'CIC', g[f'n{n}.w'])):
        b = oracle.get_interlaced_field_fft(pos.copy(), Lb, n, paste, w)
        assert b.dtype == np.complex64 and np.abs(b - g[key]).max() <= 3e-6 * np.abs(g[key]).max(), key
