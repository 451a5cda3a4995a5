"""HIP power-spectrum path (deposit -> hipFFT -> fused binning) vs golden vectors of the reference's calc_power and
vs the float64-accumulating CPU oracle.  Tolerance: 1e-5 relative on P (north_star), exact N_mode.
Needs an MI355X: run with `-m gpu`."""
import warnings

import numpy as np
import pytest
from conftest import assert_spectrum_close, load_golden

from abacusutils_amd import synth

pytestmark = pytest.mark.gpu

L, N, NMESH = 500.0, 20000, 32
RTOL = 1e-5


def _pos():
    return synth.synth_positions(N, L, seed=300, clustered=True)


def _check_golden(tab, g, name, rtol=RTOL):
    """against what the REFERENCE returned (tests/golden/power_cases.npz), at the north_star tolerance for every case - CIC and
    the cross spectrum included; values more than a decade below the array's largest (zero crossings of a cross spectrum
    or of the l = 2, 4 multipoles, the DC bin) are held to 1e-6 of the largest instead (conftest.assert_spectrum_close)"""
    for k in ('power', 'k_avg', 'poles'):
        if f'{name}.{k}' in g:
            assert_spectrum_close(tab[k], g[f'{name}.{k}'], rtol=rtol, err_msg=f'{name}.{k}')
    for k in ('N_mode', 'N_mode_poles'):
        if f'{name}.{k}' in g:
            np.testing.assert_array_equal(np.asarray(tab[k]), g[f'{name}.{k}'], err_msg=f'{name}.{k}')
    np.testing.assert_allclose(tab['k_mid'], g[f'{name}.k_mid'], rtol=1e-15)


def _check_oracle(tab, ref, rtol=RTOL, poles_rtol_factor=1.0):
    np.testing.assert_array_equal(np.asarray(tab['N_mode']), ref['N_mode'])
    for k in ('power', 'k_avg', 'poles'):
        if k in ref:
            assert_spectrum_close(tab[k], ref[k], rtol=rtol * (poles_rtol_factor if k == 'poles' else 1.0), err_msg=k)


@pytest.mark.parametrize('paste', ['TSC', 'CIC'])
@pytest.mark.parametrize('comp', [False, True])
@pytest.mark.parametrize('inter', [False, True])
def test_calc_power_modes(paste, comp, inter):
    """the 8 paste x compensated x interlaced combinations of the reference's tests/test_power.py:22-35"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    g = load_golden('power_cases')
    kw = dict(kbins=12, mubins=4, k_max=np.pi * NMESH / L + 1e-6, paste=paste, nmesh=NMESH, compensated=comp,
              interlaced=inter, poles=[0, 2, 4])
    tab = calc_power(_pos(), L, **kw)
    # vs what the reference returned
    _check_golden(tab, g, f'{paste}_c{int(comp)}_i{int(inter)}')
    # vs the float64-accumulating oracle: the north_star tolerance
    _check_oracle(tab, oracle.calc_power(_pos(), L, nthread=4, accum64=True, **kw))
    # tests/test_power.py:58-61: monopole == mode-weighted mean of the wedges
    p, nm = np.asarray(tab['power'], dtype='f8'), np.asarray(tab['N_mode'], dtype='f8')
    ok = nm.sum(axis=1) > 0
    mono = (p * nm).sum(axis=1)[ok] / nm.sum(axis=1)[ok]
    np.testing.assert_allclose(np.asarray(tab['poles'])[ok, 0], mono, rtol=1e-6)


def test_weights_cross_logk_defaults_odd():
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    g = load_golden('power_cases')
    rng = np.random.default_rng(5)
    w = (0.5 + rng.random(N, dtype='f4')).astype('f4')
    pos2 = synth.synth_positions(N // 2, L, seed=301, clustered=True)
    cases = [
        ('TSC_weights_squeeze', dict(kbins=10, mubins=None, paste='TSC', nmesh=NMESH, compensated=True,
                                     interlaced=False, w=w, poles=[0, 2]), None),
        ('TSC_cross', dict(kbins=9, mubins=3, paste='TSC', nmesh=NMESH, compensated=True, interlaced=True,
                           poles=[0, 2, 4]), pos2),
        ('TSC_logk', dict(kbins=8, mubins=2, logk=True, paste='TSC', nmesh=NMESH, compensated=False,
                          interlaced=False), None),
        ('TSC_defaults_n24', dict(paste='TSC', nmesh=24, compensated=True, interlaced=True), None),
        ('TSC_odd27', dict(kbins=7, mubins=2, paste='TSC', nmesh=27, compensated=True, interlaced=True,
                           poles=[0, 2]), None),
    ]
    for name, kw, p2 in cases:
        tab = calc_power(_pos(), L, pos2=None if p2 is None else p2.copy(), **kw)
        _check_golden(tab, g, name)
        ref = oracle.calc_power(_pos(), L, pos2=None if p2 is None else p2.copy(), nthread=4, accum64=True, **kw)
        _check_oracle(tab, ref)
        if kw.get('mubins', 1) is None:
            assert np.asarray(tab['power']).ndim == 1 and 'mu_mid' not in tab


def test_pk_from_deltak():
    """calc_pk_from_deltak on caller-supplied spectra (the zcv call pattern, hod/zcv/tracer_power.py:160,208)"""
    from abacusutils_amd.analysis.power_spectrum import calc_pk_from_deltak, get_k_mu_edges
    g = load_golden('power_cases')
    n = 20
    ke, me = get_k_mu_edges(L, np.pi * n / L, 6, 3, False)
    r = calc_pk_from_deltak(g['deltak.f1'], L, ke, me, field2_fft=g['deltak.f2'], poles=np.array([0, 2, 4, 6]))
    for k in ('N_mode', 'N_mode_poles'):
        np.testing.assert_array_equal(r[k], g[f'deltak.cross.{k}'])
    for k in ('power', 'binned_poles', 'k_avg'):
        want = g[f'deltak.cross.{k}']
        np.testing.assert_allclose(r[k], want, rtol=1e-5, atol=3e-6 * np.abs(want).max(), err_msg=k)
    r = calc_pk_from_deltak(g['deltak.f1'], L, ke, me, squeeze_mu_axis=False)
    np.testing.assert_array_equal(r['N_mode'], g['deltak.auto.N_mode'])
    np.testing.assert_allclose(r['power'], g['deltak.auto.power'], rtol=5e-6)
    np.testing.assert_allclose(r['k_avg'], g['deltak.auto.k_avg'], rtol=5e-6)
    assert r['binned_poles'].shape == (0, 6)


def test_get_field_fft_and_window():
    from abacusutils_amd.analysis.power_spectrum import get_field_fft, get_W_compensated
    from oracle import oracle
    g = load_golden('power_cases')
    for paste in ('TSC', 'CIC'):
        for inter in (False, True):
            np.testing.assert_array_equal(get_W_compensated(L, 32, paste, inter), g[f'W.{paste}_i{int(inter)}'])
    with pytest.raises(ValueError):
        get_W_compensated(L, 32, 'NGP', True)
    W = get_W_compensated(L, NMESH, 'TSC', True)
    a = get_field_fft(_pos(), L, NMESH, 'TSC', None, W, True, True)
    b = oracle.get_field_fft(_pos(), L, NMESH, 'TSC', None, W, True, True, nthread=2)
    assert a.shape == b.shape and a.dtype == np.complex64
    assert np.abs(a - b).max() < 2e-6 * np.abs(b).max()


def test_errors():
    from abacusutils_amd.analysis.power_spectrum import calc_power
    with pytest.raises(ValueError):
        calc_power(_pos(), L, paste='NGP', nmesh=16)


@pytest.mark.parametrize('nmesh,interlaced', [(256, False), (384, True)])
def test_shot_noise_and_oracle_at_scale(nmesh, interlaced):
    """uniform randoms: P(k) -> L^3/N (SURVEY 8d known answer), and 1e-5 agreement with the oracle at 3e6 particles"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    n, box = 3_000_000, 2000.0
    pos = synth.synth_positions(n, box, seed=300)
    kw = dict(kbins=64, mubins=4, paste='TSC', nmesh=nmesh, compensated=True, interlaced=interlaced, poles=[0, 2, 4])
    tab = calc_power(pos.copy(), box, **kw)
    shot = box**3 / n
    p = np.asarray(tab['poles'])[:, 0]
    assert abs(np.mean(p[8:48]) / shot - 1) < 0.01
    ref = oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw)
    _check_oracle(tab, ref)


@pytest.mark.parametrize('nmesh', [64, 128, 512])
def test_native_fft_sizes(nmesh):
    """power-of-two meshes go through the hand-written three-pass FFT (csrc/fft.hip): spectrum and P(k) vs the oracle"""
    from abacusutils_amd.analysis.power_spectrum import calc_power, get_field_fft
    from oracle import oracle
    n, box = 400_000, 1000.0
    pos = synth.synth_positions(n, box, seed=77, clustered=True)
    if nmesh <= 128:
        # uncompensated spectra (the window division amplifies float32 noise near Nyquist by ~60x): the transform
        # agrees with scipy's pocketfft at the float32 rounding level of the deposit (observed 3e-7 of the maximum)
        for inter in (False, True):
            a = get_field_fft(pos.copy(), box, nmesh, 'TSC', None, None, False, inter)
            b = oracle.get_field_fft(pos.copy(), box, nmesh, 'TSC', None, None, False, inter, nthread=4)
            assert np.abs(a - b).max() < 1e-6 * np.abs(b).max()
    kw = dict(kbins=32, mubins=3, paste='TSC', nmesh=nmesh, compensated=True, interlaced=nmesh < 512, poles=[0, 2, 4])
    tab = calc_power(pos.copy(), box, **kw)
    ref = oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw)
    _check_oracle(tab, ref)


def test_fused_fft_against_oracle(options):
    """fft.hip's fused form (first radix-2 stage of y and x inside the z pass, permuted row order undone by the binning)
    on a 256^3 mesh against the CPU oracle: auto and cross spectra, interlaced + compensated, multipoles"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    options.set('fft_fuse_small', 1)
    n, box = 600_000, 1000.0
    pos = synth.synth_positions(n, box, seed=79, clustered=True)
    pos2 = synth.synth_positions(n // 2, box, seed=80, clustered=True)
    for extra in (dict(), dict(pos2=pos2.copy())):
        kw = dict(kbins=40, mubins=4, paste='TSC', nmesh=256, compensated=True, interlaced=True, poles=[0, 2, 4], **extra)
        tab = calc_power(pos.copy(), box, **kw)
        ref = oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw)
        _check_oracle(tab, ref)


@pytest.mark.parametrize('cross', [False, True])
def test_float64_positions_use_float64_cloud_weights(cross):
    """positions given in float64 (analysis/tsc.py:400: the cloud arithmetic follows the dtype of the positions; the mesh
    stays float32): against the oracle, which deposits float64 positions in float64; in place wrapping of the caller's array"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    rng = np.random.default_rng(17)
    n, box = 300_000, 2000.0
    pos = (rng.random((n, 3)) * 1.2 - 0.1) * box                      # float64, some outside [0, L): wrapped in place
    w = rng.random(n) + 0.5
    extra = dict(pos2=(rng.random((n // 2, 3)) * box)) if cross else {}
    kw = dict(kbins=24, mubins=3, paste='TSC', nmesh=128, compensated=True, interlaced=True, poles=[0, 2], w=w)
    a_in = pos.copy()
    tab = calc_power(a_in, box, **kw, **extra)
    assert a_in.dtype == np.float64 and a_in.min() >= 0.0 and a_in.max() < box        # wrapped like tsc_parallel does
    ref = oracle.calc_power(pos.copy(), box, nthread=4, accum64=True, **{k: (v.copy() if hasattr(v, 'copy') else v) for k, v in {**kw, **extra}.items()})
    _check_oracle(tab, ref)


@pytest.mark.parametrize('n', [50_000, 2_500_000])       # the per-tile atomic list build and the multisplit
def test_positions_are_wrapped_in_place_only_when_needed(n):
    """the reference wraps the caller's positions in place (tsc.py:171-173, through calc_power / get_field / get_field_fft);
    positions already inside the box come back untouched - the host entry points copy them back only when one moved"""
    from abacusutils_amd.analysis.power_spectrum import calc_power, get_field, get_field_fft
    from oracle import oracle
    rng = np.random.default_rng(23)
    box = 1000.0
    inside = (rng.random((n, 3), dtype=np.float32) * np.float32(box * 0.999999)).astype(np.float32)
    outside = inside.copy()
    outside[::7] += np.float32(box)
    outside[3::11, 1] -= np.float32(box)
    want = outside.copy()
    oracle.wrap_inplace(want, box)
    kw = dict(kbins=16, mubins=2, paste='TSC', nmesh=128, compensated=False, interlaced=False)
    calls = {'calc_power': lambda p: calc_power(p, box, **kw), 'get_field': lambda p: get_field(p, box, 128, 'TSC'),
             'get_field_fft': lambda p: get_field_fft(p, box, 128, 'TSC', None, None, False, False)}
    results = {}
    for name, f in calls.items():
        a, b = inside.copy(), outside.copy()
        ra, rb = f(a), f(b)
        np.testing.assert_array_equal(a, inside, err_msg=name)
        np.testing.assert_array_equal(b, want, err_msg=name)
        results[name] = (ra, rb)
    # the wrapped copy describes the same particles up to the rounding of x + L - L
    np.testing.assert_allclose(results['calc_power'][1]['power'], results['calc_power'][0]['power'], rtol=2e-3)
    # CIC does not wrap (cic.py): the array stays as it is
    c = inside.copy()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        get_field(c, box, 64, 'CIC')
    np.testing.assert_array_equal(c, inside)


@pytest.mark.parametrize('nmesh', [1024])
def test_fused_fft_matches_three_pass(options, nmesh):
    """production sizes of the fused form against the plain three-pass transform on the same particles"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    pos = synth.synth_positions(3_000_000, 1000.0, seed=81, clustered=True)
    kw = dict(kbins=64, mubins=4, paste='TSC', nmesh=nmesh, compensated=False, interlaced=True, poles=[0, 2, 4])
    a = calc_power(pos.copy(), 1000.0, **kw)
    options.set('fft_nofuse', 1)
    b = calc_power(pos.copy(), 1000.0, **kw)
    np.testing.assert_array_equal(a['N_mode'], b['N_mode'])
    np.testing.assert_allclose(a['power'], b['power'], rtol=2e-5, atol=1e-6 * np.abs(b['power']).max())
    np.testing.assert_allclose(a['poles'], b['poles'], rtol=2e-5, atol=1e-6 * np.abs(b['power']).max())


@pytest.mark.parametrize('paste,d', [('TSC', 0.0), ('TSC', 3.9), ('CIC', 0.0)])
def test_get_field(paste, d):
    """get_field (analysis/power_spectrum.py:808-857): deposit + normalisation fused on the device"""
    from abacusutils_amd.analysis.power_spectrum import get_field
    from oracle import oracle
    box, nmesh, n = 500.0, 64, 50_000
    pos = synth.synth_positions(n, box, seed=91, clustered=True)
    w = np.random.default_rng(2).random(n, dtype=np.float32) + np.float32(0.5)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        a = get_field(pos.copy(), box, nmesh, paste, w=w, d=d)
    b = oracle.get_field(pos.copy(), box, nmesh, paste, w=w, d=d, nthread=2)
    assert a.shape == (nmesh,) * 3 and a.dtype == np.float32
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5 * np.abs(b).max())


@pytest.mark.parametrize('n', [16, 21])
def test_zcv_helpers_against_reference_goldens(n):
    """bin_kppi, project_3d_to_poles, pk_to_xi, expand_poles_to_3d, get_smoothing, get_delta_mu2 on the device against the
    outputs of the shimmed reference (tests/golden/power_helpers.npz)"""
    from abacusutils_amd.analysis import power_spectrum as ps
    g = load_golden('power_helpers')
    L = float(g['meta.L'])
    p3d, ke, xi, re = g[f'n{n}.p3d'], g[f'n{n}.kedges'], g[f'n{n}.xi'], g[f'n{n}.redges']
    m, c = ps.bin_kppi(n, L, ke, np.pi * n / L * 1.01, 5, p3d)
    np.testing.assert_array_equal(c, g[f'n{n}.kppi.counts'])
    np.testing.assert_allclose(m, g[f'n{n}.kppi.mean'], rtol=2e-5)
    m, c = ps.bin_kppi(n, L, re, L / 2 * 1.01, 4, xi, fourier=False)
    np.testing.assert_array_equal(c, g[f'n{n}.rppi.counts'])
    np.testing.assert_allclose(m, g[f'n{n}.rppi.mean'], rtol=2e-4, atol=2e-5)
    bp, npo = ps.project_3d_to_poles(ke, p3d, L, [0, 2, 4])
    np.testing.assert_array_equal(npo, g[f'n{n}.p2poles.N'])
    np.testing.assert_allclose(bp, g[f'n{n}.p2poles.poles'], rtol=2e-5, atol=1e-5 * np.abs(g[f'n{n}.p2poles.poles']).max())
    rb, xp, nr = ps.pk_to_xi(p3d.copy(), L, re, poles=[0, 2, 4])
    np.testing.assert_allclose(rb, g[f'n{n}.pk2xi.r'])
    np.testing.assert_array_equal(nr, g[f'n{n}.pk2xi.N'])
    np.testing.assert_allclose(xp, g[f'n{n}.pk2xi.poles'], rtol=2e-4, atol=2e-5 * np.abs(g[f'n{n}.pk2xi.poles']).max())
    np.testing.assert_allclose(ps.expand_poles_to_3d(g[f'n{n}.expand.k_ell'], g[f'n{n}.expand.P_ell'], n, L, [0, 2, 4]),
                               g[f'n{n}.expand.Pk'], rtol=2e-5, atol=2e-4 * np.abs(g[f'n{n}.expand.Pk']).max())
    np.testing.assert_allclose(ps.get_smoothing(n, L, 7.5), g[f'n{n}.smoothing'], rtol=3e-6)
    np.testing.assert_allclose(ps.get_delta_mu2(g[f'n{n}.delta'], n), g[f'n{n}.delta_mu2'], rtol=2e-6, atol=1e-7)


def test_zcv_helpers_larger_mesh_against_oracle():
    from abacusutils_amd.analysis import power_spectrum as ps
    from oracle import oracle
    n, L = 96, 700.0
    rng = np.random.default_rng(3)
    p3d = (rng.random((n, n, n // 2 + 1), dtype=np.float32) * 50).astype(np.float32)
    ke = np.linspace(0.0, np.pi * n / L, 25)
    bp, npo = ps.project_3d_to_poles(ke, p3d, L, [0, 2, 4])
    bo, no = oracle.project_3d_to_poles(ke, p3d, L, [0, 2, 4], nthread=4)
    np.testing.assert_array_equal(npo, no)
    np.testing.assert_allclose(bp, bo, rtol=2e-5, atol=1e-5 * np.abs(bo).max())
    re = np.linspace(0.0, 150.0, 16)
    _, xp, nr = ps.pk_to_xi(p3d.copy(), L, re)
    _, xo, no = oracle.pk_to_xi(p3d.copy(), L, re, nthread=4)
    np.testing.assert_array_equal(nr, no)
    np.testing.assert_allclose(xp, xo, rtol=2e-4, atol=3e-5 * np.abs(xo).max())
    m, c = ps.bin_kppi(n, L, ke, np.pi * n / L * 1.01, 12, p3d)
    mo, co = oracle.bin_kppi(n, L, ke, np.pi * n / L * 1.01, 12, p3d)
    np.testing.assert_array_equal(c, co)
    np.testing.assert_allclose(m, mo, rtol=2e-5)
    m, c = ps.bin_kppi(n, L, ke, 0.2, 6, p3d)            # pi range shorter than the grid: the kz loop breaks
    mo, co = oracle.bin_kppi(n, L, ke, 0.2, 6, p3d)
    np.testing.assert_array_equal(c, co)
    np.testing.assert_allclose(m, mo, rtol=2e-5)


def test_full_size_2048_properties(options):
    """BASELINE size (nmesh 2048, 1e8 particles), size-independent properties: (1) the fused form of the hand-written
    FFT and its plain three-pass form give the same binned spectrum (hipFFT is no comparator here: its padded in-place
    2048^3 R2C is wrong on ROCm 7.2 - 0.94 of the shot-noise answer - and the library refuses it); (2) uniform randoms -> P0(k) = L^3/N (known answer) with
    interlacing + compensation; (3) N_mode of the innermost k bins equals a brute-force enumeration of the modes."""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    n, box, nmesh = 100_000_000, 2000.0, 2048
    rng = np.random.default_rng(300)
    pos = rng.random((n, 3), dtype=np.float32) * np.float32(box)
    kw = dict(kbins=256, mubins=4, k_max=np.pi * nmesh / box, paste='TSC', nmesh=nmesh, poles=[0, 2, 4])
    a = calc_power(pos, box, compensated=False, interlaced=False, **kw)
    options.set('fft_nofuse', 1)
    b = calc_power(pos, box, compensated=False, interlaced=False, **kw)
    options.set('fft_nofuse', 0)
    options.set('fft_hipfft', 1)
    with pytest.raises(Exception, match='not usable'):
        calc_power(pos, box, compensated=False, interlaced=False, **kw)
    options.set('fft_hipfft', 0)
    np.testing.assert_array_equal(a['N_mode'], b['N_mode'])
    np.testing.assert_allclose(a['power'], b['power'], rtol=2e-5)
    np.testing.assert_allclose(np.asarray(a['poles'])[:, 0], np.asarray(b['poles'])[:, 0], rtol=2e-5)
    # (3) modes with |k| below the 4th edge, enumerated: kz >= 0 half-space, weight 2 for kz > 0 (bin_kmu :258-262)
    dk = 2 * np.pi / box
    edges = np.linspace(0.0, np.pi * nmesh / box, 257) / dk
    m = int(np.ceil(edges[4])) + 1
    g = np.arange(-m, m + 1)
    kx, ky, kz = np.meshgrid(g, g, np.arange(0, m + 1), indexing='ij')
    k2 = (kx * kx + ky * ky + kz * kz).astype(np.float32)
    wt = np.where(kz == 0, 1, 2)
    e2 = (edges ** 2).astype(np.float32)
    nm = np.asarray(a['N_mode']).sum(axis=1)
    for bk in range(4):
        sel = (k2 > e2[bk]) & (k2 <= e2[bk + 1]) if bk else (k2 >= e2[0]) & (k2 <= e2[1])
        assert nm[bk] == wt[sel].sum(), bk
    # (2) shot noise
    c = calc_power(pos, box, compensated=True, interlaced=True, **kw)
    p0 = np.asarray(c['poles'])[:, 0]
    shot = box**3 / n
    assert abs(np.mean(p0[32:224]) / shot - 1) < 2e-3
    assert np.all(np.abs(p0[64:224] / shot - 1) < 0.02)


@pytest.mark.parametrize('paste,nmesh', [('CIC', 256), ('TSC', 98), ('CIC', 98), ('TSC', 130)])
def test_interlaced_shared_lists(paste, nmesh, options):
    """>= 2e6 particles: the two deposits of an interlaced pair share one list build (tsc.hip `list_mode`, lists for the
    4-cell union of both clouds).  CIC and meshes whose last tile has fewer than 4 cells (generic tile enumeration)
    against the oracle, and against the same call with the sharing switched off"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    n, box = 2_500_000, 1000.0
    pos = synth.synth_positions(n, box, seed=91, clustered=True)
    kw = dict(kbins=24, mubins=3, paste=paste, nmesh=nmesh, compensated=True, interlaced=True, poles=[0, 2, 4])
    tab = calc_power(pos.copy(), box, **kw)
    ref = oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw)
    _check_oracle(tab, ref)
    options.set('tsc_noshare', 1)
    tab2 = calc_power(pos.copy(), box, **kw)
    np.testing.assert_array_equal(tab['N_mode'], tab2['N_mode'])
    np.testing.assert_allclose(tab['power'], tab2['power'], rtol=1e-6, atol=1e-9 * np.abs(np.asarray(tab2['power'])).max())


def test_many_multipoles_generic_path():
    """three or more ell != 0 multipoles (or ell > 4) take the binning kernel's generic pole loop"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    pos = synth.synth_positions(300_000, 1000.0, seed=93, clustered=True)
    for poles in ([0, 2, 4, 6], [2, 6], [0, 8, 4, 2, 6]):
        kw = dict(kbins=20, mubins=3, paste='TSC', nmesh=96, compensated=True, interlaced=True, poles=poles)
        tab = calc_power(pos.copy(), 1000.0, **kw)
        ref = oracle.calc_power(pos.copy(), 1000.0, nthread=oracle.max_threads(), accum64=True, **kw)
        _check_oracle(tab, ref)


@pytest.mark.parametrize('seed', range(16))     # more seeds: scripts/gpu_pk_fuzz.sh
def test_random_option_sweep(seed):
    """seeded random calc_power calls (paste, compensation, interlacing, linear / log / explicit k bins, mu bins, k_max,
    weights, cross spectra, multipole sets, even and odd meshes, native and hipFFT sizes) against the float64 oracle"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    rng = np.random.default_rng(5000 + seed)
    box = float(rng.choice([250.0, 1000.0, 2000.0]))
    nmesh = int(rng.choice([16, 21, 24, 32, 45, 64, 72]))
    n = int(rng.integers(3000, 40000))
    pos = synth.synth_positions(n, box, seed=600 + seed, clustered=bool(seed % 2))
    kw = dict(paste=str(rng.choice(['TSC', 'CIC'])), nmesh=nmesh, compensated=bool(rng.integers(2)),
              interlaced=bool(rng.integers(2)), mubins=int(rng.integers(1, 7)))
    kn = np.pi * nmesh / box
    mode = seed % 3
    if mode == 0:
        kw.update(kbins=int(rng.integers(3, 20)), k_max=float(kn * rng.uniform(0.4, 1.0)), logk=False)
    elif mode == 1:
        kw.update(kbins=int(rng.integers(3, 12)), k_max=float(kn * rng.uniform(0.5, 1.0)), logk=True)
    else:
        kw.update(kbins=np.sort(rng.uniform(0.0, kn, int(rng.integers(3, 10)))))
    poles = [None, [0], [0, 2], [0, 2, 4], [2, 4], [0, 2, 4, 6]][int(rng.integers(6))]
    if poles is not None:
        kw['poles'] = poles
    if rng.random() < 0.4:
        kw['w'] = rng.uniform(0.5, 1.5, n).astype(np.float32)
    if rng.random() < 0.35:
        n2 = int(rng.integers(2000, 20000))
        kw['pos2'] = synth.synth_positions(n2, box, seed=700 + seed, clustered=True)
        if rng.random() < 0.5:
            kw['w2'] = rng.uniform(0.5, 1.5, n2).astype(np.float32)
    cp = lambda d: {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in d.items()}   # noqa: E731
    tab = calc_power(pos.copy(), box, **cp(kw))
    ref = oracle.calc_power(pos.copy(), box, nthread=4, accum64=True, **cp(kw))
    # window-compensated CIC amplifies float32 noise near Nyquist: the oracle comparison of those cases is looser
    # 13 P_6(mu) has coefficients up to 650 with alternating signs: the float32 noise of the modes (1e-6 of |delta_k|^2, FFT
    # round-off that differs between the two transforms) is amplified accordingly in a bin of few modes (seed 5103 of the
    # fuzz: 3.5e-5 in one l = 6 value)
    _check_oracle(tab, ref, rtol=1e-5 if kw['paste'] == 'TSC' else 3e-5, poles_rtol_factor=5.0 if 6 in (poles or []) else 1.0)
    if 'N_mode_poles' in ref:
        np.testing.assert_array_equal(np.asarray(tab['N_mode_poles']), ref['N_mode_poles'])


def _kmu_edges(res, nmu):
    return np.concatenate([np.asarray(res['k_min']), np.asarray(res['k_max'])[-1:]]), np.linspace(0, 1, nmu + 1)


@pytest.mark.parametrize('comp,inter', [(False, False), (True, True)])     # (True, False): the compensated fused pass is held at 1024 by test_fused_last_pass_matches_spectrum_bin (suite budget, r06)
def test_full_size_2048_analytic_known_answer(comp, inter):
    """nmesh 2048 against a closed form: a handful of weighted particles at generic positions; the
    discrete transform of their TSC clouds is a sum of separable terms, evaluated in float64 and binned by bin_kmu's rule
    without any mesh (oracle.pk_of_particles_analytic, itself held to the oracle's calc_power on a buildable mesh by
    tests/test_oracle_power.py).  Every (k, mu) bin, the multipoles and k_avg to 1e-5; N_mode exact - over all 4.3e9
    modes, through the fused last pass (non-interlaced) and through spectrum_bin (interlaced + compensated)."""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    box, nmesh = 2000.0, 2048
    rng = np.random.default_rng(41)
    pos = (rng.random((7, 3)) * box).astype(np.float32)
    pos[0] = (np.floor(pos[0] / (box / nmesh)) + 0.5) * (box / nmesh)      # half-way between cell centres (round-half-even)
    pos[1] = np.floor(pos[1] / (box / nmesh)) * (box / nmesh)              # exactly on a cell centre
    w = (rng.random(7) + 0.5).astype(np.float32)
    kw = dict(kbins=128, mubins=4, k_max=np.pi * nmesh / box, paste='TSC', nmesh=nmesh, poles=[0, 2, 4], compensated=comp,
              interlaced=inter)
    got = calc_power(pos.copy(), box, w=w, **kw)
    kedges, muedges = _kmu_edges(got, 4)
    want = oracle.pk_of_particles_analytic(pos, w, box, nmesh, kedges, muedges, [0, 2, 4], paste='TSC', compensated=comp,
                                           interlaced=inter, nthread=oracle.max_threads())
    np.testing.assert_array_equal(np.asarray(got['N_mode']), want['N_mode'])
    np.testing.assert_array_equal(np.asarray(got['N_mode_poles']), want['N_mode_poles'])
    scale = np.abs(want['power']).max()
    np.testing.assert_allclose(np.asarray(got['power']), want['power'], rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(np.asarray(got['poles']), want['poles'], rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(np.asarray(got['k_avg']), want['k_avg'], rtol=1e-6)


@pytest.mark.parametrize('comp,inter', [(False, False), (True, True)])
def test_c3_full_size_against_oracle(comp, inter):
    """BASELINE config 3 at full size: 1e8 uniform particles (seed 300) -> 1024^3 TSC + FFT + binning,
    HIP against the CPU oracle (float64 accumulation) on the same inputs: N_mode exact, P(k, mu), multipoles <= 1e-5.
    Both modes SURVEY 8(d) names for the config: (compensated, interlaced) = (False, False) and the reference's defaults
    (True, True) (analysis/power_spectrum.py:1131)"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    n, box, nmesh = 100_000_000, 2000.0, 1024
    pos = np.random.default_rng(300).random((n, 3), dtype=np.float32) * np.float32(box)
    kw = dict(kbins=512, mubins=4, k_max=np.pi * nmesh / box + 1e-6, paste='TSC', nmesh=nmesh, poles=[0, 2, 4],
              compensated=comp, interlaced=inter)
    a = calc_power(pos, box, **kw)
    b = oracle.calc_power(pos, box, nthread=oracle.max_threads(), accum64=True, **kw)
    np.testing.assert_array_equal(np.asarray(a['N_mode']), b['N_mode'])
    ok = b['N_mode'] > 0
    np.testing.assert_allclose(np.asarray(a['power'])[ok], b['power'][ok], rtol=1e-5)
    scale = np.abs(b['power']).max()
    np.testing.assert_allclose(np.asarray(a['poles']), b['poles'], rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(np.asarray(a['k_avg'])[ok], b['k_avg'][ok], rtol=1e-6)
    shot = box**3 / n
    # flat shot noise where the (uncompensated) TSC window is still ~1: k <= k_Nyquist / 14 -> W^2 >= 0.99; the few thousand
    # modes of these bins scatter at the per-cent level
    assert abs(np.mean(np.asarray(a['poles'])[4:36, 0]) / shot - 1) < 2.5e-2


@pytest.mark.parametrize('nmesh,comp', [(1024, False), (1024, True)])
def test_fused_last_pass_matches_spectrum_bin(options, nmesh, comp):
    """fft_x_bin (last FFT pass + binning in one kernel, xbin.hip) against the x pass + spectrum_bin on the same particles:
    identical |delta_k|^2 per mode, so the float64 sums agree to rounding.  Three forms of the fused pass: the
    cached-geometry kernel with register runs (production) and with one LDS atomic per pair, and the first-generation
    kernel that walks the edges itself (also what more than 8 mu bins fall back to)."""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power
    pos = synth.synth_positions(3_000_000, 1000.0, seed=83, clustered=True)
    w = np.random.default_rng(3).random(len(pos), dtype=np.float32) + np.float32(0.5)
    gen = _lib.lib().abacus_power_xbin_generation
    for kw in (dict(kbins=64, mubins=4, poles=[0, 2, 4]), dict(kbins=200, mubins=None, poles=[0, 2]),
               dict(kbins=48, mubins=7, poles=[]), dict(kbins=32, mubins=3, poles=[4], logk=True, k_max=1.5),
               dict(kbins=500, mubins=2, poles=[0, 2, 4], k_max=np.pi * nmesh / 1000.0 + 1e-6),     # the bench's 2-dk bins
               dict(kbins=np.array([0.02, 0.021, 0.0215, 0.3, 0.31, 2.0, 9.0]), mubins=np.array([0.0, 0.05, 0.5, 0.51, 1.0]),
                    poles=[2]),                                                                     # ragged edges, past the corner
               dict(kbins=40, mubins=12, poles=[0, 2])):                                            # > 8 mu bins: generation 1
        kw = dict(kw, paste='TSC', nmesh=nmesh, compensated=comp, interlaced=False, w=w)
        options.set('pk_noxbin', 1)
        b = calc_power(pos.copy(), 1000.0, **kw)
        options.set('pk_noxbin', 0)
        nmu = np.asarray(b['N_mode']).shape[1] if np.asarray(b['N_mode']).ndim > 1 else 1
        for name, opts, want_gen in (('runs', {}, 2), ('pairs', {'pk_xbin_pairs': 1}, 2), ('gen1', {'pk_xbin_gen': 1}, 1)):
            for k, v in opts.items():
                options.set(k, v)
            a = calc_power(pos.copy(), 1000.0, **kw)
            for k in opts:
                options.set(k, 0)
            assert gen() == (want_gen if nmu <= 8 else 1), (name, kw['kbins'] if np.isscalar(kw['kbins']) else 'edges')
            np.testing.assert_array_equal(a['N_mode'], b['N_mode'], err_msg=name)
            np.testing.assert_allclose(a['power'], b['power'], rtol=2e-6, atol=1e-7 * np.abs(b['power']).max(), err_msg=name)
            np.testing.assert_allclose(a['k_avg'], b['k_avg'], rtol=1e-6, err_msg=name)
            if kw['poles']:
                np.testing.assert_allclose(a['poles'], b['poles'], rtol=2e-6, atol=2e-7 * np.abs(b['power']).max(), err_msg=name)


@pytest.mark.parametrize('nmesh,comp', [(1024, True), (1024, False), (2048, False)])
def test_fused_last_pass_cross_power_matches_spectrum_bin(options, nmesh, comp):
    """the cross form of the fused last pass (fft_x_bin2<.., CROSS>: the first field's tile kept in registers, the second
    field's through the same LDS, Re(conj(a) b) f32(1 / M)^2 binned from there - get_raw_power with field2_fft,
    analysis/power_spectrum.py:707-727) against two complete transforms + spectrum_bin on the same particles (option
    pk_noxbin_cross), and against the oracle at 1024: LRG x ELG of BASELINE config 5 runs this"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power
    box = 1000.0
    pos = synth.synth_positions(2_000_000, box, seed=191, clustered=True)
    pos2 = synth.synth_positions(1_200_000, box, seed=192, clustered=True)
    pos2[:400_000] = pos[:400_000]            # a shared population: the cross power is not just noise around zero
    w = np.random.default_rng(14).random(len(pos), dtype=np.float32) + np.float32(0.5)
    cases = (dict(kbins=64, mubins=4, poles=[0, 2, 4]), dict(kbins=300, mubins=None, poles=[0, 2], k_max=np.pi * nmesh / box + 1e-6),
             dict(kbins=24, mubins=8, poles=[], logk=True, k_max=2.0, w=w),
             dict(kbins=40, mubins=3, poles=[0, 2, 4], k_max=1.7 * np.pi * nmesh / box))
    for kw in (cases if nmesh == 1024 else cases[:2]):
        kw = dict(kw, paste='TSC', nmesh=nmesh, compensated=comp, interlaced=False)
        _lib.profile_reset()
        _lib.profile_enable(True)
        a = calc_power(pos.copy(), box, pos2=pos2.copy(), **kw)
        _lib.profile_enable(False)
        prof = _lib.profile_get()
        assert 'fft_x_bin' in prof and 'spectrum_bin' not in prof and 'fft_cols_x' not in prof, sorted(prof)
        options.set('pk_noxbin_cross', 1)
        b = calc_power(pos.copy(), box, pos2=pos2.copy(), **kw)
        options.set('pk_noxbin_cross', 0)
        np.testing.assert_array_equal(a['N_mode'], b['N_mode'])
        scale = np.abs(np.asarray(b['power'])).max()
        np.testing.assert_allclose(a['power'], b['power'], rtol=3e-6, atol=3e-7 * scale)
        np.testing.assert_allclose(a['k_avg'], b['k_avg'], rtol=1e-6)
        if kw['poles']:
            np.testing.assert_allclose(a['poles'], b['poles'], rtol=3e-6, atol=5e-7 * scale)
    if nmesh == 1024:
        from oracle import oracle
        kw = dict(kbins=20, mubins=3, k_max=1.2, paste='TSC', nmesh=1024, compensated=comp, interlaced=False, poles=[0, 2])
        s1 = synth.synth_positions(300_000, box, seed=193, clustered=True)
        s2 = synth.synth_positions(200_000, box, seed=194, clustered=True)
        s2[:80_000] = s1[:80_000]
        _check_oracle(calc_power(s1.copy(), box, pos2=s2.copy(), **kw),
                      oracle.calc_power(s1.copy(), box, pos2=s2.copy(), nthread=oracle.max_threads(), accum64=True, **kw))


@pytest.mark.parametrize('nmesh,comp', [(1024, True), (1024, False), (2048, True)])
def test_fused_last_pass_interlaced_pair_matches_spectrum_bin(options, nmesh, comp):
    """the interlaced form of the fused last pass (fft_x_bin2<.., INTER>: the unshifted field's tile kept in registers, the
    shifted field's tile through the same LDS, (a + a' exp(i pi m / n)) f32(0.5 / M) binned from there) against two x passes
    + spectrum_bin<INTER> on the same particles (option pk_noxbin_inter): the reference's default mode of calc_power
    (interlaced=True, compensated=True; analysis/power_spectrum.py:951-998), identical values per mode, float64 sums to
    rounding; and against the oracle at 1024"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power
    box = 1000.0
    pos = synth.synth_positions(2_000_000, box, seed=91, clustered=True)
    w = np.random.default_rng(4).random(len(pos), dtype=np.float32) + np.float32(0.5)
    cases = (dict(kbins=64, mubins=4, poles=[0, 2, 4]), dict(kbins=300, mubins=None, poles=[0, 2], k_max=np.pi * nmesh / box + 1e-6),
             dict(kbins=np.array([0.0, 0.02, 0.021, 0.3, 0.31, 2.0]), mubins=np.array([0.0, 0.05, 0.5, 0.51, 1.0]), poles=[2]),
             dict(kbins=24, mubins=8, poles=[], logk=True, k_max=2.0, w=w),
             # edges past k_Nyquist: the i = n/2 plane is binned with its phase folded to -n/2 (power_spectrum.py:940-942)
             dict(kbins=40, mubins=3, poles=[0, 2, 4], k_max=1.7 * np.pi * nmesh / box))
    for kw in (cases if nmesh == 1024 else cases[:2] + cases[4:]):
        kw = dict(kw, paste='TSC', nmesh=nmesh, compensated=comp, interlaced=True)
        _lib.profile_reset()
        _lib.profile_enable(True)
        a = calc_power(pos.copy(), box, **kw)
        _lib.profile_enable(False)
        prof = _lib.profile_get()
        assert 'fft_x_bin' in prof and 'spectrum_bin' not in prof and 'fft_cols_x' not in prof, sorted(prof)
        options.set('pk_noxbin_inter', 1)
        b = calc_power(pos.copy(), box, **kw)
        options.set('pk_noxbin_inter', 0)
        np.testing.assert_array_equal(a['N_mode'], b['N_mode'])
        scale = np.abs(np.asarray(b['power'])).max()
        np.testing.assert_allclose(a['power'], b['power'], rtol=3e-6, atol=3e-7 * scale)
        np.testing.assert_allclose(a['k_avg'], b['k_avg'], rtol=1e-6)
        if kw['poles']:
            np.testing.assert_allclose(a['poles'], b['poles'], rtol=3e-6, atol=5e-7 * scale)
    if nmesh == 1024:
        from oracle import oracle
        kw = dict(kbins=20, mubins=3, k_max=1.2, paste='TSC', nmesh=1024, compensated=comp, interlaced=True, poles=[0, 2])
        small = synth.synth_positions(300_000, box, seed=92, clustered=True)
        _check_oracle(calc_power(small.copy(), box, **kw), oracle.calc_power(small.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw))
        kw.update(k_max=1.7 * np.pi * nmesh / box, kbins=34)       # past Nyquist: every x-Nyquist mode is binned
        _check_oracle(calc_power(small.copy(), box, **kw), oracle.calc_power(small.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw))


@pytest.mark.parametrize('nmesh,comp', [(1024, True), (1024, False), (2048, True)])
def test_fused_last_pass_interlaced_cross_matches_spectrum_bin(options, nmesh, comp):
    """the four-field form of the fused last pass (fft_x_bin2<.., QUAD>): the cross power of two INTERLACED fields - the
    reference's defaults with a second catalogue, calc_power(pos, pos2=..., interlaced=True, compensated=True)
    (analysis/power_spectrum.py:1200-1260: get_interlaced_field_fft per catalogue, then get_raw_power's cross form :722-726) -
    against four complete transforms + spectrum_bin<INTER, CROSS> on the same particles (option pk_noxbin_cross) mode by mode,
    incl. edges past Nyquist (the folded i = n/2 plane), and against the oracle at 1024"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power
    box = 1000.0
    pos = synth.synth_positions(2_000_000, box, seed=291, clustered=True)
    pos2 = synth.synth_positions(1_200_000, box, seed=292, clustered=True)
    pos2[:400_000] = pos[:400_000]            # a shared population: the cross power is not just noise around zero
    w = np.random.default_rng(24).random(len(pos), dtype=np.float32) + np.float32(0.5)
    cases = (dict(kbins=64, mubins=4, poles=[0, 2, 4]), dict(kbins=300, mubins=None, poles=[0, 2], k_max=np.pi * nmesh / box + 1e-6),
             dict(kbins=24, mubins=8, poles=[], logk=True, k_max=2.0, w=w),
             dict(kbins=np.array([0.0, 0.02, 0.021, 0.3, 0.31, 2.0]), mubins=np.array([0.0, 0.05, 0.5, 0.51, 1.0]), poles=[2]),
             dict(kbins=40, mubins=3, poles=[0, 2, 4], k_max=1.7 * np.pi * nmesh / box))
    for kw in (cases if nmesh == 1024 else cases[:2] + cases[4:]):
        kw = dict(kw, paste='TSC', nmesh=nmesh, compensated=comp, interlaced=True)
        _lib.profile_reset()
        _lib.profile_enable(True)
        a = calc_power(pos.copy(), box, pos2=pos2.copy(), **kw)
        _lib.profile_enable(False)
        prof = _lib.profile_get()
        assert 'fft_x_bin' in prof and 'spectrum_bin' not in prof and 'fft_cols_x' not in prof, sorted(prof)
        options.set('pk_noxbin_cross', 1)
        b = calc_power(pos.copy(), box, pos2=pos2.copy(), **kw)
        options.set('pk_noxbin_cross', 0)
        np.testing.assert_array_equal(a['N_mode'], b['N_mode'])
        scale = np.abs(np.asarray(b['power'])).max()
        np.testing.assert_allclose(a['power'], b['power'], rtol=3e-6, atol=3e-7 * scale)
        np.testing.assert_allclose(a['k_avg'], b['k_avg'], rtol=1e-6)
        if kw['poles']:
            np.testing.assert_allclose(a['poles'], b['poles'], rtol=3e-6, atol=5e-7 * scale)
    if nmesh == 1024:
        from oracle import oracle
        kw = dict(kbins=20, mubins=3, k_max=1.2, paste='TSC', nmesh=1024, compensated=comp, interlaced=True, poles=[0, 2])
        s1 = synth.synth_positions(300_000, box, seed=293, clustered=True)
        s2 = synth.synth_positions(200_000, box, seed=294, clustered=True)
        s2[:80_000] = s1[:80_000]
        _check_oracle(calc_power(s1.copy(), box, pos2=s2.copy(), **kw),
                      oracle.calc_power(s1.copy(), box, pos2=s2.copy(), nthread=oracle.max_threads(), accum64=True, **kw))
        kw.update(k_max=1.7 * np.pi * nmesh / box, kbins=34)       # past Nyquist: every x-Nyquist mode is binned
        _check_oracle(calc_power(s1.copy(), box, pos2=s2.copy(), **kw),
                      oracle.calc_power(s1.copy(), box, pos2=s2.copy(), nthread=oracle.max_threads(), accum64=True, **kw))


def _random_edges(rng, nmesh, box):
    """k / mu edges of every flavour the geometry descriptor has to resolve or decline: linear, logarithmic, ragged, bins far
    narrower than a fundamental mode, first edge above zero, last edge short of / beyond Nyquist and beyond the corner"""
    kny = np.pi * nmesh / box
    kind = rng.integers(0, 5)
    nk = int(rng.integers(1, 700))
    kmax = kny * rng.choice([0.1, 0.5, 1.0, 1.0 + 1e-9, 1.3, 1.8])
    if kind == 0:
        ke = np.linspace(0.0, kmax, nk + 1)
    elif kind == 1:
        ke = np.geomspace(2 * np.pi / box * rng.choice([0.3, 0.9999, 1.7]), kmax, nk + 1)
    elif kind == 2:
        ke = np.sort(np.concatenate(([rng.random() * 0.02], rng.random(min(nk, 60)) * kmax)))
    elif kind == 3:
        ke = np.linspace(kmax * 0.2, kmax, min(nk, 40) + 1)
    else:
        ke = np.concatenate((np.linspace(0, 0.01 * kmax, 30), np.geomspace(0.011 * kmax, kmax, 20)))
    nmu = int(rng.integers(1, 9))
    nmu = max(1, min(nmu, 3800 // max(len(ke) - 1, 1)))          # the comparator (spectrum_bin) holds 20 B of LDS per bin
    me = np.linspace(0.0, 1.0, nmu + 1) if rng.random() < 0.6 else np.sort(np.concatenate(([0.0, 1.0], rng.random(nmu - 1))))
    poles = [[], [0], [0, 2], [0, 2, 4], [4], [2, 4]][int(rng.integers(0, 6))]
    return ke, me, poles


@pytest.mark.parametrize('interlaced,cross', [(False, False), (True, False), (True, True)])
def test_more_bins_than_one_histogram_holds(interlaced, cross):
    """the reference takes any number of (k, mu) bins (power_spectrum.py:150-300); a binning finer than the LDS histogram of
    one launch is done in several passes over runs of k bins.  k edges ON mode radii (|k|^2 integers in units of the
    fundamental, every fourth edge) make the edges shared by two passes matter: against the oracle, N_mode exact"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    nmesh, box = 64, 500.0
    kf = 2 * np.pi / box
    Nk, Nmu = 900, 7
    r2 = np.sort(np.random.default_rng(3).choice(np.arange(1, 3 * (nmesh // 2) ** 2), Nk + 1, replace=False)).astype(np.float64)
    ke = np.sqrt(r2) * kf
    ke[1::4] *= 1.0 + 1e-4                                                # the others between radii
    ke = np.sort(ke)
    me = np.linspace(0, 1, Nmu + 1)
    pos = synth.synth_positions(60_000, box, seed=12, clustered=True)
    extra = dict(pos2=synth.synth_positions(40_000, box, seed=13, clustered=True)) if cross else {}
    kw = dict(kbins=ke, mubins=me, poles=[0, 2, 4], paste='TSC', nmesh=nmesh, compensated=True, interlaced=interlaced)
    tab = calc_power(pos.copy(), box, **kw, **extra)
    ref = oracle.calc_power(pos.copy(), box, nthread=4, accum64=True, **kw, **{k: v.copy() for k, v in extra.items()})
    assert int(np.asarray(tab['N_mode']).sum()) > 0.5 * nmesh ** 3 / 2
    # bins of a handful of modes: Re(d1 d2*) of a cross spectrum cancels, its float32 products show at 1e-5 of the bin's own value
    _check_oracle(tab, ref, rtol=1e-4 if cross else RTOL)
    # and the same numbers as the coarse binning they refine: every 100 fine bins merged
    coarse = calc_power(pos.copy(), box, **dict(kw, kbins=ke[::100]), **extra)
    fine_n = np.asarray(tab['N_mode']).reshape(Nk // 100, 100, Nmu).sum(axis=1)
    np.testing.assert_array_equal(fine_n, coarse['N_mode'])


@pytest.mark.parametrize('seed', range(8))      # more seeds: scripts/gpu_xbin_fuzz.sh
def test_fused_last_pass_random_edges(options, seed):
    """the cached-geometry kernel (integer thresholds, cell table, per-kz mu thresholds, validated over every mode of the mesh
    by xbin_geometry) against the x pass + spectrum_bin on random bin edges: N_mode exact, sums to rounding; edges the table
    cannot resolve must fall back to the first-generation kernel and agree all the same"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power
    rng = np.random.default_rng(1000 + seed)
    nmesh, box = 1024, 1000.0
    pos = synth.synth_positions(400_000, box, seed=200 + seed, clustered=True)
    pos2 = np.concatenate((pos[:100_000], synth.synth_positions(150_000, box, seed=700 + seed, clustered=True)))
    for it in range(5):
        ke, me, poles = _random_edges(rng, nmesh, box)
        comp = bool(rng.integers(0, 2))
        inter = it in (2, 4)   # the interlaced pair through the same descriptor (edges past Nyquist: the folded i = n/2 plane)
        cross = it in (3, 4)   # the cross power of two fields through the two-tile schedule; 4: of two INTERLACED fields (four tiles)
        kw = dict(kbins=ke, mubins=me, poles=poles, paste='TSC', nmesh=nmesh, compensated=comp, interlaced=inter)
        if cross:
            kw['pos2'] = pos2.copy()
        a = calc_power(pos.copy(), box, **kw)
        gen = _lib.lib().abacus_power_xbin_generation()
        assert gen in (1, 2)
        off = 'pk_noxbin_cross' if cross else 'pk_noxbin_inter' if inter else 'pk_noxbin'
        options.set(off, 1)
        b = calc_power(pos.copy(), box, **kw)
        options.set(off, 0)
        np.testing.assert_array_equal(a['N_mode'], b['N_mode'], err_msg=f'gen {gen}')
        scale = np.abs(np.asarray(b['power'])).max()
        np.testing.assert_allclose(a['power'], b['power'], rtol=3e-6, atol=3e-7 * scale, err_msg=f'gen {gen}')
        np.testing.assert_allclose(a['k_avg'], b['k_avg'], rtol=1e-6)
        if poles:
            np.testing.assert_allclose(a['poles'], b['poles'], rtol=3e-6, atol=5e-7 * scale)


@pytest.mark.parametrize('n', [16, 21])
def test_exported_pieces_of_the_chain(n):
    """bin_kmu, get_raw_power, shift_field_fft, get_interlaced_field_fft as callables of their own with the reference's
    signatures (analysis/power_spectrum.py:150-300, 707-727, 904-998), against what the reference returned
    (tests/golden/power_exports.npz)"""
    from abacusutils_amd.analysis import power_spectrum as ps
    g = load_golden('power_exports')
    Lb = float(g['meta.L'])
    f1, f2 = g[f'n{n}.f1'], g[f'n{n}.f2']
    raw = ps.get_raw_power(f1)
    assert raw.dtype == np.float32 and raw.shape == f1.shape
    np.testing.assert_allclose(raw, g[f'n{n}.raw_auto'], rtol=1e-6)
    np.testing.assert_allclose(ps.get_raw_power(f1, f2), g[f'n{n}.raw_cross'], rtol=1e-5, atol=1e-6)
    res = ps.bin_kmu(n, Lb, g[f'n{n}.kedges'], g[f'n{n}.muedges'], g[f'n{n}.raw_auto'], poles=np.array([0, 2, 4]))
    for name, a in zip(('power', 'N_mode', 'poles', 'N_mode_poles', 'k_avg'), res):
        want = g[f'n{n}.kmu.{name}']
        assert a.shape == want.shape and a.dtype == want.dtype, name
        if name.startswith('N_'):
            np.testing.assert_array_equal(a, want)
        else:
            assert_spectrum_close(a, want, rtol=1e-5, err_msg=name)
    res = ps.bin_kmu(n, Lb, g[f'n{n}.redges'], np.array([0.0, 0.5, 1.0]), g[f'n{n}.xi'], poles=np.array([0, 2]), fourier=False)
    for name, a in zip(('power', 'N_mode', 'poles', 'N_mode_poles', 'k_avg'), res):
        want = g[f'n{n}.rmu.{name}']
        if name.startswith('N_'):
            np.testing.assert_array_equal(a, want)
        else:
            assert_spectrum_close(a, want, rtol=1e-5, err_msg=name)
    a = f1.copy()
    assert ps.shift_field_fft(a, f2, n, Lb, Lb / n) is None          # in place, like the reference
    assert np.abs(a - g[f'n{n}.shifted']).max() <= 2e-6 * np.abs(g[f'n{n}.shifted']).max()
    pos = synth.synth_positions(3000, Lb, seed=70 + n, clustered=True)
    for key, paste, w in ((f'n{n}.il_tsc', 'TSC', None), (f'n{n}.il_cic_w', 'CIC', g[f'n{n}.w'])):
        b = ps.get_interlaced_field_fft(pos.copy(), Lb, n, paste, w)
        assert b.dtype == np.complex64 and b.shape == g[key].shape
        assert np.abs(b - g[key]).max() <= 5e-6 * np.abs(g[key]).max(), key
    with pytest.raises(NotImplementedError):
        ps.get_raw_power(f1.astype(np.complex128))


@pytest.mark.parametrize('nmesh,npart', [(72, 40_000), (182, 100_000), (384, 400_000), (550, 2_500_000)])   # radices 3 | 7, 13 | 3 | 5, 11; 96 and 110: scripts/gpu_gfft_sizes.sh
def test_mixed_radix_meshes_take_the_native_transform(nmesh, npart, options):
    """meshes with factors 3, 5, 7, 11, 13 - the reference's own test mesh 72 (tests/test_power.py:33), compute_power's default
    num_cells = 550 (hod/abacus_hod.py:1347) - through the hand-written mixed-radix passes of csrc/gfft.hip (the default
    for every even size with factors up to 13; the float64 meshes take the same kernels in double):
    calc_power against the oracle (scipy's pocketfft = the reference's transform) at 1e-5, and the spectrum itself against
    the hipFFT path"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power, get_field_fft
    from oracle import oracle
    box = 1000.0
    pos = synth.synth_positions(npart, box, seed=nmesh, clustered=True)
    kw = dict(kbins=20, mubins=3, k_max=np.pi * nmesh / box, paste='TSC', nmesh=nmesh, compensated=True, interlaced=True, poles=[0, 2, 4])
    _lib.profile_reset()
    _lib.profile_enable(True)
    tab = calc_power(pos.copy(), box, **kw)
    _lib.profile_enable(False)
    prof = _lib.profile_get()
    assert any(k.startswith('gfft_') for k in prof) and 'hipfft_r2c' not in prof, sorted(prof)
    _check_oracle(tab, oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw))
    # the interlaced pair, too, is binned by the last pass straight from LDS (gfft_x_bin<INTER>: both fields' tiles side by side)
    # - against two x passes + spectrum_bin<INTER> (option pk_noxbin_inter) mode by mode
    assert 'gfft_x_bin' in prof and 'spectrum_bin' not in prof and 'gfft_cols_x' not in prof, sorted(prof)
    options.set('pk_noxbin_inter', 1)
    tab_u = calc_power(pos.copy(), box, **kw)
    options.set('pk_noxbin_inter', 0)
    np.testing.assert_array_equal(tab['N_mode'], tab_u['N_mode'])
    sc = np.abs(np.asarray(tab_u['power'])).max()
    np.testing.assert_allclose(tab['power'], tab_u['power'], rtol=3e-6, atol=3e-7 * sc)
    np.testing.assert_allclose(tab['poles'], tab_u['poles'], rtol=3e-6, atol=5e-7 * sc)
    kw_ny = dict(kw, k_max=1.6 * np.pi * nmesh / box, compensated=False)     # past Nyquist: the folded i = n/2 plane is binned
    _check_oracle(calc_power(pos.copy(), box, **kw_ny), oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw_ny))
    if nmesh <= 182:
        a = get_field_fft(pos.copy(), box, nmesh, 'TSC', None, None, False, False)
        options.set('fft_hipfft', 1)
        b = get_field_fft(pos.copy(), box, nmesh, 'TSC', None, None, False, False)
        options.set('fft_hipfft', 0)
        assert np.abs(a - b).max() <= 3e-6 * np.abs(b).max()
    # the auto power of one non-interlaced field: the last mixed-radix pass bins straight from LDS (gfft_x_bin) - against the
    # separate x pass + spectrum_bin mode by mode, and against the oracle
    for comp, kw2 in ((True, dict(kbins=20, mubins=3, poles=[0, 2, 4])), (False, dict(kbins=37, mubins=1, poles=[0, 2])),
                      (False, dict(kbins=np.array([0.0, 0.013, 0.05, 0.051, 0.2, 0.9]) * nmesh / 72.0, mubins=np.array([0.0, 0.3, 0.31, 1.0]), poles=[]))):
        kw2 = dict(kw2, k_max=np.pi * nmesh / box, paste='TSC', nmesh=nmesh, compensated=comp, interlaced=False)
        if not np.isscalar(kw2['kbins']):
            kw2.pop('k_max')
        _lib.profile_reset()
        _lib.profile_enable(True)
        fa = calc_power(pos.copy(), box, **kw2)
        _lib.profile_enable(False)
        prof = _lib.profile_get()
        assert 'gfft_x_bin' in prof and 'spectrum_bin' not in prof and 'gfft_cols_x' not in prof, sorted(prof)
        options.set('pk_noxbin', 1)
        fb = calc_power(pos.copy(), box, **kw2)
        options.set('pk_noxbin', 0)
        np.testing.assert_array_equal(fa['N_mode'], fb['N_mode'])
        scale = np.abs(np.asarray(fb['power'])).max()
        np.testing.assert_allclose(fa['power'], fb['power'], rtol=3e-6, atol=3e-7 * scale)
        np.testing.assert_allclose(fa['k_avg'], fb['k_avg'], rtol=1e-6)
        if kw2['poles']:
            np.testing.assert_allclose(fa['poles'], fb['poles'], rtol=3e-6, atol=5e-7 * scale)
        if np.isscalar(kw2['kbins']):
            _check_oracle(fa, oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw2))


@pytest.mark.parametrize('logk', [False, True])
def test_compute_power_default_mesh_550_interlaced_against_oracle(logk):
    """what AbacusHOD.compute_power runs when asked for the reference's default estimator (hod/abacus_hod.py:1338-1347 ->
    calc_power(..., nmesh=num_cells=550, paste='TSC', compensated=True, interlaced=True)): 3e6 clustered particles + a cross
    spectrum against 1e6 others on the 550^3 mixed-radix mesh, against the oracle (N_mode exact, P and multipoles 1e-5)"""
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    box, nmesh = 2000.0, 550
    pos = synth.synth_positions(3_000_000, box, seed=550, clustered=True)
    pos2 = synth.synth_positions(1_000_000, box, seed=551, clustered=True)
    kw = dict(kbins=40, mubins=5, k_max=0.6, logk=logk, paste='TSC', nmesh=nmesh, compensated=True, interlaced=True, poles=[0, 2, 4])
    # the multipoles near their zero crossings (l = 4 at a tenth of the monopole) carry the float32 round-off of both
    # transforms: observed 1.2e-5 of the floored value = 1.2e-6 of the spectrum's scale in one of 120 values
    _check_oracle(calc_power(pos.copy(), box, **kw), oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw),
                  poles_rtol_factor=2.0)
    kw['pos2'] = pos2
    _check_oracle(calc_power(pos.copy(), box, **kw), oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw),
                  poles_rtol_factor=2.0)


@pytest.mark.parametrize('n', [24, 30, 21])
def test_float64_meshes_against_reference_goldens(n):
    """dtype=np.float64 of get_field / get_field_fft / calc_power (analysis/power_spectrum.py:808, 1001, 1148): float64 mesh,
    normalisation and transform (csrc/gfft.hip in double precision) against what the shimmed reference returned
    (tests/golden/power_f64.npz; n = 21: an odd mesh, which the mixed-radix kernels leave to hipFFT's double-precision
    transform) - the mesh and the spectrum at 1e-12 of their largest value; the binned power is float32 in
    the reference whatever the mesh (bin_kmu is called with its default dtype, :787-789), so it is held to the usual 1e-5.
    The interlaced branch of the reference ignores dtype (:1048-1052)"""
    from abacusutils_amd.analysis import power_spectrum as ps
    g = load_golden('power_f64')
    Lb, N = float(g['meta.L']), int(g['meta.N'])
    pos = synth.synth_positions(N, Lb, seed=300, clustered=True)
    pos2 = synth.synth_positions(N // 2, Lb, seed=301, clustered=True)
    w = g['w']

    def close(a, want, tol=1e-12):
        assert a.dtype == want.dtype and a.shape == want.shape
        assert np.abs(a - want).max() <= tol * np.abs(want).max()

    # float32 positions: the cloud weights are float32 (analysis/tsc.py:400) and the SHIMMED reference evaluates d**2 through
    # NumPy's float32 power where Numba (and the device) multiply - one float32 ulp of a weight, 4e-8 of the mesh (measured)
    close(ps.get_field(pos.copy(), Lb, n, 'TSC', w, dtype=np.float64), g[f'n{n}.field_tsc'], 2e-7)
    close(ps.get_field(pos.astype(np.float64), Lb, n, 'CIC', None, dtype=np.float64), g[f'n{n}.field_cic_p8'])
    W = ps.get_W_compensated(Lb, n, 'TSC', False)
    close(ps.get_field_fft(pos.copy(), Lb, n, 'TSC', w, W, True, False, dtype=np.float64), g[f'n{n}.fft_tsc_comp'], 2e-7)
    close(ps.get_field_fft(pos.astype(np.float64), Lb, n, 'TSC', None, None, False, False, dtype=np.float64), g[f'n{n}.fft_tsc_p8'])
    il = ps.get_field_fft(pos.copy(), Lb, n, 'TSC', None, W, True, True, dtype=np.float64)
    assert str(il.dtype) == str(g[f'n{n}.fft_interlaced_dtype']) == 'complex64'
    for name, kw in (('auto', dict(compensated=True, interlaced=False, w=w)), ('cross', dict(compensated=False, interlaced=False, pos2=pos2.copy())),
                     ('interlaced', dict(compensated=True, interlaced=True))):
        tab = ps.calc_power(pos.copy(), Lb, kbins=10, mubins=3, k_max=np.pi * n / Lb + 1e-6, paste='TSC', nmesh=n, poles=[0, 2, 4],
                            dtype=np.float64, **kw)
        for c in ('N_mode', 'N_mode_poles'):
            np.testing.assert_array_equal(np.asarray(tab[c]), g[f'n{n}.{name}.{c}'])
        for c in ('power', 'poles', 'k_avg'):
            assert np.asarray(tab[c]).dtype == g[f'n{n}.{name}.{c}'].dtype == np.float32
            assert_spectrum_close(tab[c], g[f'n{n}.{name}.{c}'], rtol=1e-5, err_msg=f'{name}.{c}')
    with pytest.raises(TypeError):
        ps.calc_power(pos.copy(), Lb, nmesh=n, dtype=np.float16)


@pytest.mark.parametrize('interlaced', [False, True])
def test_host_positions_uploaded_in_batches_behind_the_deposits(interlaced, options):
    """calc_power on NumPy positions (the reference's call, analysis/power_spectrum.py:1131): where the mesh is small against
    the catalogue the upload runs in batches on a copy stream and every batch is deposited - accumulating into the mesh, the
    normalisation applied by the last flush - while the next is on the PCIe link (csrc/power.hip, HostSrc).  Same spectrum as
    the one-copy path (option pk_nobatch) and as the oracle; positions wrapped in place identically"""
    import ctypes as C
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from oracle import oracle
    rng = np.random.default_rng(17)
    box, nmesh, n = 1500.0, 512, 24_000_000
    pos = ((rng.random((n, 3), dtype='f4') * np.float32(1.02) - np.float32(0.01)) * np.float32(box)).astype('f4')   # some outside the box
    kw = dict(kbins=40, mubins=3, paste='TSC', nmesh=nmesh, compensated=interlaced, interlaced=interlaced, poles=[0, 2, 4])
    p1, p2, p3 = pos.copy(), pos.copy(), pos.copy()
    a = calc_power(p1, box, **kw)
    last = _lib.lib().abacus_power_last_batches
    last.restype = C.c_double
    assert last() >= 2, last()
    options.set('pk_nobatch', 1)
    b = calc_power(p2, box, **kw)
    assert last() == 1
    np.testing.assert_array_equal(p1, p2)
    assert not np.array_equal(p1, pos)                       # something was wrapped
    np.testing.assert_array_equal(np.asarray(a['N_mode']), np.asarray(b['N_mode']))
    scale = np.abs(np.asarray(b['power'])).max()
    np.testing.assert_allclose(a['power'], b['power'], rtol=2e-6, atol=2e-7 * scale)
    _check_oracle(a, oracle.calc_power(p3, box, nthread=oracle.max_threads(), accum64=True, **kw))
    np.testing.assert_array_equal(p1, p3)


def test_mixed_radix_1536_fused_last_pass(options):
    """1536^3 (the bench's `pk_1536` leg): the 1536-row tile, the float64 histogram of 512 k bins and the cell table only share the
    LDS in the lean form of gfft_x_bin (twiddles and mu thresholds read from global memory) - against the separate x pass +
    spectrum_bin mode by mode, with the bench's 2-dk bins and with coarse compensated bins"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power
    box, nmesh = 2000.0, 1536
    pos = synth.synth_positions(3_000_000, box, seed=1536, clustered=True)
    for kw in (dict(kbins=512, mubins=4, k_max=np.pi * nmesh / box + 1e-6, poles=[0, 2, 4], compensated=False),
               dict(kbins=48, mubins=3, poles=[0, 2], compensated=True)):
        kw = dict(kw, paste='TSC', nmesh=nmesh, interlaced=False)
        _lib.profile_reset()
        _lib.profile_enable(True)
        a = calc_power(pos.copy(), box, **kw)
        _lib.profile_enable(False)
        prof = _lib.profile_get()
        assert 'gfft_x_bin' in prof and 'spectrum_bin' not in prof and 'gfft_cols_x' not in prof, sorted(prof)
        options.set('pk_noxbin', 1)
        b = calc_power(pos.copy(), box, **kw)
        options.set('pk_noxbin', 0)
        np.testing.assert_array_equal(a['N_mode'], b['N_mode'])
        scale = np.abs(np.asarray(b['power'])).max()
        np.testing.assert_allclose(a['power'], b['power'], rtol=3e-6, atol=3e-7 * scale)
        np.testing.assert_allclose(a['k_avg'], b['k_avg'], rtol=1e-6)
        np.testing.assert_allclose(a['poles'], b['poles'], rtol=3e-6, atol=5e-7 * scale)


@pytest.mark.parametrize('nmesh,npart', [(550, 1_000_000), (768, 2_000_000), (1536, 3_000_000)])
def test_mixed_radix_compile_time_plans_equal_the_runtime_plan(options, nmesh, npart):
    """550 / 768 / 1536: the row and y passes run stage sequences with the mesh size, the stage strides and the tile pitch as
    compile-time constants (csrc/gfft.hip GFixed) - the same butterflies in the same order as the runtime plan
    (option gfft_nofixed), so the spectra agree to the last few bits (FMA contraction may differ between the two compilations)"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power
    box = 2000.0
    pos = synth.synth_positions(npart, box, seed=nmesh, clustered=True)
    kw = dict(kbins=40, mubins=4, poles=[0, 2, 4], compensated=True, paste='TSC', nmesh=nmesh)
    for interlaced in (False, True):
        _lib.profile_reset()
        _lib.profile_enable(True)
        a = calc_power(pos.copy(), box, interlaced=interlaced, **kw)
        _lib.profile_enable(False)
        assert 'gfft_rows' in _lib.profile_get(), sorted(_lib.profile_get())
        options.set('gfft_nofixed', 1)
        b = calc_power(pos.copy(), box, interlaced=interlaced, **kw)
        options.set('gfft_nofixed', 0)
        np.testing.assert_array_equal(a['N_mode'], b['N_mode'])
        scale = np.abs(np.asarray(b['power'])).max()
        np.testing.assert_allclose(a['power'], b['power'], rtol=2e-6, atol=2e-7 * scale)
        np.testing.assert_allclose(a['poles'], b['poles'], rtol=2e-6, atol=5e-7 * scale)
