"""HIP pair counting (csrc/pairs.hip) vs the brute-force float32 counter of the oracle (identical float32
expressions -> identical integer counts) and the wrapper arithmetic of tpcf_corrfunc.  Corrfunc itself is a
third-party dependency absent from the reference tree and untested there: parity unpinned beyond these checks.
Needs an MI355X: run with `-m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _points(n, box, seed, centered=False, clustered=True):
    rng = np.random.default_rng(seed)
    p = rng.random((n, 3)) * box
    if clustered:  # a few clumps so that small-r bins are populated
        k = n // 4
        centers = rng.random((20, 3)) * box
        p[:k] = (centers[rng.integers(0, 20, k)] + rng.normal(0, 2.0, (k, 3))) % box
    if centered:
        p -= box / 2
    return p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy()


@pytest.mark.parametrize('mode', ['r', 'rppi', 'smu'])
@pytest.mark.parametrize('auto', [True, False])
@pytest.mark.parametrize('centered', [False, True])
def test_vs_bruteforce(mode, auto, centered):
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    from oracle import oracle
    box = 200.0
    x1, y1, z1 = _points(3000, box, 1, centered)
    x2, y2, z2 = (None, None, None) if auto else _points(2500, box, 2, centered)
    bins = np.logspace(-1, np.log10(30.0), 14)
    kw = dict(pimax=30.0, npibins=30) if mode == 'rppi' else (dict(mu_max=1.0, nmubins=20) if mode == 'smu' else {})
    want = oracle.paircount_brute(mode, x1, y1, z1, box, bins, x2, y2, z2, **kw)
    if mode == 'r':
        got = T.DD(int(auto), 4, bins, x1, y1, z1, X2=x2, Y2=y2, Z2=z2, periodic=True, boxsize=box)['npairs']
    elif mode == 'rppi':
        got = T.DDrppi(int(auto), 4, binfile=bins, pimax=30.0, X1=x1, Y1=y1, Z1=z1, X2=x2, Y2=y2, Z2=z2,
                       periodic=True, boxsize=box)['npairs']
    else:
        got = T.DDsmu(int(auto), 4, bins, 1.0, 20, x1, y1, z1, X2=x2, Y2=y2, Z2=z2, periodic=True,
                      boxsize=box)['npairs']
    assert want.sum() > 0
    np.testing.assert_array_equal(got, want)
    if auto:
        assert np.all(got % 2 == 0)   # ordered pairs


@pytest.mark.parametrize('mode', ['r', 'rppi', 'smu'])
@pytest.mark.parametrize('auto', [True, False])
@pytest.mark.parametrize('frame', [0.0, -100.0, 37.3, -1234.5, 'unwrapped'])
def test_dense_half_stencil_any_frame(mode, auto, frame):
    """dense catalogue (67 points per r_max-cell: the wave-per-cell kernel with cells of r_max / 2, 125-cell stencil, half
    stencil for the autocorrelation) with the coordinates in [a, a + L) for several a - Corrfunc takes any range; the
    periodic image per pair of cells follows from the frame - and 'unwrapped' (coordinates spilling over both box edges:
    no single period holds them, every pair takes the per-pair minimum image) against the brute-force counter"""
    from abacusutils_amd.analysis.tpcf_corrfunc import _paircount
    from oracle import oracle
    box = 200.0
    x1, y1, z1 = _points(20000, box, 11)
    x2, y2, z2 = (None, None, None) if auto else _points(15000, box, 12)
    if frame == 'unwrapped':
        sh = lambda v, s: v + np.where(np.random.default_rng(s).random(len(v)) < 0.1, box, 0.0) - 20.0   # noqa: E731
        x1, y1, z1 = sh(x1, 1), sh(y1, 2), sh(z1, 3)
        if not auto:
            x2, y2, z2 = sh(x2, 4), sh(y2, 5), sh(z2, 6)
    else:
        x1, y1, z1 = x1 + frame, y1 + frame, z1 + frame
        if not auto:
            x2, y2, z2 = x2 + frame, y2 + frame, z2 + frame
    bins = np.logspace(-1, np.log10(30.0), 14)
    kw = dict(pimax=30.0, npibins=30) if mode == 'rppi' else (dict(mu_max=1.0, nmubins=20) if mode == 'smu' else {})
    want = oracle.paircount_brute(mode, x1, y1, z1, box, bins, x2, y2, z2, nthread=oracle.max_threads(), **kw)
    got = _paircount({'r': 0, 'rppi': 1, 'smu': 2}[mode], x1, y1, z1, box, bins, x2, y2, z2, **kw)
    assert want.sum() > 0
    np.testing.assert_array_equal(got, want.ravel())


def test_device_resident_columns_match_host_call():
    """abacus_paircount_dev on float64 / float32 columns already in HBM (the HOD catalogue's layout and frame) == the
    host-array call on the same values"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.tpcf_corrfunc import _paircount
    box = 500.0
    rng = np.random.default_rng(3)
    p1 = (rng.random((40000, 3)) - 0.5) * box          # float64 in [-L/2, L/2) like the galaxy columns
    p2 = (rng.random((30000, 3)) - 0.5) * box
    bins = np.linspace(0.5, 20.0, 11)
    for dt in (np.float64, np.float32):
        d1 = [_lib.DeviceArray(np.ascontiguousarray(p1[:, i], dtype=dt)) for i in range(3)]
        d2 = [_lib.DeviceArray(np.ascontiguousarray(p2[:, i], dtype=dt)) for i in range(3)]
        for mode, kw in ((0, {}), (1, dict(pimax=20.0, npibins=20)), (2, dict(mu_max=1.0, nmubins=10))):
            want = _paircount(mode, p1[:, 0], p1[:, 1], p1[:, 2], box, bins, **kw)
            got = _paircount(mode, *d1, box, bins, **kw)
            np.testing.assert_array_equal(got, want)
            wantx = _paircount(mode, p1[:, 0], p1[:, 1], p1[:, 2], box, bins, p2[:, 0], p2[:, 1], p2[:, 2], **kw)
            gotx = _paircount(mode, *d1, box, bins, *d2, **kw)
            np.testing.assert_array_equal(gotx, wantx)
        for a in d1 + d2:
            a.free()


def test_large_reach_few_cells():
    """r_max close to L/2: fewer than 3 cells per dimension, every cell is its own neighbour"""
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    from oracle import oracle
    box = 50.0
    x, y, z = _points(1500, box, 5, clustered=False)
    bins = np.linspace(0.5, 24.0, 8)
    want = oracle.paircount_brute('r', x, y, z, box, bins)
    got = T.DD(1, 1, bins, x, y, z, periodic=True, boxsize=box)['npairs']
    np.testing.assert_array_equal(got, want)


def test_uniform_known_answer_and_wrappers():
    """uniform randoms: DD/RR - 1 ~ 0 for xi(rp,pi), wp ~ 0; wrapper shapes and error checks
    (tpcf_corrfunc.py:112-121,183-203,304-305,365-372)"""
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    box, n = 500.0, 200000
    rng = np.random.default_rng(7)
    x, y, z = (rng.random(n) * box for _ in range(3))
    rpbins = np.logspace(0.0, 1.4771212597864314, 7)
    xi = T.calc_xirppi_fast(x, y, z, rpbins, 30, 5, box, 8)
    assert xi.shape == (6, 6)
    assert np.abs(xi[2:]).max() < 0.05
    wp = T.calc_wp_fast(x, y, z, rpbins, 30, box, 8)
    assert wp.shape == (6,) and np.abs(wp[2:]).max() < 1.0
    xil = T.calc_multipole_fast(x, y, z, rpbins, box, 8, nbins_mu=10, orders=[0, 2])
    assert xil.shape == (12,) and np.abs(xil[2:6]).max() < 0.05
    with pytest.raises(ValueError):
        T.calc_xirppi_fast(x, y, z, rpbins, 30.0, 5, box, 8)
    with pytest.raises(ValueError):
        T.calc_xirppi_fast(x, y, z, rpbins, 30, 7, box, 8)
    with pytest.raises(ValueError):
        T.calc_wp_fast(x, y, z, rpbins, 30.5, box, 8)


def test_cross_symmetry_and_scale():
    """DD(A,B) == DD(B,A); 1e6 x 3e5 points run through the cell list"""
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    box = 1000.0
    a = _points(300000, box, 11, centered=True)
    b = _points(1000000, box, 12, centered=True)
    bins = np.logspace(-1, np.log10(30.0), 14)
    ab = T.DD(0, 1, bins, *a, X2=b[0], Y2=b[1], Z2=b[2], periodic=True, boxsize=box)['npairs']
    ba = T.DD(0, 1, bins, *b, X2=a[0], Y2=a[1], Z2=a[2], periodic=True, boxsize=box)['npairs']
    np.testing.assert_array_equal(ab, ba)
    assert ab.sum() > 0


def test_full_size_c5_properties(options):
    """BASELINE config 5 size (1e7 points, 2 Gpc/h box, 13 log bins to 30 Mpc/h): the persistent kernel and the
    first-generation one-workgroup-per-cell kernel count exactly the same pairs, and uniform randoms give the analytic
    expectation N (N-1) V_shell / V within 5 sigma of the Poisson error in every bin"""
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    n, box = 10_000_000, 2000.0
    rng = np.random.default_rng(500)
    p = rng.random((3, n), dtype=np.float32) * np.float32(box)
    bins = np.geomspace(0.1, 30.0, 14)
    got = T.DD(1, 1, bins, p[0], p[1], p[2], periodic=True, boxsize=box)['npairs']
    options.set('pairs_gen', 1)
    old = T.DD(1, 1, bins, p[0], p[1], p[2], periodic=True, boxsize=box)['npairs']
    np.testing.assert_array_equal(got, old)
    b32 = bins.astype(np.float32).astype(np.float64)
    expect = float(n) * (n - 1) * 4 / 3 * np.pi * (b32[1:] ** 3 - b32[:-1] ** 3) / box**3
    # ordered pairs: each unordered pair is counted twice, so the variance is 2 * expect
    assert np.all(np.abs(got - expect) < 5 * np.sqrt(2 * expect) + 2), (got, expect)


@pytest.mark.parametrize('seed', range(18))
def test_random_configuration_sweep(seed):
    """seeded random pair-count calls (mode, auto / cross, point counts from a handful to 6000, box, reach up to L/2,
    bin layout, pimax / mu bins, coordinates inside [0, L) or centred or partly outside) against the brute-force counter"""
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    from oracle import oracle
    rng = np.random.default_rng(7000 + seed)
    mode = ['r', 'rppi', 'smu'][seed % 3]
    box = float(rng.choice([50.0, 200.0, 1000.0]))
    n1, n2 = int(rng.integers(5, 6000)), int(rng.integers(5, 5000))
    auto = bool(rng.integers(2))
    place = seed % 4                       # 0: [0, L), 1: centred, 2: a few points outside the box, 3: clustered in a corner
    x1, y1, z1 = _points(n1, box, 100 + seed, centered=place == 1, clustered=bool(rng.integers(2)))
    x2, y2, z2 = (None, None, None) if auto else _points(n2, box, 200 + seed, centered=place == 1, clustered=True)
    if place == 2:
        x1[:3] += box
        z1[-2:] -= box
    if place == 3:
        x1, y1, z1 = x1 * 0.05, y1 * 0.05, z1 * 0.05
    rmax = float(box * rng.uniform(0.02, 0.49))
    nb = int(rng.integers(1, 20))
    bins = np.geomspace(rmax * 1e-3, rmax, nb + 1) if rng.integers(2) else np.linspace(0.0, rmax, nb + 1)
    if mode == 'rppi':
        pimax = float(int(min(rmax, 40.0)) or 1)
        kw = dict(pimax=pimax, npibins=int(pimax))
    elif mode == 'smu':
        kw = dict(mu_max=float(rng.choice([1.0, 0.7])), nmubins=int(rng.integers(1, 30)))
    else:
        kw = {}
    want = oracle.paircount_brute(mode, x1, y1, z1, box, bins, x2, y2, z2, **kw)
    if mode == 'r':
        got = T.DD(int(auto), 4, bins, x1, y1, z1, X2=x2, Y2=y2, Z2=z2, periodic=True, boxsize=box)['npairs']
    elif mode == 'rppi':
        got = T.DDrppi(int(auto), 4, binfile=bins, pimax=kw['pimax'], X1=x1, Y1=y1, Z1=z1, X2=x2, Y2=y2, Z2=z2,
                       periodic=True, boxsize=box)['npairs']
    else:
        got = T.DDsmu(int(auto), 4, bins, kw['mu_max'], kw['nmubins'], x1, y1, z1, X2=x2, Y2=y2, Z2=z2, periodic=True,
                      boxsize=box)['npairs']
    np.testing.assert_array_equal(got, want)


def test_wrappers_vs_reference_golden():
    """calc_xirppi_fast / calc_wp_fast / calc_multipole_fast on the HIP counters against the outputs of the REFERENCE's
    wrappers (tpcf_corrfunc.py:97-372, run under the shim with a brute-force Corrfunc stand-in): bit for bit"""
    from test_oracle_pinned import check_pair_wrappers

    from abacusutils_amd.analysis import tpcf_corrfunc as T
    check_pair_wrappers(T)


@pytest.mark.parametrize('mode,auto', [('r', True), ('rppi', False), ('smu', True)])
def test_more_bins_than_one_launch_holds(mode, auto):
    """Corrfunc takes any number of bins; one launch bins into at most 63 separation bins and 8192 LDS counters, more are
    counted in runs of consecutive separation bins.  150 linear bins (x 60 pi bins = 9000, x 120 mu bins = 18000) against
    the brute-force counter; integer-spaced points put pairs ON the edges shared by two runs"""
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    from oracle import oracle
    box = 240.0
    rng = np.random.default_rng(8)
    p1 = np.round(rng.random((2500, 3)) * box * 2) / 2                      # a half-integer lattice: separations hit the edges
    p1 = (p1 % box).astype(np.float32)
    x1, y1, z1 = (np.ascontiguousarray(p1[:, i]) for i in range(3))
    x2, y2, z2 = (None, None, None) if auto else _points(2000, box, 4)
    bins = (np.arange(151) * 0.25 + 0.5).astype(np.float32)                 # exact in float32: 0.5, 0.75, ..., 38
    kw = dict(pimax=60.0, npibins=60) if mode == 'rppi' else (dict(mu_max=1.0, nmubins=120) if mode == 'smu' else {})
    want = oracle.paircount_brute(mode, x1, y1, z1, box, bins, x2, y2, z2, nthread=oracle.max_threads(), **kw)
    if mode == 'r':
        got = T.DD(int(auto), 4, bins, x1, y1, z1, X2=x2, Y2=y2, Z2=z2, periodic=True, boxsize=box)['npairs']
    elif mode == 'rppi':
        got = T.DDrppi(int(auto), 4, binfile=bins, pimax=60.0, X1=x1, Y1=y1, Z1=z1, X2=x2, Y2=y2, Z2=z2, periodic=True,
                       boxsize=box)['npairs']
    else:
        got = T.DDsmu(int(auto), 4, bins, 1.0, 120, x1, y1, z1, X2=x2, Y2=y2, Z2=z2, periodic=True, boxsize=box)['npairs']
    assert want.sum() > 1000
    np.testing.assert_array_equal(got, want)


def test_bin_edge_conventions():
    """pairs that sit exactly ON an edge document the convention (Corrfunc's published kernels, restated from memory -
    the library is absent): r-bin b holds edges[b] <= r < edges[b+1]; DDrppi keeps |dz| < pimax with
    pi-bin = int(|dz| * npibins / pimax); DDsmu drops mu >= mu_max (a pair along the line of sight has mu = 1)."""
    from abacusutils_amd.analysis import tpcf_corrfunc as T
    from oracle import oracle
    box = 64.0
    bins = np.array([1.0, 2.0, 4.0, 8.0], dtype=np.float32)
    base = np.array([10.0, 20.0, 30.0], dtype=np.float32)
    # partner offsets (all float32-exact): r = 2 (on an inner edge, along x), r = 1 (first edge), r = 8 (last edge),
    # r = 4 along z (mu = 1, |dz| = 4), and (3, 0, 4): rp = 3, |dz| = 4, s = 5, mu = 0.8
    offs = np.array([[2, 0, 0], [0, 1, 0], [8, 0, 0], [0, 0, 4], [3, 0, 4]], dtype=np.float32)
    far = np.array([[40.0, 40.0, 40.0]], dtype=np.float32)   # keeps the sets non-degenerate
    p2 = np.concatenate([base + offs, far])
    x1, y1, z1 = (np.array([v], dtype=np.float32) for v in base)
    x2, y2, z2 = (np.ascontiguousarray(p2[:, i]) for i in range(3))
    dd = T.DD(0, 1, bins, x1, y1, z1, X2=x2, Y2=y2, Z2=z2, periodic=True, boxsize=box)['npairs']
    np.testing.assert_array_equal(dd, oracle.paircount_brute('r', x1, y1, z1, box, bins, x2, y2, z2))
    # r=1 -> bin 0 (lower edge included); r=2 -> bin 1; r=4 -> bin 2; r=5 -> bin 2; r=8 -> excluded (upper edge open)
    np.testing.assert_array_equal(dd, [1, 1, 2])
    rppi = T.DDrppi(0, 1, binfile=bins, pimax=4.0, X1=x1, Y1=y1, Z1=z1, X2=x2, Y2=y2, Z2=z2, periodic=True,
                    boxsize=box)['npairs'].reshape(3, 4)
    np.testing.assert_array_equal(rppi.ravel(), oracle.paircount_brute('rppi', x1, y1, z1, box, bins, x2, y2, z2,
                                                                       pimax=4.0, npibins=4))
    # |dz| = 4 = pimax is dropped (both the z pair and (3,0,4)); (2,0,0) -> rp bin 1, pi bin 0; (0,1,0) -> rp bin 0
    want = np.zeros((3, 4), dtype=np.uint64)
    want[1, 0] = 1
    want[0, 0] = 1
    np.testing.assert_array_equal(rppi, want)
    smu = T.DDsmu(0, 1, bins, 1.0, 5, x1, y1, z1, X2=x2, Y2=y2, Z2=z2, periodic=True, boxsize=box)['npairs'].reshape(3, 5)
    np.testing.assert_array_equal(smu.ravel(), oracle.paircount_brute('smu', x1, y1, z1, box, bins, x2, y2, z2,
                                                                      mu_max=1.0, nmubins=5))
    want = np.zeros((3, 5), dtype=np.uint64)
    want[1, 0] = 1      # (2,0,0): s = 2, mu = 0
    want[0, 0] = 1      # (0,1,0): s = 1, mu = 0
    want[2, 4] = 1      # (3,0,4): s = 5, mu = 0.8 -> int(0.8 * 5) = 4 (float32: 0.8f * 5 = 4.0000001 -> 4)
    np.testing.assert_array_equal(smu, want)   # the z pair (mu = 1 = mu_max) is dropped
