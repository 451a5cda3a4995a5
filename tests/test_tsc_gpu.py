"""HIP TSC / CIC / partition path vs the reference's saved grids, the golden vectors and the CPU oracle.
Mirrors the reference's tests/test_tsc.py.  Needs an MI355X: run with `-m gpu`."""
import numpy as np
import pytest
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _dense(g, key, ng):
    a = np.zeros(ng**3, dtype=np.float32)
    a[g[key + '.idx']] = g[key + '.val']
    return a.reshape(ng, ng, ng)


@pytest.mark.parametrize('ngrid', [10, 256])
@pytest.mark.parametrize('dtype', ['f4', 'f8'])
@pytest.mark.parametrize('nthread', [1, -1])
@pytest.mark.filterwarnings('ignore:.*dtype')
class TestTSC:
    box = 123.0

    def test_single(self, ngrid, dtype, nthread):
        """tests/test_tsc.py:25-90"""
        from abacusutils_amd.analysis.tsc import tsc_parallel
        cen = np.array([5, 6, 7])
        single = (cen / ngrid * self.box).astype(dtype).reshape(1, -1)
        dens = tsc_parallel(single, ngrid, self.box, nthread=nthread)
        assert (dens == 0).sum() == ngrid**3 - 27
        assert np.isclose(dens.sum(), 1.0)
        cube = dens[4:7, 5:8, 6:9]
        assert np.allclose([cube[0, 0, 0], cube[0, 0, 2], cube[0, 2, 0], cube[0, 2, 2], cube[2, 0, 0], cube[2, 0, 2],
                            cube[2, 2, 0], cube[2, 2, 2]], 0.5**9)
        assert np.allclose([cube[0, 0, 1], cube[0, 1, 0], cube[1, 0, 0], cube[0, 2, 1], cube[0, 1, 2], cube[1, 0, 2],
                            cube[2, 0, 1], cube[2, 1, 0], cube[1, 2, 0], cube[2, 2, 1], cube[2, 1, 2], cube[1, 2, 2]],
                           0.5**6 * 0.75)
        assert np.allclose([cube[1, 1, 0], cube[1, 0, 1], cube[0, 1, 1], cube[1, 1, 2], cube[1, 2, 1], cube[2, 1, 1]],
                           0.5**3 * 0.75**2)
        assert np.allclose(cube[1, 1, 1], 0.75**3)

    def test_multi(self, ngrid, dtype, nthread):
        """tests/test_tsc.py:92-159: mass conservation, saved reference grid, nbodykit grid (rtol 1e-4, atol 1e-5)"""
        from abacusutils_amd.analysis.tsc import tsc_parallel
        from oracle import oracle
        rng = np.random.default_rng(234)
        pos = rng.random((10000, 3), dtype='f4').astype(dtype) * self.box
        weights = rng.random((10000,), dtype='f4').astype(dtype)
        dens = tsc_parallel(pos, ngrid, self.box, nthread=nthread, weights=weights)
        assert np.isclose(dens.sum(dtype='f8'), weights.sum(dtype='f8'))
        pydens = np.zeros((ngrid, ngrid, ngrid), dtype=np.float32)
        oracle.tsc_scatter(pos, pydens, self.box, weights)
        assert np.allclose(dens, pydens)
        ref = load_golden('tsc_ref')
        assert np.allclose(dens, _dense(ref, f'tsc_ngrid{ngrid}', ngrid), rtol=1e-4, atol=1e-5)
        assert np.allclose(dens, _dense(ref, f'nbodykit_tsc_ngrid{ngrid}', ngrid), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('name,dtype', [('f4_w', 'f4'), ('f4_now', 'f4'), ('f8_w', 'f8'), ('f4_offset', 'f4'),
                                        ('f4_aniso', 'f4'), ('f8_grid64', 'f8')])
@pytest.mark.filterwarnings('ignore:.*dtype')
def test_scatter_cases(name, dtype):
    """golden vectors of _tsc_scatter: offset, anisotropic mesh, float64 mesh, no weights"""
    from abacusutils_amd.analysis.tsc import tsc_parallel
    g = load_golden('tsc_cases')
    box = float(g[name + '.box'])
    pos = (g['base'].astype(dtype) * box).astype(dtype)
    w = g['wts'].astype(dtype) if name not in ('f4_now', 'f8_grid64') else None
    want = g[name + '.grid']
    dens = np.zeros(want.shape, dtype=want.dtype)
    r = tsc_parallel(pos, dens, box, weights=w, offset=float(g[name + '.offset']), wrap=False)
    assert r is None
    np.testing.assert_allclose(dens, want, rtol=2e-5 if want.dtype == np.float32 else 1e-12, atol=1e-6)


def test_wrap_in_place_and_accumulate():
    """tsc.py:45-50,171-173: pos wrapped IN PLACE, the user grid is accumulated into, returns None"""
    from abacusutils_amd.analysis.tsc import tsc_parallel
    g = load_golden('tsc_cases')
    pos = g['parallel_wrap.pos_in'].copy()
    grid = np.full((12, 12, 12), 0.25, dtype=np.float32)
    assert tsc_parallel(pos, grid, 50.0, weights=g['wts']) is None
    np.testing.assert_array_equal(pos, g['parallel_wrap.pos_out'])
    np.testing.assert_allclose(grid, g['parallel_wrap.grid'], rtol=1e-5, atol=1e-6)


def test_returns(seed=123):
    """tests/test_tsc.py:211-230"""
    from abacusutils_amd.analysis.tsc import tsc_parallel
    rng = np.random.default_rng(seed)
    pos = rng.random((100, 3), dtype='f4') * 123.0
    dens = tsc_parallel(pos, 10, 123.0)
    assert dens.shape == (10, 10, 10)
    dens_allocated = np.zeros((10, 10, 10), dtype=np.float32)
    assert tsc_parallel(pos, dens_allocated, 123.0) is None
    np.testing.assert_allclose(dens_allocated, dens)


def test_bad_npartition():
    from abacusutils_amd.analysis.tsc import tsc_parallel
    with pytest.raises(ValueError):
        tsc_parallel(np.zeros((10, 3), dtype='f4'), 12, 1.0, nthread=4, npartition=5)
    with pytest.raises(ValueError):
        tsc_parallel(np.zeros((10, 3), dtype='f4'), 30, 1.0, nthread=4, npartition=3)


@pytest.mark.parametrize('seed', [123, 456])
@pytest.mark.parametrize('dtype', ['f4', 'f8'])
@pytest.mark.parametrize('npartition', [1, 1000])
def test_partition(seed, dtype, npartition):
    """tests/test_tsc.py:162-208 (and stronger: the stable order itself)"""
    from abacusutils_amd.analysis.tsc import partition_parallel
    rng = np.random.default_rng(seed)
    box, N = 123.0, 10000
    pos = rng.random((N, 3), dtype=dtype) * box
    weights = rng.random((N,), dtype=dtype)
    ppart, starts, wpart = partition_parallel(pos, npartition, box, weights=weights)
    keys = (pos[:, 0] * (npartition / box)).astype(np.int32)
    iord = keys.argsort(kind='stable')
    np_starts = np.zeros(npartition + 1, dtype=np.int64)
    np_starts[1:] = np.bincount(keys, minlength=npartition).cumsum()
    np.testing.assert_array_equal(starts, np_starts)
    np.testing.assert_array_equal(ppart, pos[iord])
    np.testing.assert_array_equal(wpart, weights[iord])


def test_partition_golden():
    from abacusutils_amd.analysis.tsc import partition_parallel
    g = load_golden('tsc_cases')
    pos = (g['base'] * np.float32(50.0)).astype('f4')
    ps, st, ws = partition_parallel(pos, 7, 50.0, weights=g['wts'])
    np.testing.assert_array_equal(st, g['partition.starts'])
    np.testing.assert_array_equal(ps, g['partition.psort'])
    np.testing.assert_array_equal(ws, g['partition.wsort'])


def test_cic_golden():
    from abacusutils_amd.analysis.cic import cic_serial
    g = load_golden('tsc_cases')
    pos = (g['base'] * np.float32(50.0)).astype('f4')
    dens = np.zeros((12, 12, 12), dtype=np.float32)
    cic_serial(pos, dens, 50.0, weights=g['wts'])
    assert np.allclose(dens, g['cic_f4_w.grid'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('n,ngrid', [(0, 16), (1, 3), (5, 2), (200000, 77), (3000000, 200), (2500000, 100)])
def test_vs_oracle_sizes(n, ngrid):
    """empty input, meshes smaller than a tile / than the cloud, non-multiple-of-tile meshes; the >= 2e6-particle
    cases go through the two-level multisplit (with and without a fine pass)"""
    from abacusutils_amd.analysis.tsc import tsc_parallel
    from oracle import oracle
    rng = np.random.default_rng(n + ngrid)
    box = 100.0
    pos = ((rng.random((n, 3), dtype='f4') * 1.2 - 0.1) * np.float32(box)).astype('f4')   # some outside the box
    w = rng.random(n, dtype='f4')
    p1, p2 = pos.copy(), pos.copy()
    a = tsc_parallel(p1, ngrid, box, weights=w)
    b = oracle.tsc_parallel(p2, ngrid, box, weights=w, nthread=1)
    np.testing.assert_array_equal(p1, p2)   # wrapped identically
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6)
    assert np.isclose(a.sum(dtype='f8'), w.sum(dtype='f8'), rtol=1e-6)


@pytest.mark.parametrize('dt', ['f4', 'f8'])
def test_multisplit_tile_corner_pile_up(dt, options):
    """every particle sits on a corner shared by eight 16 x 16 x 32-cell tiles, so each is listed eight times: a
    sub-chunk of the LDS-sorted scatter passes then holds five times the entries its buffer has room for (the overflow
    goes straight to its place in HBM).  Against the oracle, and against the single-level list builder"""
    from abacusutils_amd.analysis.tsc import tsc_parallel
    from oracle import oracle
    n, ng, box = 2_200_000, 256, 256.0
    rng = np.random.default_rng(77)
    corner = np.stack([rng.integers(0, ng // 16, n) * 16, rng.integers(0, ng // 16, n) * 16, rng.integers(0, ng // 32, n) * 32], axis=1)
    pos = ((corner + rng.uniform(-0.4, 0.4, (n, 3))) % ng).astype(dt)      # cell size 1: within half a cell of the corner
    w = rng.uniform(0.5, 1.5, n).astype(dt)
    p1, p2 = pos.copy(), pos.copy()
    a = tsc_parallel(p1, ng, box, weights=w)
    b = oracle.tsc_parallel(p2, ng, box, weights=w, nthread=1)
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * float(b.max()))
    assert np.isclose(a.sum(dtype='f8'), w.sum(dtype='f8'), rtol=1e-6)
    options.set('tsc_atomic', 1)
    c = tsc_parallel(pos.copy(), ng, box, weights=w)
    assert np.abs(a - c).max() <= 4e-6 * float(a.max())


def test_full_size_properties(options):
    """BASELINE config 3 size (1e8 particles, 1024^3 mesh): mass conservation (sum of the mesh = sum of the weights),
    every cell non-negative, and the two list builders (two-level multisplit vs single-level lists with per-tile atomic
    cursors, option `tsc_atomic`) feeding the tile deposit give the same mesh to float32 rounding of the cell sums"""
    from abacusutils_amd.analysis.tsc import tsc_parallel
    n, box, ng = 100_000_000, 2000.0, 1024
    rng = np.random.default_rng(300)
    pos = rng.random((n, 3), dtype=np.float32) * np.float32(box)
    w = (0.5 + rng.random(n, dtype=np.float32))
    a = tsc_parallel(pos, ng, box, weights=w)
    assert a.shape == (ng, ng, ng) and a.dtype == np.float32
    tot = float(a.sum(dtype=np.float64))
    assert abs(tot / float(w.sum(dtype=np.float64)) - 1) < 1e-6
    assert a.min() >= 0
    options.set('tsc_atomic', 1)
    b = tsc_parallel(pos, ng, box, weights=w)
    # mean cell holds ~0.09 particles' weight spread over 27 cells: float32 sums of a few terms
    assert np.abs(a - b).max() <= 4e-6 * max(a.max(), 1.0)


@pytest.mark.parametrize('seed', range(16))
@pytest.mark.filterwarnings('ignore:.*dtype')
def test_random_option_sweep(seed):
    """seeded random tsc_parallel calls (float32 / float64 positions, weights or none, offset, wrap, user mesh vs
    allocated, cubic and anisotropic meshes from 2 cells up, particle counts around the list-builder switch) against the
    oracle's stripe-partitioned scatter"""
    from abacusutils_amd.analysis.tsc import tsc_parallel
    from oracle import oracle
    rng = np.random.default_rng(9000 + seed)
    dt = ['f4', 'f8'][seed % 2]
    box = float(rng.choice([1.0, 123.0, 2000.0]))
    n = int(rng.choice([0, 1, 7, 3000, 50000, 400000, 2100000]))
    if rng.random() < 0.35:
        shape = tuple(int(v) for v in rng.integers(2, 70, 3))            # anisotropic
    else:
        shape = (int(rng.choice([2, 3, 5, 16, 31, 32, 33, 64, 100])),) * 3
    pos = ((rng.random((n, 3)) * 1.3 - 0.15) * box).astype(dt)            # some outside the box
    w = rng.uniform(0.2, 2.0, n).astype(dt) if rng.random() < 0.6 else None
    offset = float(rng.choice([0.0, 0.5 * box / shape[0], -0.3 * box / shape[0], 0.123]))
    wrap = bool(rng.integers(2)) or True    # positions outside the box need the wrap (tsc.py:45-50)
    grid0 = (rng.random(shape) * 0.1).astype(np.float32)
    ga, gb, p1, p2 = grid0.copy(), grid0.copy(), pos.copy(), pos.copy()
    assert tsc_parallel(p1, ga, box, weights=w, wrap=wrap, offset=offset) is None
    oracle.tsc_parallel(p2, gb, box, weights=w, nthread=1, wrap=wrap, offset=offset)
    np.testing.assert_array_equal(p1, p2)
    scale = max(float(np.abs(gb).max()), 1e-30)
    # the reference accumulates in the grid's float32; with 10^5 deposits per cell (2^3 ... 5^3 cells, 2e6 particles) that sum
    # itself is off by up to a per cent.  The device accumulates in float64 (deviation 1): it is held
    # tightly to the same scatter into a float64 grid, and to the float32 one within that scatter's own accumulation error
    gc, p3 = grid0.astype(np.float64), pos.copy()
    oracle.wrap_inplace(p3, box)
    oracle.tsc_scatter(p3, gc, box, weights=w, offset=offset)
    np.testing.assert_allclose(ga, gc, rtol=2e-6, atol=3e-6 * scale)
    np.testing.assert_allclose(ga, gb, rtol=2e-5, atol=3e-6 * scale + 2.0 * float(np.abs(gb - gc).max()))


def _line_case(kind, shape, n, box, rng):
    if kind == 'corners':   # clouds piled up on corners shared by eight tiles AND eight blocks of tiles
        corner = np.stack([rng.integers(0, shape[0] // 128, n) * 128, rng.integers(0, shape[1] // 128, n) * 128,
                           rng.integers(0, shape[2] // 256, n) * 256], axis=1)
        cells = (corner + rng.uniform(-0.45, 0.45, (n, 3))) % np.array(shape)
        return (cells * (box / np.array(shape))).astype('f4')
    if kind == 'blob':      # nine tenths of the catalogue inside a few cells (one tile's list holds millions of entries, one block
        # nearly every record), the rest uniform
        pos = (rng.random((n, 3), dtype='f4') * np.float32(box)).astype('f4')
        m = n - n // 10
        centre = np.array([0.37, 0.52, 0.81]) * box
        pos[:m] = (centre + rng.normal(0.0, 1.5 * box / shape[0], (m, 3))).astype('f4')
        return pos
    if kind == 'slab':      # catalogue order: sorted along x (what a halo catalogue read slab by slab looks like)
        pos = (rng.random((n, 3), dtype='f4') * np.float32(box)).astype('f4')
        return pos[np.argsort(pos[:, 0], kind='stable')]
    return ((rng.random((n, 3), dtype='f4') * 1.2 - 0.1) * np.float32(box)).astype('f4')   # some outside the box ('uniform', 'wide')


@pytest.mark.parametrize('kind,shape,n,offset', [('uniform', (512, 512, 512), 3_000_000, 0.0),
                                                 ('uniform', (256, 512, 384), 2_500_000, 0.3),
                                                 ('uniform', (512, 512, 512), 2_100_000, -0.7),
                                                 ('corners', (512, 512, 512), 2_200_000, 0.0),
                                                 ('slab', (512, 256, 512), 2_400_000, 0.5),
                                                 ('blob', (512, 512, 512), 2_600_000, 0.0),
                                                 # anisotropic meshes: an offset of -0.55 x-cells is -1.65 y-cells (nearest cells down to
                                                 # -2: what the packed lists hold); -0.8 x-cells = -2.4 y-cells goes to the first
                                                 # generation (found by scripts/gpu_lines_fuzz.sh: the grid coordinate was clamped at -2)
                                                 ('corners', (256, 768, 320), 2_300_000, -0.55),
                                                 ('corners', (256, 768, 320), 2_900_000, -0.8),
                                                 ('uniform-cfg1', (512, 512, 512), 2_300_000, 0.0),
                                                 ('corners-cfg1', (512, 512, 512), 2_050_000, 0.25)])
def test_line_lists_vs_first_generation_and_oracle(kind, shape, n, offset, options):
    """line lists (csrc/tsc_lines3.hpp block records + csrc/tsc_lines.hpp: whole-line scatter passes, packed 8-byte tile-relative
    entries, fixed-point tile sums) for unweighted float32 particles on meshes of whole 16 x 16 x 32 tiles: against the CPU oracle's
    _tsc_scatter, and against the first-generation lists (option tsc_oldlists).  Where every coordinate lies beyond the
    first 128 cells the packed entry holds the float32 offset exactly, so the two meshes agree to the rounding of the cell
    sums; below, an offset is rounded to 2^-16 of a cell without bias (up or down: weights off by at most 3e-5 of their value)."""
    from abacusutils_amd.analysis.tsc import tsc_parallel
    from oracle import oracle
    rng = np.random.default_rng(4242 + n)
    box = 700.0
    if kind.endswith('-cfg1'):   # the kernels of the 1024 x 1024 configuration (meshes beyond 131072 tiles), on a small mesh
        options.set('tsc_lines_cfg1', 1)
        kind = kind[:-5]
    pos = _line_case(kind, shape, n, box, rng)
    off = offset * box / shape[0]
    p1, p2, p3 = pos.copy(), pos.copy(), pos.copy()
    base = (rng.random(shape, dtype='f4') * np.float32(0.05))   # accumulate into a mesh that is not empty
    a, b, c = base.copy(), base.copy(), base.copy()
    assert tsc_parallel(p1, a, box, offset=off) is None
    options.set('tsc_oldlists', 1)
    tsc_parallel(p2, b, box, offset=off)
    oracle.tsc_parallel(p3, c, box, nthread=4, offset=off)
    np.testing.assert_array_equal(p1, p3)   # wrapped identically
    np.testing.assert_array_equal(p2, p3)
    scale = float(c.max())
    # (a pile-up of 2e6 particles in a few cells: the float32 MESH resolves 1e-7 of cells holding 1e5 - the sum over the cells
    # of `a - base` cannot conserve the mass better than that)
    heavy = kind == 'blob' or shape == (256, 768, 320)      # twelve corners share the catalogue: cells of 2e4 in a float32 mesh
    assert abs(float((a - base).sum(dtype='f8')) / n - 1) < (1e-5 if heavy else 2e-6)
    # blob: the oracle, like the reference, sums a cell in float32 - 1e5 addends into cells of 3e4 lose the small ones (5e-4
    # low, measured); the first generation's float64 tile sums (`b`) are the yardstick there
    np.testing.assert_allclose(a, c, rtol=2e-3 if kind == 'blob' else 5e-5, atol=4e-6 * scale)
    np.testing.assert_allclose(a, b, rtol=5e-5, atol=4e-6 * scale)
    # cells fed only by particles at p >= 128 in every dimension (the top cells also take the periodic images of p < 1/2):
    # with 64-bit tile sums (option tsc_acc64; the default 32-bit sums resolve 2^-S of a cell value, S from the tile's list)
    # the two generations agree to the rounding of the cell sums
    options.set('tsc_oldlists', 0)
    options.set('tsc_acc64', 1)
    a64 = base.copy()
    tsc_parallel(pos.copy(), a64, box, offset=off)
    np.testing.assert_allclose(a64, a, rtol=3e-6, atol=3e-6 * scale)
    assert np.abs(a64 - b)[132:shape[0] - 3, 132:shape[1] - 3, 132:shape[2] - 3].max() <= 3e-7 * scale
    options.set('tsc_acc64', 0)


@pytest.mark.parametrize('kind,nmesh,n', [('uniform', 512, 3_000_000), ('corners', 512, 2_200_000), ('uniform-cfg1', 512, 2_300_000),
                                          ('wide', 768, 2_600_000)])
def test_interlaced_pair_shares_one_block_record_build(kind, nmesh, n, options):
    """an interlaced pair (analysis/power_spectrum.py:951-998: the second deposit at offset = half a cell) builds its block
    records ONCE (csrc/tsc_lines3.hpp, EXT: the blocks of the 4-cell union of both clouds) and derives the shifted deposit's
    tile entries from them by adding half a cell to the fixed-point coordinates.  The shifted mesh against the oracle's
    _tsc_scatter at that offset and against the unshared build (option tsc_noshare: float32 (x + d/2) n/L evaluated per
    particle), the spectrum against the oracle; the profiler must show ONE count / coarse pass and TWO fine passes"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power, get_field_fft
    from abacusutils_amd.analysis.tsc import tsc_parallel
    from oracle import oracle
    rng = np.random.default_rng(777 + n)
    box = 700.0
    if kind.endswith('-cfg1'):
        options.set('tsc_lines_cfg1', 1)
        kind = kind[:-5]
    pos = _line_case(kind, (nmesh,) * 3, n, box, rng)
    kw = dict(kbins=24, mubins=3, paste='TSC', nmesh=nmesh, compensated=True, interlaced=True, poles=[0, 2, 4])
    _lib.profile_reset()
    _lib.profile_enable(True)
    tab = calc_power(pos.copy(), box, **kw)
    _lib.profile_enable(False)
    prof = _lib.profile_get()
    assert prof['tsc_lines_coarse'][1] == 1 and prof['tsc_lines_count'][1] == 1 and prof['tsc_lines_fine'][1] == 2, \
        {k: v for k, v in prof.items() if k.startswith('tsc_')}
    ref = oracle.calc_power(pos.copy(), box, nthread=oracle.max_threads(), accum64=True, **kw)
    np.testing.assert_array_equal(np.asarray(tab['N_mode']), ref['N_mode'])
    scale = np.abs(ref['power']).max()
    np.testing.assert_allclose(np.asarray(tab['power']), ref['power'], rtol=1e-5, atol=1e-6 * scale)
    np.testing.assert_allclose(np.asarray(tab['poles']), ref['poles'], rtol=1e-5, atol=2e-6 * scale)
    # the interlaced spectrum itself, shared against unshared lists
    a = get_field_fft(pos.copy(), box, nmesh, 'TSC', None, None, False, True)
    options.set('tsc_noshare', 1)
    b = get_field_fft(pos.copy(), box, nmesh, 'TSC', None, None, False, True)
    options.set('tsc_noshare', 0)
    assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
    # and the shifted mesh alone against the oracle's deposit at that offset
    off = 0.5 * box / nmesh
    c = np.zeros((nmesh,) * 3, dtype='f4')
    wrapped = pos.copy()
    oracle.tsc_parallel(wrapped, c, box, nthread=4, offset=off)
    d = np.zeros((nmesh,) * 3, dtype='f4')
    tsc_parallel(pos.copy(), d, box, offset=off)          # unshared gen-3 build at the float32 offset
    np.testing.assert_allclose(d, c, rtol=5e-5, atol=4e-6 * float(c.max()))


def test_deferred_list_build_and_its_overflow_path(options):
    """inside calc_power the list build of a mesh seen before sizes its buffers from the previous build and makes its tables on
    the device (csrc/tsc.hip deferred mode: no stream synchronise between the counting and the scattering pass).  Same
    spectrum as the synchronous build (option tsc_lines_sync = 1), bit for bit; a second catalogue ten times as clustered -
    and option tsc_lines_sync = 2, buffers for a twentieth of the particles - overflows them: the pipeline notices after its
    final synchronise and runs again with exact sizes"""
    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.power_spectrum import calc_power
    rng = np.random.default_rng(31)
    box, nmesh, n = 900.0, 512, 2_400_000
    pos = (rng.random((n, 3), dtype='f4') * np.float32(box)).astype('f4')
    kw = dict(kbins=32, mubins=3, paste='TSC', nmesh=nmesh, compensated=False, interlaced=False, poles=[0, 2])
    options.set('tsc_lines_sync', 1)
    ref = calc_power(pos.copy(), box, **kw)
    options.set('tsc_lines_sync', 0)

    def run(p):
        _lib.profile_reset()
        _lib.profile_enable(True)
        t = calc_power(p.copy(), box, **kw)
        _lib.profile_enable(False)
        return t, _lib.profile_get()

    first, _ = run(pos)                         # (the exact build above left its sizes behind: this one is deferred already)
    second, prof = run(pos)
    assert 'tsc_lines_tables' in prof and prof['tsc_lines_count'][1] == 1, sorted(prof)
    for t in (first, second):
        np.testing.assert_array_equal(np.asarray(t['N_mode']), np.asarray(ref['N_mode']))
        np.testing.assert_array_equal(np.asarray(t['power']), np.asarray(ref['power']))
    # a catalogue that needs far more room per block than the last one: half of it inside one block of tiles
    clustered = pos.copy()
    clustered[: n // 2] *= np.float32(0.12)
    options.set('tsc_lines_sync', 1)
    want = calc_power(clustered.copy(), box, **kw)
    options.set('tsc_lines_sync', 0)
    calc_power(pos.copy(), box, **kw)           # sizes of the uniform catalogue again
    got, prof = run(clustered)
    np.testing.assert_array_equal(np.asarray(got['power']), np.asarray(want['power']))
    options.set('tsc_lines_sync', 2)
    got2, prof2 = run(pos)
    assert prof2['tsc_lines_count'][1] == 2 and prof2['tsc_lines_tables'][1] == 1, {k: v[1] for k, v in prof2.items() if k.startswith('tsc_')}
    np.testing.assert_array_equal(np.asarray(got2['power']), np.asarray(ref['power']))


def test_garbage_positions_stay_inside_the_tables():
    """NaN, +-inf and absurd coordinates among 2.2e6 ordinary particles: the list build clamps the grid coordinate before anything
    indexes with it (csrc/tsc_lines3.hpp: l3_S), so such particles are deposited somewhere on the mesh with ordinary weights
    instead of taking the process (or the GPU) down; every particle still deposits exactly one unit of mass"""
    from abacusutils_amd.analysis.tsc import tsc_parallel
    rng = np.random.default_rng(99)
    box, ng, n = 500.0, 512, 2_200_000
    pos = (rng.random((n, 3), dtype='f4') * np.float32(box)).astype('f4')
    bad = rng.choice(n, 120, replace=False)
    vals = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3.4e38], dtype='f4')
    pos[bad, rng.integers(0, 3, bad.size)] = vals[rng.integers(0, vals.size, bad.size)]
    mesh = np.zeros((ng,) * 3, dtype='f4')
    with np.errstate(all='ignore'):
        tsc_parallel(pos.copy(), mesh, box)
    assert np.isfinite(mesh).all()
    assert abs(float(mesh.sum(dtype='f8')) / n - 1) < 2e-6
