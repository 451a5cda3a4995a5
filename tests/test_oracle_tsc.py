"""CPU oracle TSC / CIC / partition vs the reference's saved grids (tests/ref_tsc/*.asdf, decoded into
tests/golden/tsc_ref.npz) and vs golden vectors from the reference's functions.  No GPU needed."""
import numpy as np
import pytest
from conftest import load_golden

from oracle import oracle


def _dense(g, key, ng):
    a = np.zeros(ng**3, dtype=np.float32)
    a[g[key + '.idx']] = g[key + '.val']
    return a.reshape(ng, ng, ng)


def _test_multi_inputs(dtype):
    # tests/test_tsc.py:102-109 of the reference
    rng = np.random.default_rng(234)
    pos = rng.random((10000, 3), dtype='f4').astype(dtype) * 123.0
    weights = rng.random((10000,), dtype='f4').astype(dtype)
    return pos, weights


@pytest.mark.parametrize('ngrid', [10, 256])
@pytest.mark.parametrize('dtype', ['f4', 'f8'])
@pytest.mark.parametrize('nthread', [1, 4])
def test_multi_vs_reference_grids(ngrid, dtype, nthread):
    """tests/test_tsc.py:92-159: mass conservation, saved pure-Python grid, nbodykit grid (rtol 1e-4, atol 1e-5)"""
    ref = load_golden('tsc_ref')
    pos, weights = _test_multi_inputs(dtype)
    dens = oracle.tsc_parallel(pos, ngrid, 123.0, weights=weights, nthread=nthread)
    assert np.isclose(dens.sum(dtype='f8'), weights.sum(dtype='f8'))
    assert np.allclose(dens, _dense(ref, f'tsc_ngrid{ngrid}', ngrid), rtol=1e-4, atol=1e-5)
    assert np.allclose(dens, _dense(ref, f'nbodykit_tsc_ngrid{ngrid}', ngrid), rtol=1e-4, atol=1e-5)
    if dtype == 'f8' and nthread == 1:
        # the saved grid was made by the serial f8 pure-Python scatter: same order, same arithmetic -> bitwise
        np.testing.assert_array_equal(dens, _dense(ref, f'tsc_ngrid{ngrid}', ngrid))


def test_single_particle():
    """tests/test_tsc.py:25-90 analytic 27-cell weights"""
    ngrid, box = 10, 123.0
    cen = np.array([5, 6, 7])
    single = (cen / ngrid * box).astype('f4').reshape(1, -1)
    dens = oracle.tsc_parallel(single, ngrid, box, nthread=1)
    assert (dens == 0).sum() == ngrid**3 - 27
    assert np.isclose(dens.sum(), 1.0)
    cube = dens[4:7, 5:8, 6:9]
    assert np.allclose(cube[1, 1, 1], 0.75**3)
    assert np.allclose(cube[0, 0, 0], 0.5**9)
    assert np.allclose(cube[0, 1, 1], 0.5**3 * 0.75**2)
    assert np.allclose(cube[0, 0, 1], 0.5**6 * 0.75)


@pytest.mark.parametrize('name,dtype', [('f4_w', 'f4'), ('f4_now', 'f4'), ('f8_w', 'f8'), ('f4_offset', 'f4'),
                                        ('f4_aniso', 'f4'), ('f8_grid64', 'f8')])
def test_scatter_cases_bitwise(name, dtype):
    """_tsc_scatter golden vectors: serial order + IEEE arithmetic.  float64 cases are bitwise equal; in float32
    the shimmed reference evaluates `dx**2` through powf(dx, 2) (NumPy scalar power) where Numba and the oracle
    multiply, which moves an occasional cell by one ulp -> 2e-7 relative."""
    g = load_golden('tsc_cases')
    box = float(g[name + '.box'])
    pos = (g['base'].astype(dtype) * box).astype(dtype)
    w = g['wts'].astype(dtype) if name not in ('f4_now', 'f8_grid64') else None
    want = g[name + '.grid']
    dens = np.zeros(want.shape, dtype=want.dtype)
    oracle.tsc_scatter(pos, dens, box, w, float(g[name + '.offset']))
    if dtype == 'f8':
        np.testing.assert_array_equal(dens, want)
    else:
        np.testing.assert_allclose(dens, want, rtol=2e-7, atol=0)
        assert (dens != want).mean() < 0.01


def test_parallel_wrap_and_accumulate():
    """tsc_parallel contract (analysis/tsc.py:45-50,171-173): wraps pos IN PLACE, accumulates into the grid"""
    g = load_golden('tsc_cases')
    pos = g['parallel_wrap.pos_in'].copy()
    grid = np.full((12, 12, 12), 0.25, dtype=np.float32)
    r = oracle.tsc_parallel(pos, grid, 50.0, weights=g['wts'], nthread=1)
    assert r is None
    np.testing.assert_array_equal(pos, g['parallel_wrap.pos_out'])
    np.testing.assert_allclose(grid, g['parallel_wrap.grid'], rtol=2e-7, atol=0)
    # stripes change only the summation order
    pos = g['parallel_wrap.pos_in'].copy()
    grid2 = np.full((12, 12, 12), 0.25, dtype=np.float32)
    oracle.tsc_parallel(pos, grid2, 50.0, weights=g['wts'], nthread=4)
    assert np.allclose(grid2, grid, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('nthread', [1, 3])
def test_partition_golden(nthread):
    g = load_golden('tsc_cases')
    pos = (g['base'] * np.float32(50.0)).astype('f4')
    ps, st, ws = oracle.partition_parallel(pos, 7, 50.0, weights=g['wts'], nthread=nthread)
    np.testing.assert_array_equal(st, g['partition.starts'])
    np.testing.assert_array_equal(ps, g['partition.psort'])   # stable -> identical order
    np.testing.assert_array_equal(ws, g['partition.wsort'])


@pytest.mark.parametrize('seed', [123, 456])
@pytest.mark.parametrize('dtype', ['f4', 'f8'])
@pytest.mark.parametrize('npartition', [1, 1000])
def test_partition_vs_numpy(seed, dtype, npartition):
    """tests/test_tsc.py:162-208"""
    rng = np.random.default_rng(seed)
    box, N = 123.0, 10000
    pos = rng.random((N, 3), dtype=dtype) * box
    weights = rng.random((N,), dtype=dtype)
    ppart, starts, wpart = oracle.partition_parallel(pos, npartition, box, weights=weights, nthread=4)
    keys = (pos[:, 0] * (npartition / box)).astype(np.int32)
    iord = keys.argsort(kind='stable')
    np_starts = np.zeros(npartition + 1, dtype=np.int64)
    np_starts[1:] = np.bincount(keys, minlength=npartition).cumsum()
    np.testing.assert_array_equal(starts, np_starts)
    np.testing.assert_array_equal(ppart, pos[iord])
    np.testing.assert_array_equal(wpart, weights[iord])


def test_cic_golden():
    g = load_golden('tsc_cases')
    pos = (g['base'] * np.float32(50.0)).astype('f4')
    dens = np.zeros((12, 12, 12), dtype=np.float32)
    oracle.cic_serial(pos, dens, 50.0, weights=g['wts'])
    # the shimmed reference divides f32 positions in float32 (NumPy-2 promotion), Numba and the oracle in float64
    assert np.allclose(dens, g['cic_f4_w.grid'], rtol=1e-4, atol=1e-5)
    assert np.isclose(dens.sum(dtype='f8'), g['wts'].sum(dtype='f8'))


def test_bad_npartition():
    pos = np.zeros((10, 3), dtype='f4')
    with pytest.raises(ValueError):
        oracle.tsc_parallel(pos, 12, 1.0, nthread=4, npartition=5)
