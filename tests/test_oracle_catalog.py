"""The oracle's catalogue-side restatements (unpack_rvint, unpack_pids, menv_brute) against golden vectors of the
shimmed reference on the Mini_N64_L32 subsample files and seeded halos (tests/golden/catalog_cases.npz, written by
oracle/make_golden.py catalog)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle

G = np.load(Path(__file__).parent / 'golden' / 'catalog_cases.npz')
MENV_CASES = sorted({k.split('.')[1] for k in G.files if k.startswith('menv.')})


@pytest.mark.parametrize('tag,ft', [('f4', np.float32), ('f8', np.float64)])
def test_unpack_rvint(tag, ft):
    pos, vel = oracle.unpack_rvint(G['rvint.in'], 32.0, ft)
    assert pos.dtype == ft and np.array_equal(pos, G[f'rvint.pos.{tag}'])
    assert np.array_equal(vel, G[f'rvint.vel.{tag}'])
    # 20-bit positions span [-0.5, 0.5) of the box, 12-bit velocities +-6000 km/s
    assert np.abs(pos).max() <= 16.78 and np.abs(vel).max() <= 6000.0


@pytest.mark.parametrize('tag,ft', [('f4', np.float32), ('f8', np.float64)])
def test_unpack_pids(tag, ft):
    r = oracle.unpack_pids(G['pids.in'], box=32.0, ppd=64, float_dtype=ft)
    for k in ('pid', 'lagr_idx', 'lagr_pos', 'tagged', 'density'):
        want = G[f'pids.{k}.{tag}']
        assert r[k].dtype == want.dtype and np.array_equal(r[k], want), k


@pytest.mark.parametrize('name', MENV_CASES)
def test_menv(name):
    c = {k.split('.', 2)[2]: G[k] for k in G.files if k.startswith(f'menv.{name}.')}
    want = c.pop('Menv')
    got = oracle.menv_brute(c['pos'], c['mass'], c['r_inner'][()], c['r_outer'][()], bool(c['halo_lc']), float(c['Lbox']),
                            mcut=float(c['mcut']))
    assert got.dtype == want.dtype
    # the reference sums the neighbour masses pairwise in the mass dtype, in tree order: agreement to rounding of M(<r_outer)
    scale = np.abs(want).max()
    tol = 1e-12 if c['mass'].dtype == np.float64 else 3e-6
    assert np.abs(got - want).max() <= tol * scale
    assert np.array_equal(got == 0, want == 0) or tol > 1e-9


def test_against_the_numba_compiled_reference():
    """outputs of the real (Numba) reference kept by its own tests (ref_data/test_read_asdf.asdf, test_pack9.asdf,
    test_pack9_pid.asdf): bit-equal, which also settles the mixed int/float32/float64 typing of the loops"""
    pos, vel = oracle.unpack_rvint(G['real.rvint.in'], 32.0)
    assert np.array_equal(pos, G['real.rvint.pos']) and np.array_equal(vel, G['real.rvint.vel'])
    for case in ('pids', 'pack9pid'):
        r = oracle.unpack_pids(G[f'real.{case}.in'], box=32.0, ppd=64)
        for k in ('pid', 'lagr_pos', 'lagr_idx', 'tagged', 'density'):
            want = G[f'real.{case}.{k}']
            assert r[k].dtype == want.dtype and np.array_equal(r[k], want), (case, k)
    pos, vel = oracle.unpack_pack9(G['real.pack9.in'], float(G['real.pack9.box']), float(G['real.pack9.velz']))
    assert pos.dtype == np.float32 and np.array_equal(pos, G['real.pack9.pos'])
    assert np.array_equal(vel, G['real.pack9.vel'])


def test_pack9_float64_and_headerless_prefix():
    # float64: no golden exists (the shim cannot run pack9, see oracle/make_golden.py); float32 rounding of the float64
    # result must land within one ulp of the pinned float32 path
    pos, vel = oracle.unpack_pack9(G['real.pack9.in'], 32.0, 3200.0, np.float64)
    assert pos.dtype == np.float64
    np.testing.assert_allclose(pos, G['real.pack9.pos'], rtol=0, atol=4e-6)
    np.testing.assert_allclose(vel, G['real.pack9.vel'], rtol=3e-7, atol=0)
    # particles ahead of the first header: NaN state of the reference (pack9.py:66-71)
    d = G['real.pack9.in']
    first = int(np.nonzero(d[:, 0] == 0xFF)[0][0])
    tail = np.concatenate([d[first + 1:first + 4], d[first:]])
    pos, vel = oracle.unpack_pack9(tail, 32.0, 3200.0)
    assert np.isnan(pos[:3]).all() and np.isnan(vel[:3]).all() and not np.isnan(pos[3:]).any()
