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
