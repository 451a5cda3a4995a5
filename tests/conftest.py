import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
GOLD = REPO / 'tests' / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(GOLD / f'{name}.npz', allow_pickle=False))


def unpack_inputs(g):
    """inverse of oracle/make_golden.py:pack_inputs"""
    hd = {k[2:]: v for k, v in g.items() if k.startswith('h.')}
    pd = {k[2:]: v for k, v in g.items() if k.startswith('p.')}
    params = {}
    for k, v in g.items():
        if k.startswith('params.'):
            params[k[7:]] = v if v.ndim else v.item()
    params.setdefault('origin', None)
    return hd, pd, params


def unpack_mock(g, prefix):
    mock = {}
    for k, v in g.items():
        if k.startswith(prefix + '.'):
            _, tr, col = k.split('.')
            mock.setdefault(tr, {})[col] = v
    return mock


def synth_case(g):
    """rebuild the seeded synthetic inputs a hod_synth_*.npz golden was generated from"""
    from abacusutils_amd import synth
    origin = g['meta.origin'] if 'meta.origin' in g else None
    hd, pd, params = synth.synth_hod_inputs(int(g['meta.nh']), int(g['meta.np']), seed=int(g['meta.seed']),
                                            with_ranks=bool(g['meta.ranks']), origin=origin)
    tracers = json.loads(str(g['meta.tracers']))
    return hd, pd, params, tracers, bool(g['meta.ranks']), bool(g['meta.rsd'])


def input_checksum(hd, pd):
    s = 0.0
    for d in (hd, pd):
        for k in sorted(d):
            a = np.asarray(d[k], dtype=np.float64).ravel()
            s += float(np.dot(a, np.cos(np.arange(a.size) * 0.001)))
    return s


def assert_mock_equal(got, want, exact=True, rtol=0.0):
    assert set(got) == set(want)
    for tr in want:
        assert int(got[tr]['Ncent']) == int(want[tr]['Ncent']), tr
        for col in ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass', 'id'):
            a, b = np.asarray(got[tr][col]), np.asarray(want[tr][col])
            assert a.shape == b.shape, (tr, col, a.shape, b.shape)
            if col == 'id' or exact:
                np.testing.assert_array_equal(a, b, err_msg=f'{tr}.{col}')
            else:
                np.testing.assert_allclose(a, b, rtol=rtol, atol=0, err_msg=f'{tr}.{col}')


def assert_spectrum_close(got, want, rtol=1e-5, floor=0.1, err_msg=''):
    """north_star tolerance on a binned spectrum: |got - want| <= rtol * max(|want|, floor * max|want|) element by element -
    relative `rtol` for every value within a decade of the array's largest, and below that (a cross spectrum or an
    l = 2, 4 multipole changing sign, the DC bin) an absolute floor of rtol * floor * max|want|: a relative error means
    nothing at a zero crossing.  NaN (empty bins) must match."""
    got, want = np.asarray(got, dtype='f8'), np.asarray(want, dtype='f8')
    assert got.shape == want.shape, (err_msg, got.shape, want.shape)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan), f'{err_msg}: empty bins differ'
    if nan.all():
        return
    scale = np.abs(want[~nan]).max()
    tol = rtol * np.maximum(np.abs(want), floor * scale)
    bad = ~nan & ~(np.abs(got - want) <= tol)
    if bad.any():
        rel = np.abs(got - want)[bad] / np.maximum(np.abs(want[bad]), floor * scale)
        raise AssertionError(f'{err_msg}: {int(bad.sum())} of {want.size} values outside {rtol:g} (worst {rel.max():.3g} of the '
                             f'floored value, {(np.abs(got - want)[bad] / scale).max():.3g} of the largest)')


SYNTH_CASES = ['lrg', 'all_rich', 'all_rich_ranks', 'all_rich_norsd', 'all_rich_lc', 'elg_only', 'qso_only', 'lrg_qso']


class _Options:
    def __init__(self):
        self._set = set()

    def set(self, name, value=1):
        from abacusutils_amd import _lib
        _lib.set_option(name, value)
        self._set.add(name)

    def reset(self):
        from abacusutils_amd import _lib
        for name in self._set:
            _lib.set_option(name, 0)
        self._set.clear()


@pytest.fixture
def options():
    """diagnostic options of the library (comparator code paths), back to the production defaults after the test"""
    o = _Options()
    yield o
    o.reset()
