"""Host-side behaviour of the mock containers of run_hod (abacusutils_amd/hod/GRAND_HOD.py): MockDict / LazyTracer are
dicts in every way the reference's plain `mock_dict` is used - indexing, iteration, dict(), pickling - with a stand-in
for the staged catalogue (no GPU)."""
import pickle

import numpy as np
import pytest

from abacusutils_amd import _lib
from abacusutils_amd.hod.GRAND_HOD import COLS, LazyTracer, MockDict


class FakeStaged:
    _h = 1
    generation = 1
    fetched = 0

    def fetch(self, tracer):
        self.fetched += 1
        d = {'Ncent': 2}
        for q, c in enumerate(COLS):
            d[c] = np.arange(5, dtype=np.float64) + q
        d['id'] = np.arange(5, dtype=np.int64)
        return d


def test_lazy_tracer_is_a_dict_that_loads_once():
    st = FakeStaged()
    t = LazyTracer(st, 'LRG', 2)
    assert t['Ncent'] == 2 and 'x' in t and list(t) == ['Ncent'] + list(COLS) + ['id'] and st.fetched == 0
    assert 'columns in HBM' in repr(t)
    np.testing.assert_array_equal(t['vx'], np.arange(5.0) + 3)
    assert st.fetched == 1
    plain = dict(t)
    assert type(plain) is dict and plain['id'].dtype == np.int64 and st.fetched == 1
    assert {**t}.keys() == plain.keys() and t.get('nope', 7) == 7 and t.pop('Ncent') == 2


def test_lazy_tracer_refuses_a_replaced_catalogue():
    st = FakeStaged()
    t = LazyTracer(st, 'ELG', 0)
    st.generation += 1                      # a later populate rewrote the device columns
    with pytest.raises(RuntimeError, match='never read'):
        t['x']
    with pytest.raises(RuntimeError, match='never read'):
        dict(t)
    assert t['Ncent'] == 0                  # the count never needed the device


def test_writes_to_a_lazy_tracer_materialise_it_first():
    """an assignment into a never-read lazy tracer must survive the load of the other columns, and the tracer
    then counts as read (MockDict.device_xyz compares checksums of loaded columns, so an edited x is seen)"""
    for write in (lambda t: t.__setitem__('x', np.full(5, -1.0)), lambda t: t.update(x=np.full(5, -1.0)),
                  lambda t: t.setdefault('w', np.ones(5))):
        st = FakeStaged()
        t = LazyTracer(st, 'LRG', 2)
        write(t)
        assert st.fetched == 1 and '_staged' not in t.__dict__          # loaded BEFORE the write, no longer lazy
        np.testing.assert_array_equal(t['y'], np.arange(5.0) + 1)       # reading another column does not reload
        assert st.fetched == 1
        if 'w' in t:
            np.testing.assert_array_equal(t['x'], np.arange(5.0))
        else:
            np.testing.assert_array_equal(t['x'], np.full(5, -1.0))     # the caller's column is still there
    st = FakeStaged()
    t = LazyTracer(st, 'LRG', 2)
    st.generation += 1
    with pytest.raises(RuntimeError, match='never read'):
        t['x'] = 0                                                       # stale: a write refuses like a read


def test_mock_dict_pickles_as_a_plain_dict():
    st = FakeStaged()
    m = MockDict({'LRG': st.fetch('LRG'), 'ELG': LazyTracer(st, 'ELG', 2)})._bind(st)
    back = pickle.loads(pickle.dumps(m))
    assert type(back) is dict and type(back['ELG']) is dict and type(back['LRG']) is dict
    np.testing.assert_array_equal(back['ELG']['z'], np.arange(5.0) + 2)
    assert back['LRG']['Ncent'] == 2


def test_poshash_sees_single_rows_and_swaps():
    a = np.random.default_rng(1).random(1000)
    h = _lib.poshash_host(a)
    b = a.copy()
    b[123] = np.nextafter(b[123], 2.0)
    assert _lib.poshash_host(b) != h
    c = a.copy()
    c[[3, 900]] = c[[900, 3]]
    assert _lib.poshash_host(c) != h and _lib.poshash_host(a.copy()) == h


def test_pinned_pool_falls_back_to_plain_memory_without_a_gpu():
    if _lib.device_count() > 0:
        pytest.skip('a GPU is present')
    a = _lib.pinned_empty((8, 100), np.float64)
    assert a.shape == (8, 100) and a.dtype == np.float64
    a[:] = 1.0
