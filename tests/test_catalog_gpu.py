"""HIP catalogue-side kernels (abacus_unpack_rvint / abacus_unpack_pids / abacus_menv) through the reference's Python
interface, against the golden vectors of the shimmed reference (bit-exact for the unpackers) and the oracle."""
from pathlib import Path

import numpy as np
import pytest

from abacusutils_amd import _lib
from abacusutils_amd.data import bitpacked
from abacusutils_amd.hod.menv import do_Menv_from_tree
from oracle import oracle

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).parent / 'golden' / 'catalog_cases.npz')
MENV_CASES = sorted({k.split('.')[1] for k in G.files if k.startswith('menv.')})


@pytest.mark.parametrize('tag,ft', [('f4', np.float32), ('f8', np.float64)])
def test_unpack_rvint_golden(tag, ft):
    rv = G['rvint.in']
    pos, vel = bitpacked.unpack_rvint(rv, 32.0, float_dtype=ft)
    assert pos.dtype == ft and pos.shape == (len(rv), 3)
    assert np.array_equal(pos, G[f'rvint.pos.{tag}']) and np.array_equal(vel, G[f'rvint.vel.{tag}'])
    # the reference's output-argument protocol (:61-97): False skips, an array is filled and the count returned
    p2 = np.empty((len(rv), 3), dtype=ft)
    r = bitpacked.unpack_rvint(rv.ravel(), 32.0, float_dtype=ft, posout=p2, velout=False)
    assert r == (len(rv), 0) and np.array_equal(p2, pos)
    r = bitpacked.unpack_rvint(rv, 32.0, float_dtype=ft, posout=False)
    assert r[0] == 0 and np.array_equal(r[1], vel)


@pytest.mark.parametrize('n', [0, 1, 2, 3, 5, 1000003])
def test_unpack_rvint_ragged_sizes(n):
    rng = np.random.default_rng(n)
    rv = rng.integers(-2**31, 2**31, size=(n, 3), dtype=np.int64).astype(np.int32)
    pos, vel = bitpacked.unpack_rvint(rv, 2000.0)
    wp, wv = oracle.unpack_rvint(rv, 2000.0)
    assert np.array_equal(pos, wp) and np.array_equal(vel, wv)


def test_unpack_rvint_device_resident():
    rng = np.random.default_rng(1)
    rv = rng.integers(-2**31, 2**31, size=(4096, 3), dtype=np.int64).astype(np.int32)
    d_in = _lib.DeviceArray(rv)
    d_pos = _lib.DeviceArray(nbytes=rv.size * 4, dtype=np.float32, shape=rv.shape)
    d_vel = _lib.DeviceArray(nbytes=rv.size * 4, dtype=np.float32, shape=rv.shape)
    assert bitpacked.unpack_rvint(d_in, 500.0, posout=d_pos, velout=d_vel) == (4096, 4096)
    wp, wv = oracle.unpack_rvint(rv, 500.0)
    assert np.array_equal(d_pos.get(), wp) and np.array_equal(d_vel.get(), wv)


@pytest.mark.parametrize('tag,ft', [('f4', np.float32), ('f8', np.float64)])
def test_unpack_pids_golden(tag, ft):
    r = bitpacked.unpack_pids(G['pids.in'], box=32.0, ppd=64, pid=True, lagr_pos=True, tagged=True, density=True,
                              lagr_idx=True, float_dtype=ft)
    assert sorted(r) == ['density', 'lagr_idx', 'lagr_pos', 'pid', 'tagged']
    for k, v in r.items():
        want = G[f'pids.{k}.{tag}']
        assert v.dtype == want.dtype and v.shape == want.shape and np.array_equal(v, want), k


def test_unpack_pids_subsets_and_errors():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 2**63, size=100001, dtype=np.int64).astype(np.uint64)
    want = oracle.unpack_pids(a, box=2000.0, ppd=6912)
    r = bitpacked.unpack_pids(a, box=2000.0, ppd=6912.0, lagr_pos=True, density=True)
    assert sorted(r) == ['density', 'lagr_pos']
    assert np.array_equal(r['lagr_pos'], want['lagr_pos']) and np.array_equal(r['density'], want['density'])
    assert bitpacked.unpack_pids(a) == {}
    assert np.array_equal(bitpacked.unpack_pids(a, pid=True)['pid'], want['pid'])
    assert len(bitpacked.unpack_pids(a[:0], pid=True, tagged=True)['tagged']) == 0
    with pytest.raises(ValueError):
        bitpacked.unpack_pids(a, lagr_pos=True, ppd=64)
    with pytest.raises(ValueError):
        bitpacked.unpack_pids(a, lagr_pos=True, box=1.0)
    with pytest.raises(ValueError):
        bitpacked.unpack_pids(a, lagr_pos=True, box=1.0, ppd=64.5)


@pytest.mark.parametrize('name', MENV_CASES)
def test_menv_golden(name):
    c = {k.split('.', 2)[2]: G[k] for k in G.files if k.startswith(f'menv.{name}.')}
    want = c.pop('Menv')
    pos0 = c['pos'].copy()
    got = do_Menv_from_tree(c['pos'], c['mass'], r_inner=c['r_inner'][()], r_outer=c['r_outer'][()],
                            halo_lc=bool(c['halo_lc']), Lbox=float(c['Lbox']), nthread=4, mcut=float(c['mcut']))
    assert np.array_equal(c['pos'], pos0)            # "don't modify the user's input in place" (menv.py:38)
    assert got.dtype == want.dtype and got.shape == want.shape
    scale = np.abs(want).max()
    tol = 1e-12 if c['mass'].dtype == np.float64 else 3e-6   # the reference sums float32 masses in float32
    assert np.abs(got - want).max() <= tol * scale
    assert np.array_equal(got[c['mass'] <= c['mcut']], np.zeros(int((c['mass'] <= c['mcut']).sum()), dtype=got.dtype))
    # and the float64 brute-force restatement, to rounding of the float64 sums
    ref = oracle.menv_brute(c['pos'], c['mass'], c['r_inner'][()], c['r_outer'][()], bool(c['halo_lc']), float(c['Lbox']),
                            mcut=float(c['mcut']))
    if c['mass'].dtype == np.float64:
        assert np.abs(got - ref).max() <= 1e-13 * scale


def test_menv_large_uniform_and_tiny_boxes():
    rng = np.random.default_rng(11)
    # 2x10^5 halos, cells of r_outer: every 27-cell stencil path incl. the periodic wrap
    n, L = 200_000, 500.0
    pos = (rng.random((n, 3)) * L - L / 2).astype(np.float32)
    mass = 10 ** (10.5 + rng.exponential(0.4, n))
    rin = (0.1 + 0.5 * rng.random(n)).astype(np.float32)
    got = do_Menv_from_tree(pos, mass, rin, 5.0, False, L, mcut=1e11)
    sel = rng.choice(np.nonzero(mass > 1e11)[0], 300, replace=False)
    sub = np.zeros(n, dtype=bool)
    sub[sel] = True
    # brute force for a sample of centres: mask the others by raising their mass cut individually
    p = ((pos + L / 2.0) % L).astype(np.float64)
    for i in sel[:300]:
        d = np.abs(p - p[i])
        d = np.where(d > L / 2, L - d, d)
        d2 = (d * d).sum(axis=1)
        w = mass[d2 <= 25.0].sum() - mass[d2 <= float(rin[i]) ** 2].sum()
        assert abs(got[i] - w) <= 1e-12 * max(abs(w), mass[i])
    # fewer than 3 cells per dimension: box barely larger than 2 r_outer
    n2 = 500
    pos2 = (rng.random((n2, 3)) * 11.0 - 5.5)
    m2 = 10 ** (11 + rng.random(n2))
    got2 = do_Menv_from_tree(pos2, m2, 0.3, 5.0, False, 11.0, mcut=1e11)
    ref2 = oracle.menv_brute(pos2, m2, 0.3, 5.0, False, 11.0, mcut=1e11)
    assert np.abs(got2 - ref2).max() <= 1e-12 * np.abs(ref2).max()
    # empty input and no centres
    assert len(do_Menv_from_tree(np.zeros((0, 3)), np.zeros(0), 1.0, 2.0, False, 10.0)) == 0
    assert not do_Menv_from_tree(pos2, m2, 0.3, 5.0, False, 11.0, mcut=1e20).any()


def test_against_the_numba_compiled_reference():
    """the reference's own stored outputs (ref_data/test_read_asdf.asdf, test_pack9.asdf, test_pack9_pid.asdf)"""
    from abacusutils_amd.data.pack9 import unpack_pack9
    pos, vel = bitpacked.unpack_rvint(G['real.rvint.in'], 32.0)
    assert np.array_equal(pos, G['real.rvint.pos']) and np.array_equal(vel, G['real.rvint.vel'])
    for case in ('pids', 'pack9pid'):
        r = bitpacked.unpack_pids(G[f'real.{case}.in'], box=32.0, ppd=64, pid=True, lagr_pos=True, tagged=True,
                                  density=True, lagr_idx=True)
        for k, v in r.items():
            want = G[f'real.{case}.{k}']
            assert v.dtype == want.dtype and np.array_equal(v, want), (case, k)
    pos, vel = unpack_pack9(G['real.pack9.in'], float(G['real.pack9.box']), float(G['real.pack9.velz']))
    assert pos.dtype == np.float32 and pos.shape == G['real.pack9.pos'].shape
    assert np.array_equal(pos, G['real.pack9.pos']) and np.array_equal(vel, G['real.pack9.vel'])


def test_pack9_float64_protocol_and_edges():
    from abacusutils_amd.data.pack9 import unpack_pack9
    d8, velz = G['pack9.field.in'], float(G['real.pack9.velz'])
    for ft in (np.float32, np.float64):
        pos, vel = unpack_pack9(d8, 32.0, velz, float_dtype=ft)
        wp, wv = oracle.unpack_pack9(d8, 32.0, velz, ft)
        assert pos.dtype == ft and np.array_equal(pos, wp) and np.array_equal(vel, wv)
    # caller-provided / skipped outputs (pack9.py:24-56)
    buf = np.full((len(d8), 3), -7.0)
    r = unpack_pack9(d8, 32.0, velz, float_dtype=np.float64, posout=False, velout=buf)
    assert r == (0, len(pos)) and np.array_equal(buf[:len(pos)], vel) and (buf[len(pos):] == -7.0).all()
    # chunk boundaries, headers as the last / first record of a chunk, no header at all, empty input
    d = G['real.pack9.in']
    rng = np.random.default_rng(9)
    hdrs = np.nonzero(d[:, 0] == 0xFF)[0]
    for n in (1, 2, 1535, 1536, 1537, 3072, 3073, 20000):
        for start in (0, int(hdrs[3]), int(hdrs[3]) + 1):
            sub = d[start:start + n]
            gp, gv = unpack_pack9(sub, 32.0, 3200.0)
            wp, wv = oracle.unpack_pack9(sub, 32.0, 3200.0)
            assert gp.shape == wp.shape and np.array_equal(gp, wp, equal_nan=True) and np.array_equal(gv, wv, equal_nan=True)
    parts = d[d[:, 0] != 0xFF][:5000]           # 5000 particles, no header: more than three chunks of NaN state
    gp, gv = unpack_pack9(parts, 32.0, 3200.0)
    assert gp.shape == (5000, 3) and np.isnan(gp).all() and np.isnan(gv).all()
    big = np.concatenate([d[hdrs[0]:hdrs[0] + 1], parts, d[hdrs[1]:hdrs[1] + 1], parts[:7]])   # one header governs 3+ chunks
    gp, gv = unpack_pack9(big, 32.0, 3200.0)
    wp, wv = oracle.unpack_pack9(big, 32.0, 3200.0)
    assert np.array_equal(gp, wp) and np.array_equal(gv, wv)
    gp, gv = unpack_pack9(np.zeros((0, 9), dtype=np.uint8), 32.0, 3200.0)
    assert gp.shape == (0, 3) and gv.shape == (0, 3)
    rnd = rng.integers(0, 256, size=(100000, 9), dtype=np.int64).astype(np.uint8)   # random bytes: ~1/256 are headers
    gp, gv = unpack_pack9(rnd, 2000.0, 208774.9, float_dtype=np.float64)
    wp, wv = oracle.unpack_pack9(rnd, 2000.0, 208774.9, np.float64)
    assert np.array_equal(gp, wp, equal_nan=True) and np.array_equal(gv, wv, equal_nan=True)


def test_concurrent_callers_are_serialised():
    """ctypes releases the GIL: entry points share one stream and scratch buffers, so they lock (ABACUS_ENTER)"""
    import threading
    from abacusutils_amd.analysis.tpcf_corrfunc import _paircount
    rng = np.random.default_rng(21)
    rv = rng.integers(-2**31, 2**31, size=(300000, 3), dtype=np.int64).astype(np.int32)
    want_rv = oracle.unpack_rvint(rv, 100.0)
    pts = (rng.random((3, 40000), dtype=np.float32) * np.float32(200.0))
    bins = np.linspace(0.5, 10, 8).astype(np.float32)
    want_dd = _paircount(0, pts[0], pts[1], pts[2], 200.0, bins)
    pos = (rng.random((30000, 3)) * 100 - 50).astype(np.float32)
    mass = 10 ** (11 + rng.random(30000))
    want_me = do_Menv_from_tree(pos, mass, 0.4, 4.0, False, 100.0)
    bad = []

    def work(kind):
        for _ in range(20):
            if kind == 0:
                p, v = bitpacked.unpack_rvint(rv, 100.0)
                ok = np.array_equal(p, want_rv[0]) and np.array_equal(v, want_rv[1])
            elif kind == 1:
                ok = np.array_equal(_paircount(0, pts[0], pts[1], pts[2], 200.0, bins), want_dd)
            else:
                ok = np.allclose(do_Menv_from_tree(pos, mass, 0.4, 4.0, False, 100.0), want_me, rtol=1e-12, atol=0)
            if not ok:
                bad.append(kind)
    ts = [threading.Thread(target=work, args=(k % 3,)) for k in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, bad
