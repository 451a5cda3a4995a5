"""abacusutils_amd.comm.RcclComm on the one GPU of the test box: a ONE-rank RCCL communicator through the C ABI
(file rendezvous -> ncclCommInitRank -> collectives on the library stream).  More than one rank needs more than one GPU
(RCCL refuses two ranks on one device); the multi-rank orchestration is covered on the CPU by the gloo stand-in
(tests/test_slab_power.py, test_slab_pairs.py, test_hod_shard.py) and the launcher by tests/test_launcher.py."""
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CODE = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from abacusutils_amd import _lib
from abacusutils_amd.comm import RcclComm
from abacusutils_amd.analysis.slab_power import HipBuf
c = RcclComm.from_env()
assert 'torch' not in sys.modules
i = c.info()
assert (i['rank'], i['world']) == (0, 1) and i['rccl_version'] >= 20000, i
a = np.arange(5, dtype=np.float64)
assert np.array_equal(c.all_reduce_array(a.copy()), a)
assert np.array_equal(c.all_reduce_array(np.array([3, 9], dtype=np.int64), 'max'), [3, 9])
assert c.all_reduce_int(7) == 7 and c.all_reduce_float(2.5, 'max') == 2.5
assert np.array_equal(c.all_gather_array(np.array([[1, 2], [3, 4]], dtype=np.int64)), [[[1, 2], [3, 4]]])
assert c.all_gather_object({'k': (1, 'x')}) == [{'k': (1, 'x')}]
got = c.all_to_all_host([np.arange(12, dtype=np.float32)])
assert len(got) == 1 and np.array_equal(got[0], np.arange(12, dtype=np.float32))
# device all-to-all (whole, a strided piece on the communicator's stream) and the ring exchange
n = 1 << 16
src, dst = HipBuf(n), HipBuf(2 * n)
x = np.random.default_rng(1).random(n, dtype=np.float32)
src.set(0, x)
c.all_to_all(None, src, dst, n)
_lib.sync()
assert np.array_equal(dst.get(0, n), x)
dst.set(0, np.zeros(2 * n, dtype=np.float32))
c.all_to_all_piece(None, src, dst, n, 1024, 4096, overlap=True)
c.join()
_lib.sync()
out = dst.get(0, n)
assert np.array_equal(out[1024:5120], x[1024:5120]) and not out[:1024].any() and not out[5120:].any()
# a piece of a transpose with peer blocks of different sizes (the compact transpose): offsets / counts per peer, async form
dst.set(0, np.zeros(2 * n, dtype=np.float32))
c.all_to_all_piece_v(None, src, dst, [512], [3000], [2048], [3000], overlap=True)
c.join()
_lib.sync()
out = dst.get(0, n)
assert np.array_equal(out[2048:5048], x[512:3512]) and not out[:2048].any() and not out[5048:].any()
c.ring_exchange(None, src, 0, n // 2, dst, n // 2)
_lib.sync()
assert np.array_equal(dst.get(0, n // 2), x[:n // 2]) and np.array_equal(dst.get(n // 2, n // 2), x[n // 2:])
# routing on the device keeps every particle (one rank owns the box) in a stable order
pos = (np.random.default_rng(2).random((5000, 3), dtype=np.float32) * 3 - 1) * np.float32(100.0)
w = np.random.default_rng(3).random(5000, dtype=np.float32)
rp, rw = c.route_particles(_lib.DeviceArray(pos), _lib.DeviceArray(w), 100.0)
assert np.array_equal(rp.get(), pos) and np.array_equal(rw.get(), w)
# the slab-decomposed P(k) through THIS communicator (device routing, ring exchange of the ghost planes, chunked
# all-to-all on the communicator's stream, all-reduce of the histogram) against the single-GPU calc_power: the per-GPU
# pipeline of BASELINE config 4, at 512^3 and at the full 2048^3 of the north star
from abacusutils_amd.analysis.slab_power import HipSlabBackend, calc_power_slab
from abacusutils_amd.analysis.power_spectrum import calc_power
# 512: plain three-pass form; 1024 / 2048: the fused form with the last pass binning straight from LDS (2048: compensated;
# 1024 a second time with `slab_nofuse` = the plain form as comparator)
for nmesh, n, nofuse in ((512, 3_000_000, 0), (1024, 5_000_000, 0), (1024, 5_000_000, 1), (2048, 20_000_000, 0)):
    L = 2000.0
    pos = np.random.default_rng(11).random((n, 3), dtype=np.float32) * np.float32(L)
    pos[:, 0] -= np.float32(L / 2)                      # x outside [0, L): routing wraps it
    kw = dict(kbins=min(256, nmesh // 2), mubins=4, k_max=np.pi * nmesh / L, paste='TSC', nmesh=nmesh, compensated=nmesh != 1024,
              interlaced=(nmesh == 512), poles=[0, 2, 4])
    dpos, _ = c.route_particles(_lib.DeviceArray(pos), None, L, fold=True)
    _lib.set_option('slab_nofuse', nofuse)
    assert _lib.lib().abacus_slab_fused(nmesh) == (1 if nmesh >= 1024 and not nofuse else 0)
    tab = calc_power_slab(dpos, L, comm=c, backend=HipSlabBackend(), n_total=n, **kw)
    _lib.set_option('slab_nofuse', 0)
    ref = calc_power(pos.copy(), L, **kw)
    assert np.array_equal(np.asarray(tab['N_mode']), np.asarray(ref['N_mode'])), nmesh
    ok = np.asarray(ref['N_mode']) > 0
    rel = np.abs(np.asarray(tab['power'])[ok] / np.asarray(ref['power'])[ok] - 1).max()
    scale = np.abs(np.asarray(ref['poles'])).max()
    dpole = np.abs(np.asarray(tab['poles']) - np.asarray(ref['poles'])).max() / scale
    assert rel < 1e-5 and dpole < 1e-5, (nmesh, rel, dpole)
    dpos.free()
    _lib.lib().abacus_power_release()
    print('SLAB-OK', nmesh, rel, dpole)
c.barrier()
c.free()
print('COMM-OK')
'''


def test_rccl_single_rank_collectives():
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', CODE % repo], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'COMM-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize('fold', [False, True])
def test_route_buckets_match_numpy(fold):
    """abacus_slab_route_dev: owners and the stable order inside a bucket equal the host formula of route_particles"""
    import ctypes as C

    from abacusutils_amd import _lib
    from abacusutils_amd.analysis.slab_power import slab_owner
    n, W, L = 200000, 8, 500.0
    pos = (np.random.default_rng(5).random((n, 3), dtype=np.float32) * 3 - 1) * np.float32(L)
    w = np.random.default_rng(6).random(n, dtype=np.float32)
    dp, dw = _lib.DeviceArray(pos), _lib.DeviceArray(w)
    op = _lib.DeviceArray(nbytes=n * 12, dtype=np.float32, shape=(n, 3))
    ow = _lib.DeviceArray(nbytes=n * 4, dtype=np.float32, shape=(n,))
    counts = np.zeros(W, dtype=np.int64)
    _lib.check(_lib.lib().abacus_slab_route_dev(dp.ptr, C.c_int64(n), dw.ptr, C.c_double(L), W, int(fold), op.ptr, ow.ptr,
                                                _lib.ptr(counts)))
    xw = pos[:, 0] - np.floor(pos[:, 0] / np.float32(L)) * np.float32(L)
    owner = slab_owner(xw, L, W, fold)
    order = np.argsort(owner, kind='stable')
    np.testing.assert_array_equal(counts, np.bincount(owner, minlength=W))
    np.testing.assert_array_equal(op.get(), pos[order])
    np.testing.assert_array_equal(ow.get(), w[order])
