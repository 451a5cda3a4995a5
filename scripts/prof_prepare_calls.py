import sys, time, collections
sys.path.insert(0, '.')
import numpy as np
from abacusutils_amd import _lib
from abacusutils_amd.hod import prepare_sim as prep
from abacusutils_amd.synth import synth_compaso_slabs
_lib.set_device(0)
slabs, header = synth_compaso_slabs(numslabs=1, n_halo=1_000_000, seed=900, lbox=2000.0, subsample_frac=0.006)
halos, parts = slabs[0]['halos'], slabs[0]['parts']
Mpart, h = header['ParticleMassHMsun'], header['H0'] / 100.0
kw = dict(MT=True, want_ranks=True, want_AB=True, Lbox=header['BoxSize'])
prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=7, **kw)
L = _lib.lib()
acc = collections.OrderedDict()
class W:
    def __init__(s, name, f): s.name, s.f = name, f
    def __call__(s, *a):
        t = time.perf_counter(); r = s.f(*a); acc[s.name] = acc.get(s.name, 0) + time.perf_counter() - t; return r
names = [n for n in dir(L) if n.startswith('abacus_')]
import ctypes
class Proxy:
    def __getattr__(s, n):
        f = getattr(L, n)
        return W(n, f) if n.startswith('abacus_') else f
_lib_lib = _lib.lib
_lib.lib = lambda: Proxy()
t0 = time.perf_counter()
prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=8, **kw)
print('total', (time.perf_counter() - t0) * 1e3)
for k, v in acc.items(): print(k, round(v * 1e3, 2))
