"""Local mass environment on the MI355X (drop-in for abacusnbody/hod/menv.py:19-87 `do_Menv_from_tree`).

The reference builds a scipy KDTree of all halos, queries the neighbours inside `r_outer` and `r_inner` of every halo
above `mcut` in batches and sums their masses (`msum_core` :142-150).  Here the neighbour search is a cell list on the
device (cells >= max r_outer, 27-cell stencil, float64 distances and sums) behind `abacus_menv`
(include/abacus_hip.h); `nthread` and `batch_size` are accepted and ignored.  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib

__all__ = ['do_Menv_from_tree']

DEFAULT_BATCH_SIZE = 10**5


def do_Menv_from_tree(pos, mass, r_inner, r_outer, halo_lc, Lbox, nthread=1, mcut=1e11,
                      batch_size=DEFAULT_BATCH_SIZE):
    """Difference in total neighbour halo mass at two apertures.  Neighbour mass includes all halos, but only halos
    above mcut are used as centres (0 returned for all others).  Returns an array like `mass`."""
    pos = np.asarray(pos)
    mass = np.asarray(mass)
    if pos.ndim != 2 or pos.shape[1] != 3 or len(mass) != len(pos):
        raise ValueError('pos must be (N,3) and mass (N,)')
    periodic = not halo_lc   # then the kernels apply `(pos + Lbox / 2.0) % Lbox` (:39) in the dtype of pos
    n = len(pos)
    Menv = np.zeros(n, dtype=np.float64)
    if n == 0:
        return np.zeros_like(mass)

    def as_real(a):
        a = np.ascontiguousarray(a)
        return a if a.dtype in (np.float32, np.float64) else a.astype(np.float64)

    pos, massf = as_real(pos), as_real(mass)
    ri, ro = np.asarray(r_inner), np.asarray(r_outer)
    for r, name in ((ri, 'r_inner'), (ro, 'r_outer')):
        if r.ndim > 0 and len(r) != n:
            raise ValueError(f'{name} must be a scalar or have one value per halo')
    rdt = np.float64   # the tree query compares float64 distances with float64 radii, whatever the caller's dtype
    ri = np.ascontiguousarray(np.atleast_1d(ri), dtype=rdt)
    ro = np.ascontiguousarray(np.atleast_1d(ro), dtype=rdt)
    # `mass > mcut` (:43) compares in the dtype of mass: a Python-float mcut is rounded to float32 for float32 masses
    mcut = float(np.asarray(mcut, dtype=massf.dtype)) if np.ndim(mcut) == 0 and isinstance(mcut, (int, float)) else float(mcut)
    # only centres' radii matter for the cell size (:48-54)
    if ro.size == 1:
        ro_max = float(ro[0])
    else:
        mmask = massf > mcut
        ro_max = float(ro[mmask].max()) if mmask.any() else 0.0
    if periodic:
        lo = hi = None
    else:
        lo = np.ascontiguousarray(pos.min(axis=0), dtype=np.float64)
        hi = np.ascontiguousarray(pos.max(axis=0), dtype=np.float64)
    _lib.check(_lib.lib().abacus_menv(
        _lib.ptr(pos), int(pos.dtype == np.float64), _lib.ptr(massf), int(massf.dtype == np.float64), C.c_int64(n),
        _lib.ptr(ri), C.c_int64(ri.size), _lib.ptr(ro), C.c_int64(ro.size), int(rdt == np.float64),
        C.c_double(max(ro_max, 0.0)), C.c_double(float(Lbox) if periodic else 0.0), int(periodic), _lib.ptr(lo),
        _lib.ptr(hi), C.c_double(float(mcut)), _lib.ptr(Menv)))
    # the kernel already returns 0 for halos at or below mcut (:84-85)
    return Menv if mass.dtype == np.float64 else Menv.astype(mass.dtype)
