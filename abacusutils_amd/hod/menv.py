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
                      batch_size=DEFAULT_BATCH_SIZE, out=None, r_outer_max=None):
    """Difference in total neighbour halo mass at two apertures.  Neighbour mass includes all halos, but only halos
    above mcut are used as centres (0 returned for all others).  Returns an array like `mass`.

    Extension: `pos`, `mass`, per-halo `r_inner` / `r_outer` may be `_lib.DeviceArray`s already resident in HBM (float32 or
    float64) and `out` a float64 DeviceArray for the result - the call is then the three kernels alone (prepare_sim keeps the
    halo table on the device).  A per-halo device `r_outer` needs `r_outer_max` (the largest radius of a centre); an open
    geometry (halo_lc) with a device `pos` needs host copies of nothing else - the bounding box is taken from `pos.get()`
    only in that case."""
    dev = _lib.DeviceArray
    on_dev = isinstance(pos, dev)
    if not on_dev:
        pos = np.asarray(pos)
    if not isinstance(mass, dev):
        mass = np.asarray(mass)
    if len(pos.shape) != 2 or pos.shape[1] != 3 or len(mass) != len(pos):
        raise ValueError('pos must be (N,3) and mass (N,)')
    periodic = not halo_lc   # then the kernels apply `(pos + Lbox / 2.0) % Lbox` (:39) in the dtype of pos
    n = len(pos)
    if n == 0:
        return np.zeros(0, dtype=mass.dtype)

    def as_real(a):
        if isinstance(a, dev):
            if a.dtype not in (np.float32, np.float64):
                raise TypeError('device arrays must be float32 or float64')
            return a
        a = np.ascontiguousarray(a)
        return a if a.dtype in (np.float32, np.float64) else a.astype(np.float64)

    pos, massf = as_real(pos), as_real(mass)

    def radius(r, name):
        """the tree query compares float64 distances with float64 radii: a float32 radius is widened exactly on the device"""
        if isinstance(r, dev):
            if len(r) != n:
                raise ValueError(f'{name} must be a scalar or have one value per halo')
            return as_real(r)
        r = np.asarray(r)
        if r.ndim > 0 and len(r) != n:
            raise ValueError(f'{name} must be a scalar or have one value per halo')
        return np.ascontiguousarray(np.atleast_1d(r), dtype=r.dtype if r.dtype in (np.float32, np.float64) else np.float64)

    ri, ro = radius(r_inner, 'r_inner'), radius(r_outer, 'r_outer')
    if ri.dtype != ro.dtype:     # one precision flag for both in the C ABI: a host scalar follows the other argument ...
        if not isinstance(ri, dev) and len(ri) == 1 and float(ri.astype(ro.dtype)[0]) == float(ri[0]):
            ri = ri.astype(ro.dtype)
        elif not isinstance(ro, dev) and len(ro) == 1 and float(ro.astype(ri.dtype)[0]) == float(ro[0]):
            ro = ro.astype(ri.dtype)
        elif isinstance(ri, dev) or isinstance(ro, dev):
            raise TypeError('device r_inner / r_outer must share a dtype')
        else:                    # ... otherwise both are widened
            ri, ro = ri.astype(np.float64), ro.astype(np.float64)
    # `mass > mcut` (:43) compares in the dtype of mass: a Python-float mcut is rounded to float32 for float32 masses
    mcut = float(np.asarray(mcut, dtype=massf.dtype)) if np.ndim(mcut) == 0 and isinstance(mcut, (int, float)) else float(mcut)
    # only centres' radii matter for the cell size (:48-54)
    if r_outer_max is not None:
        ro_max = float(r_outer_max)
    elif len(ro) == 1 and not isinstance(ro, dev):
        ro_max = float(ro[0])
    else:
        roh = ro.get() if isinstance(ro, dev) else ro
        mmask = (massf.get() if isinstance(massf, dev) else massf) > mcut
        ro_max = float(roh[mmask].max()) if mmask.any() else 0.0
    if periodic:
        lo = hi = None
    else:
        ph = pos.get() if on_dev else pos
        lo = np.ascontiguousarray(ph.min(axis=0), dtype=np.float64)
        hi = np.ascontiguousarray(ph.max(axis=0), dtype=np.float64)
    p_ = lambda a: a.ptr if isinstance(a, dev) else _lib.ptr(a)   # noqa: E731
    if out is not None:
        if not isinstance(out, dev) or out.dtype != np.float64 or len(out) != n:
            raise TypeError('out must be a float64 DeviceArray with one value per halo')
        Menv = out
    else:
        Menv = np.empty(n, dtype=np.float64)
    _lib.check(_lib.lib().abacus_menv(
        p_(pos), int(pos.dtype == np.float64), p_(massf), int(massf.dtype == np.float64), C.c_int64(n),
        p_(ri), C.c_int64(len(ri)), p_(ro), C.c_int64(len(ro)), int(ri.dtype == np.float64),
        C.c_double(max(ro_max, 0.0)), C.c_double(float(Lbox) if periodic else 0.0), int(periodic), _lib.ptr(lo),
        _lib.ptr(hi), C.c_double(float(mcut)), p_(Menv)))
    if out is not None:
        return out
    # the kernel already returns 0 for halos at or below mcut (:84-85)
    return Menv if mass.dtype == np.float64 else Menv.astype(mass.dtype)
