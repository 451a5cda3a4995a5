"""MI355X drop-in for `abacusnbody.analysis.tsc` (reference: abacusnbody/analysis/tsc.py).

`tsc_parallel` and `partition_parallel` keep the reference signatures, return values and side effects:
positions are wrapped IN PLACE (tsc.py:171-173), a user-supplied grid is accumulated into and `None` is
returned (tsc.py:45-50,204-206), bad `npartition` values raise the same ValueError (tsc.py:141-147).
The deposit itself runs on the GPU (csrc/tsc.hip, LDS-tiled, no stripes): `nthread`, `npartition`, `sort` only
take part in the argument checks.
"""
import ctypes as C
import os
import warnings

import numpy as np

from .. import _lib
from .._lib import check, ptr

__all__ = ['tsc_parallel', 'partition_parallel']

_DT = {np.dtype('f4'): 0, np.dtype('f8'): 1}


def _nthreads(nthread):
    return nthread if nthread >= 0 else (os.cpu_count() or 1)


def tsc_parallel(pos, densgrid, box, weights=None, nthread=-1, wrap=True, npartition=None, sort=False, coord=0,
                 verbose=False, offset=0.0):
    """TSC mass assignment; see abacusnbody/analysis/tsc.py:10-206 for the parameter documentation."""
    nthread = _nthreads(nthread)
    if isinstance(densgrid, (int, np.integer)):
        densgrid = (densgrid, densgrid, densgrid)
    if isinstance(densgrid, tuple):
        densgrid = np.zeros(densgrid, dtype=np.float32)
        user_supplied_grid = False
    else:
        user_supplied_grid = True
    if densgrid.ndim != 3:
        # the reference's _tsc_scatter indexes a 2-D grid with three indices (tsc.py:471) and fails to compile
        raise ValueError('densgrid must be 3-D')
    n1d = densgrid.shape[coord]

    # The stripes of the reference do not exist here; an explicit `npartition` is still held to its rules (tsc.py:141-147:
    # at most ngrid//3 stripes, or exactly ngrid//2, and an even number of them).  The default the reference derives from the
    # thread count (tsc.py:126-139) satisfies them by construction, so nothing is computed for it.
    if npartition and nthread > 1:
        third, half = n1d // 3, n1d // 2
        if npartition > third and npartition != half:
            raise ValueError(f'npartition {npartition} must be less than ngrid//3 = {third} or equal to ngrid//2 = {half}')
        if npartition > 1 and npartition % 2:
            raise ValueError(f'npartition {npartition} not divisible by 2')

    for name, arr in (('pos', pos), ('densgrid', densgrid), ('weights', weights)):    # the reference's advice (tsc.py:149-165)
        if arr is not None and arr.itemsize > 4:
            warnings.warn(f'{name}.dtype={arr.dtype} instead of np.float32. float32 is recommended for performance.')
    if pos.dtype not in _DT or densgrid.dtype not in _DT:
        raise TypeError('pos and densgrid must be float32 or float64')
    if not (pos.flags.c_contiguous and pos.flags.writeable) and wrap:
        raise ValueError('pos must be a writeable C-contiguous array (it is wrapped in place)')
    if not densgrid.flags.c_contiguous:
        raise ValueError('densgrid must be C-contiguous')
    posc = np.ascontiguousarray(pos)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=pos.dtype)
    gx, gy, gz = densgrid.shape
    check(_lib.lib().abacus_tsc_deposit(ptr(posc), C.c_int64(len(posc)), ptr(w), _DT[pos.dtype], ptr(densgrid),
                                        int(gx), int(gy), int(gz), _DT[densgrid.dtype], C.c_double(box),
                                        C.c_double(offset), int(bool(wrap))))
    if user_supplied_grid:
        return None
    return densgrid


def partition_parallel(pos, npartition, boxsize, weights=None, coord=0, nthread=-1, sort=False):
    """Stable partition of the positions into `npartition` stripes along `coord`
    (abacusnbody/analysis/tsc.py:259-384).  Returns (partitioned, part_starts int64[npartition+1], wpart)."""
    assert pos.shape[1] == 3
    if pos.dtype not in _DT:
        raise TypeError('pos must be float32 or float64')
    posc = np.ascontiguousarray(pos)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=pos.dtype)
    psort = np.empty_like(posc)
    wsort = None if w is None else np.empty_like(w)
    starts = np.empty(npartition + 1, dtype=np.int64)
    check(_lib.lib().abacus_partition(ptr(posc), C.c_int64(len(posc)), ptr(w), _DT[pos.dtype], int(npartition),
                                      C.c_double(boxsize), int(coord), ptr(psort), ptr(starts), ptr(wsort)))
    if sort:  # tsc.py:361-382: sort on `coord` inside every stripe (argsort is not stable in the reference either)
        for i in range(npartition):
            part = psort[starts[i]:starts[i + 1]]
            iord = part[:, coord].argsort()
            part[:] = part[iord]
            if wsort is not None:
                wsort[starts[i]:starts[i + 1]] = wsort[starts[i]:starts[i + 1]][iord]
    return psort, starts, wsort
