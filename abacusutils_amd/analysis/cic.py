"""MI355X drop-in for `abacusnbody.analysis.cic.cic_serial` (reference: abacusnbody/analysis/cic.py:13-125):
float64 weights, float32 grid accumulated in place, positions are not wrapped (cell indices are)."""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import check, ptr

_DT = {np.dtype('f4'): 0, np.dtype('f8'): 1}


def cic_serial(positions, density, boxsize, weights=None):
    if density.dtype != np.float32 or density.ndim != 3 or not density.flags.c_contiguous:
        raise ValueError('density must be a C-contiguous float32 (gx, gy, gz) array')
    pos = np.ascontiguousarray(positions)
    if pos.dtype not in _DT:
        raise TypeError('positions must be float32 or float64')
    w = None if weights is None else np.ascontiguousarray(weights, dtype=pos.dtype)
    gx, gy, gz = density.shape
    check(_lib.lib().abacus_cic_deposit(ptr(pos), C.c_int64(len(pos)), ptr(w), _DT[pos.dtype], ptr(density),
                                        int(gx), int(gy), int(gz), C.c_double(boxsize)))
