"""pack9 particle data (pos + vel in 9 bytes, cell headers in the stream) on the MI355X.

Drop-in for abacusnbody/data/pack9.py:16-56 `unpack_pack9`; the serial record walk `_unpack_pack9` (:59-123) is a
two-pass chunked HIP kernel behind `abacus_unpack_pack9` (include/abacus_hip.h).  No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib

__all__ = ['unpack_pack9']


def unpack_pack9(data, boxsize, velzspace_to_kms, float_dtype=np.float32, posout=None, velout=None):
    """Returns (pos, vel): arrays of the `npart` particles (cell headers removed) when allocated here, 0 for an output
    given as False, `npart` for an output array given by the caller (its first npart rows are filled)."""
    data = np.ascontiguousarray(np.asanyarray(data, dtype=np.ubyte))
    if data.ndim != 2 or data.shape[1] != 9:
        raise ValueError('pack9 data must have shape (N, 9)')
    Nmax = len(data)  # some pack9s will be cell headers
    float_dtype = np.dtype(float_dtype)
    if float_dtype not in (np.float32, np.float64):
        raise TypeError('float_dtype must be float32 or float64')

    def resolve(out):
        if out is None:
            return np.empty((Nmax, 3), dtype=float_dtype)
        if out is False:
            return None
        if out.dtype != float_dtype or not out.flags.c_contiguous or out.size < Nmax * 3:
            raise ValueError('output arrays must be C-contiguous (N, 3) arrays of float_dtype')
        return out

    _pos, _vel = resolve(posout), resolve(velout)
    npart = C.c_int64(0)
    _lib.check(_lib.lib().abacus_unpack_pack9(_lib.ptr(data), C.c_int64(Nmax), C.c_double(float(boxsize)),
                                              C.c_double(float(velzspace_to_kms)), int(float_dtype == np.float64),
                                              _lib.ptr(_pos), _lib.ptr(_vel), C.byref(npart)))
    npart = npart.value
    ret = []
    for given, made in ((posout, _pos), (velout, _vel)):
        ret.append(made[:npart] if given is None else (0 if given is False else npart))
    return tuple(ret)
