"""Abacus bit-packed particle formats on the MI355X: RVint and the fields encoded in the PIDs.

Drop-in for abacusnbody/data/bitpacked.py (`unpack_rvint` :32-97, `unpack_pids` :118-221, `empty_bitpacked_arrays`
:224-271): same arguments, return structures and exceptions; the per-particle loops (`_unpack_rvint` :100-116,
`_unpack_pids` :274-330) run as HIP kernels behind `abacus_unpack_rvint` / `abacus_unpack_pids`
(include/abacus_hip.h).  Inputs and outputs may be NumPy arrays or `_lib.DeviceArray`s (then nothing crosses PCIe).
No CPU fallback.
"""
import ctypes as C

import numpy as np

from .. import _lib

__all__ = ['unpack_rvint', 'unpack_pids']

PID_FIELDS = ['pid', 'lagr_pos', 'tagged', 'density', 'lagr_idx', 'packedpid']


def _is_dev(a):
    return isinstance(a, _lib.DeviceArray)


def _p(a):
    if a is None:
        return None
    return a.ptr if _is_dev(a) else _lib.ptr(a)


def _check_out(a, n, width, what):
    if a.dtype not in (np.float32, np.float64):
        raise TypeError(f'{what} must be float32 or float64')
    if not _is_dev(a) and not a.flags.c_contiguous:
        raise ValueError(f'{what} must be C-contiguous')
    size = int(np.prod(a.shape))
    if size != n * width:
        raise ValueError(f'{what} has {size} elements, expected {n * width}')


def unpack_rvint(intdata, boxsize, float_dtype=np.float32, posout=None, velout=None):
    """Unpack rvint data into pos and vel (reference :32-97).

    posout / velout: None (allocate and return), False (skip; 0 is returned in its place) or an array to fill
    (then the particle count is returned in its place)."""
    if _is_dev(intdata):
        assert intdata.dtype == np.int32
        N = int(np.prod(intdata.shape)) // 3
        src = intdata
    else:
        intdata = intdata.reshape(-1, 3)
        assert intdata.dtype == np.int32
        N = len(intdata)
        src = np.ascontiguousarray(intdata)
    float_dtype = np.dtype(float_dtype)

    def resolve(out, what):
        if out is None:
            if float_dtype not in (np.float32, np.float64):
                raise TypeError('float_dtype must be float32 or float64')
            return np.empty((N, 3), dtype=float_dtype)
        if out is False:
            return None
        _check_out(out, N, 3, what)
        return out

    _pos, _vel = resolve(posout, 'posout'), resolve(velout, 'velout')
    dts = {a.dtype for a in (_pos, _vel) if a is not None}
    if len(dts) > 1:
        raise TypeError('posout and velout must have the same dtype')
    if dts:
        f64 = int(dts.pop() == np.float64)
        _lib.check(_lib.lib().abacus_unpack_rvint(_p(src), C.c_int64(N), C.c_double(float(boxsize)), f64,
                                                  _p(_pos), _p(_vel)))
    ret = []
    for given, made in ((posout, _pos), (velout, _vel)):
        ret.append(made if given is None else (0 if given is False else N))
    return tuple(ret)


# field -> (dtype or None for the caller's float dtype, columns per particle); one table serves both allocators
_FIELD_LAYOUT = {'pid': (np.int64, 1), 'lagr_pos': (None, 3), 'lagr_idx': (np.int16, 3), 'tagged': (np.uint8, 1),
                 'density': (None, 1), 'packedpid': (np.uint64, 1)}


def _alloc(N, fields, float_dtype):
    out = {}
    for name in _FIELD_LAYOUT:                       # the table's order, whatever order the caller listed
        if name in fields:
            dt, width = _FIELD_LAYOUT[name]
            out[name] = np.empty(N if width == 1 else (N, width), dtype=float_dtype if dt is None else dt)
    return out


def empty_bitpacked_arrays(N, unpack_bits, float_dtype=np.float32):
    """Uninitialised output arrays for the bit-packed fields (reference :224-271): `unpack_bits` is True (every field of
    PID_FIELDS), False (the pid alone), one field name or a list of names"""
    wanted = PID_FIELDS if unpack_bits is True else ['pid'] if unpack_bits is False else np.atleast_1d(unpack_bits).tolist()
    return _alloc(N, wanted, float_dtype)


def unpack_pids(packed, box=None, ppd=None, pid=False, lagr_pos=False, tagged=False, density=False, lagr_idx=False,
                float_dtype=np.float32):
    """Extract fields from bit-packed PIDs (reference :118-221).  Returns a dict of the requested arrays (a field is
    produced when its flag is exactly True, as in the reference)."""
    if _is_dev(packed):
        assert packed.dtype == np.uint64
    else:
        packed = np.ascontiguousarray(np.asanyarray(packed, dtype=np.uint64))
    if lagr_pos is not False and (box is None or ppd is None):
        raise ValueError('Must supply `box` if requesting `lagr_pos`' if box is None else
                         'Must supply `ppd` if requesting `lagr_pos`')
    if ppd is None:
        ppd = 1
    else:
        if not np.isclose(ppd, int(round(ppd))):
            raise ValueError(f'ppd "{ppd}" not valid int?')
        ppd = int(round(ppd))
    float_dtype = np.dtype(float_dtype)
    if float_dtype not in (np.float32, np.float64):
        raise TypeError('float_dtype must be float32 or float64')
    N = int(np.prod(packed.shape))
    flags = dict(pid=pid, lagr_pos=lagr_pos, lagr_idx=lagr_idx, tagged=tagged, density=density)
    arr = _alloc(N, [k for k, v in flags.items() if v is True], float_dtype)
    if arr:
        _lib.check(_lib.lib().abacus_unpack_pids(_p(packed), C.c_int64(N), C.c_double(1.0 if box is None else float(box)),
                                                 C.c_int64(ppd), int(float_dtype == np.float64), *(_p(arr.get(k)) for k in flags)))
    return arr
