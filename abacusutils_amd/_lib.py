"""ctypes binding of libabacus_hip.so (C ABI: include/abacus_hip.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is
visible when a compute entry point is called, the call raises.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / 'libabacus_hip.so'
_lib = None


class AbacusHipError(RuntimeError):
    pass


_D = C.c_double
_LRG_KEYS = ['logM_cut', 'logM1', 'sigma', 'alpha', 'kappa', 'alpha_c', 'alpha_s', 's', 's_v', 's_p', 's_r',
             'Acent', 'Asat', 'Bcent', 'Bsat', 'ic']
_ELG_KEYS = ['p_max', 'Q', 'logM_cut', 'kappa', 'sigma', 'logM1', 'alpha', 'gamma', 'A_s', 'alpha_c', 'alpha_s',
             's', 's_v', 's_p', 's_r', 'Acent', 'Asat', 'Bcent', 'Bsat', 'Ccent', 'Csat', 'ic',
             'logM1_EE', 'alpha_EE', 'logM1_EL', 'alpha_EL']
_QSO_KEYS = ['logM_cut', 'kappa', 'sigma', 'logM1', 'alpha', 'alpha_c', 'alpha_s', 's', 's_v', 's_p', 's_r',
             'Acent', 'Asat', 'Bcent', 'Bsat', 'ic']
TRACER_KEYS = {'LRG': ('L_', _LRG_KEYS), 'ELG': ('E_', _ELG_KEYS), 'QSO': ('Q_', _QSO_KEYS)}


class HodParams(C.Structure):
    """struct abacus_hod_params"""
    _fields_ = (
        [(n, C.c_int32) for n in ('want_LRG', 'want_ELG', 'want_QSO', 'rsd', 'has_origin', 'enable_ranks',
                                  'pad0', 'pad1')]
        + [('inv_velz2kms', _D), ('lbox', _D), ('origin', _D * 3)]
        + [('L_' + k, _D) for k in _LRG_KEYS]
        + [('E_' + k, _D) for k in _ELG_KEYS]
        + [('Q_' + k, _D) for k in _QSO_KEYS]
    )


_HALO_F8 = ('hpos', 'hvel', 'hmass')
_P = C.c_void_p


class HodArrays(C.Structure):
    """struct abacus_hod_arrays"""
    _fields_ = (
        [('n_halo', C.c_int64)]
        + [(k, _P) for k in ('hpos', 'hvel', 'hmass', 'hid', 'hmultis', 'hrandoms', 'hveldev', 'hdeltac', 'hfenv',
                             'hshear')]
        + [('n_part', C.c_int64)]
        + [(k, _P) for k in ('ppos', 'pvel', 'phvel', 'phmass', 'phid', 'pweights', 'prandoms', 'pdeltac', 'pfenv',
                             'pshear', 'pranks', 'pranksv', 'pranksp', 'pranksr', 'pinds')]
    )


class NfwParams(C.Structure):
    """struct abacus_nfw_params"""
    _fields_ = [('seed', C.c_uint64), ('f_sigv', C.c_double * 3), ('exp_frac', C.c_double), ('exp_scale', C.c_double),
                ('nfw_rescale', C.c_double), ('halo_index0', C.c_int64)]


def available():
    """True if the shared library exists (says nothing about a GPU being present)."""
    return _SO.exists()


def lib():
    global _lib
    if _lib is None:
        if not _SO.exists():
            raise AbacusHipError(
                f'{_SO} not found: build it with `make -C {_HERE / "csrc"}` (or __graft_entry__.build()). '
                'abacusutils_amd has no CPU fallback.')
        L = C.CDLL(str(_SO))
        L.abacus_last_error.restype = C.c_char_p
        L.abacus_get_stream.restype = C.c_void_p
        L.abacus_power_geometry_ms.restype = C.c_double
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise AbacusHipError(lib().abacus_last_error().decode())


class _ArrayPtr(C.c_void_p):
    """void* that keeps its array alive: `ptr(x.copy())` or `ptr(a + b)` inside a call would otherwise hand the library the
    address of an array freed before the call starts"""


def ptr(a):
    """void* of a C-contiguous ndarray (None -> NULL); the array lives at least as long as the returned object"""
    if a is None:
        return None
    assert a.flags.c_contiguous
    p = _ArrayPtr(a.ctypes.data)
    p._keep = a
    return p


def device_name():
    buf = C.create_string_buffer(256)
    check(lib().abacus_device_name(buf, 256))
    return buf.value.decode()


def device_count():
    n = C.c_int(0)
    try:
        check(lib().abacus_device_count(C.byref(n)))
    except AbacusHipError:
        return 0
    return n.value


def set_device(i):
    check(lib().abacus_set_device(int(i)))


def sync():
    check(lib().abacus_device_sync())


class DeviceArray:
    """A raw HBM allocation holding a copy of a NumPy array (bench / tests keep inputs resident this way)."""

    def __init__(self, host=None, nbytes=None, dtype=None, shape=None):
        self.ptr = C.c_void_p()
        if host is not None:
            host = np.ascontiguousarray(host)
            nbytes, dtype, shape = host.nbytes, host.dtype, host.shape
        self.nbytes, self.dtype, self.shape = int(nbytes), np.dtype(dtype), tuple(shape)
        check(lib().abacus_malloc(C.byref(self.ptr), C.c_uint64(self.nbytes)))
        if host is not None and self.nbytes:
            check(lib().abacus_memcpy_h2d(self.ptr, ptr(host), C.c_uint64(self.nbytes)))

    @classmethod
    def view(cls, ptr_, dtype, shape):
        """a NON-OWNING view of device memory the library manages (e.g. a column of a staged catalogue): free() is a
        no-op, the caller must not use it after the owner rewrote or released the memory"""
        self = cls.__new__(cls)
        self.ptr = C.c_void_p(ptr_ if isinstance(ptr_, int) else ptr_.value)
        self.dtype, self.shape = np.dtype(dtype), tuple(shape)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._view = True
        return self

    def __len__(self):
        return self.shape[0]

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            check(lib().abacus_memcpy_d2h(ptr(out), self.ptr, C.c_uint64(self.nbytes)))
        return out

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes == self.nbytes
        check(lib().abacus_memcpy_h2d(self.ptr, ptr(host), C.c_uint64(self.nbytes)))

    def free(self):
        if getattr(self, '_view', False):
            self.ptr = C.c_void_p()
            return
        if self.ptr:
            lib().abacus_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PinnedPool:
    """page-locked host blocks (abacus_host_alloc) handed out as NumPy arrays and recycled: a block returns to the pool
    when the last array viewing it is garbage-collected.  Device-to-host copies of the galaxy catalogues land in these
    (one DMA, no page faults of a fresh allocation).  Blocks are power-of-two sized; at most `cap_bytes` stay pinned
    (beyond that, or if pinning fails, plain NumPy memory is returned)."""

    def __init__(self, cap_bytes=2 << 30):
        self.free = {}          # size class -> [address]
        self.total = 0
        self.cap = cap_bytes

    def _release(self, addr, size):
        self.free.setdefault(size, []).append(addr)

    def empty(self, shape, dtype):
        import weakref
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if nbytes == 0:
            return np.empty(shape, dtype=dtype)
        size = 1 << max(12, (nbytes - 1).bit_length())
        pool = self.free.get(size)
        if pool:
            addr = pool.pop()
        else:
            if self.total + size > self.cap:
                return np.empty(shape, dtype=dtype)
            p = C.c_void_p()
            if lib().abacus_host_alloc(C.byref(p), C.c_uint64(size)) != 0 or not p.value:
                return np.empty(shape, dtype=dtype)
            addr = p.value
            self.total += size
        buf = (C.c_char * size).from_address(addr)
        weakref.finalize(buf, self._release, addr, size)     # the array below keeps `buf` alive through its base chain
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def drain(self):
        for size, pool in self.free.items():
            for addr in pool:
                lib().abacus_host_free(C.c_void_p(addr))
                self.total -= size
        self.free = {}


_pinned = _PinnedPool()


def pinned_empty(shape, dtype=np.float64):
    """uninitialised array in page-locked host memory (recycled by the pool above), or plain memory as a fallback"""
    return _pinned.empty(shape, dtype)


def poshash_host(a):
    """sum_i word[i] * (2 i + 1) mod 2^64 of a contiguous array of 8-byte values (the host side of abacus_poshash_u64)"""
    w = np.ascontiguousarray(a).view(np.uint64).ravel()
    with np.errstate(over='ignore'):
        return int((w * (np.arange(w.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))


def poshash_device(dptr, n):
    out = C.c_uint64(0)
    dptr = dptr if isinstance(dptr, C.c_void_p) else C.c_void_p(dptr)    # a bare int would be passed as a 32-bit C int
    check(lib().abacus_poshash_u64(dptr, C.c_int64(int(n)), C.byref(out)))
    return int(out.value)


class Event:
    def __init__(self):
        self.ev = C.c_void_p()
        check(lib().abacus_event_create(C.byref(self.ev)))

    def record(self):
        check(lib().abacus_event_record(self.ev))

    def elapsed_ms_since(self, start):
        ms = C.c_float(0)
        check(lib().abacus_event_elapsed_ms(start.ev, self.ev, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib().abacus_event_destroy(self.ev)
        except Exception:
            pass


def set_option(name, value=1):
    """diagnostic option of the library (abacus_set_option): comparator paths for tests / A-B timing"""
    check(lib().abacus_set_option(name.encode(), int(value)))


def get_option(name):
    """current value of a diagnostic option (0 when never set)"""
    return int(lib().abacus_get_option(name.encode()))


def scratch_release():
    """free the library's idle scratch blocks (abacus_scratch_release): temporaries kept between calls"""
    check(lib().abacus_scratch_release())


def profile_enable(on=True):
    check(lib().abacus_profile_enable(int(on)))


def profile_select(name=None):
    check(lib().abacus_profile_select(None if not name else name.encode()))


def profile_reset():
    check(lib().abacus_profile_reset())


def profile_get():
    """{kernel name: (total ms, launches)} measured with HIP events on the library stream"""
    cap = 64
    names = (C.c_char_p * cap)()
    ms = (C.c_double * cap)()
    n = (C.c_int64 * cap)()
    k = lib().abacus_profile_get(names, ms, n, cap)
    return {names[i].decode(): (ms[i], n[i]) for i in range(min(k, cap))}
