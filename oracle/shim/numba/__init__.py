"""Identity-decorator stand-in for `numba`, used ONLY by oracle/make_golden.py.

TEST INFRASTRUCTURE - never imported by the product path.

The reference hot path (abacusnbody/hod/GRAND_HOD.py, analysis/tsc.py,
analysis/power_spectrum.py) is plain Python under @njit decorators.  numba is
not importable in the build container, so this module makes the decorators
no-ops; the reference functions then execute as IEEE-strict CPython/NumPy.
That is how the golden vectors under tests/golden/ were produced (see
oracle/make_golden.py).  Nothing in here is derived from numba's sources.
"""
import types as _pytypes

import numpy as _np


def _identity_decorator(*dargs, **dkwargs):
    # @njit  or  @njit(parallel=True, fastmath=True)
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        f = dargs[0]
        try:
            f.py_func = f
        except AttributeError:
            pass
        return f

    def deco(f):
        try:
            f.py_func = f
        except AttributeError:
            pass
        return f

    return deco


njit = jit = _identity_decorator


def vectorize(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return _np.vectorize(dargs[0])

    def deco(f):
        return _np.vectorize(f)

    return deco


prange = range


def set_num_threads(n):
    pass


def get_num_threads():
    return 1


def get_thread_id():
    return 0


class _Config:
    NUMBA_NUM_THREADS = 1


config = _Config()


class _AnyType:
    """numba.types.* placeholder: indexable, callable, inert."""

    def __getitem__(self, item):
        return self

    def __call__(self, *a, **k):
        return self


class _Types(_pytypes.ModuleType):
    def __getattr__(self, name):
        return _AnyType()


types = _Types('numba.types')


class _TypedDict(dict):
    @classmethod
    def empty(cls, key_type=None, value_type=None):
        return cls()


class _Typed(_pytypes.ModuleType):
    Dict = _TypedDict


typed = _Typed('numba.typed')

import sys as _sys

_sys.modules.setdefault('numba.types', types)
_sys.modules.setdefault('numba.typed', typed)
