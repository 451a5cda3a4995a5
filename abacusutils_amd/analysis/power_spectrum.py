r"""MI355X drop-in for `abacusnbody.analysis.power_spectrum` (reference: abacusnbody/analysis/power_spectrum.py).

Same public functions, signatures and return structures for the calc_power chain:

    calc_power            (:1131-1319)   deposit -> FFT -> binning without the mesh leaving HBM
    calc_pk_from_deltak   (:730-805)     binning of caller-supplied spectra (the zcv call pattern)
    get_k_mu_edges        (:663-704)
    get_field_fft         (:1001-1070)   returns the complex64 half-spectrum as a NumPy array
    get_field             (:808-857)     returns the float32 overdensity mesh
    get_W_compensated     (:1081-1128)
    normalize_field       (:860-901)

`nthread` arguments are accepted and ignored.  Multipoles are limited to even l <= 10 (the range `P_n` is
stated to be valid for, :124-125).  Sums are accumulated in float64 on the device (the reference uses per-thread
float32 accumulators, :221-229), so results agree with the reference to its own float32 accumulation error.
"""
import ctypes as C
import warnings

import numpy as np

from .. import _lib
from .._lib import check, ptr
from .cic import cic_serial
from .tsc import tsc_parallel

__all__ = ['calc_power', 'calc_power_spectrum', 'calc_pk_from_deltak', 'get_k_mu_edges', 'get_field_fft', 'get_field',
           'get_W_compensated', 'normalize_field', 'bin_kmu', 'get_raw_power', 'shift_field_fft',
           'get_interlaced_field_fft']

MAX_THREADS = 1

try:  # the reference returns an astropy Table; fall back to a dict with `.meta` and the same column access
    from astropy.table import Table
except Exception:  # astropy is not part of this image
    class Table(dict):
        def __init__(self, data=None, meta=None):
            super().__init__(data or {})
            self.meta = meta or {}

        @property
        def colnames(self):
            return list(self.keys())


_PASTE = {'TSC': 0, 'CIC': 1}


def _paste_code(paste, where):
    p = paste.upper()
    if p not in _PASTE:
        raise ValueError(f'Unknown pasting method{where} {paste}')
    return _PASTE[p]


def get_k_mu_edges(Lbox, k_max, kbins, mubins, logk):
    """Bin edges of k and mu (:663-704): ints become linspace/geomspace edges, array-likes pass through."""
    if isinstance(kbins, int):
        if logk:
            k_min = (1.0 - 1.0e-4) * 2.0 * np.pi / Lbox
            kbins = np.geomspace(k_min, k_max, kbins + 1)
        else:
            kbins = np.linspace(0.0, k_max, kbins + 1)
    if isinstance(mubins, int):
        mubins = np.linspace(0.0, 1.0, mubins + 1)
    return kbins, mubins


def get_W_compensated(Lbox, nmesh, paste, interlaced):
    """1-D float32 TSC/CIC window (:1081-1128)."""
    d = Lbox / nmesh
    kN = np.pi / d
    k = (np.fft.fftfreq(nmesh, d=d) * 2.0 * np.pi).astype(np.float32)
    paste = paste.upper()
    if interlaced:
        if paste == 'TSC':
            p = 3.0
        elif paste == 'CIC':
            p = 2.0
        else:
            raise ValueError(f'Unknown pasting method {paste}')
        W = np.sinc(0.5 * k / kN) ** p
    else:
        s = np.sin(0.5 * np.pi * k / kN) ** 2
        if paste == 'TSC':
            W = (1 - s + 2.0 / 15 * s**2) ** 0.5
        elif paste == 'CIC':
            W = (1 - 2.0 / 3 * s) ** 0.5
        else:
            raise ValueError(f'Unknown pasting method {paste}')
    return W


def normalize_field(field, tot_weight=None, inplace=False, nthread=MAX_THREADS):
    """overdens = field * dtype(field.size / tot_weight) - 1 (:860-901); small host helper kept for API parity
    (inside calc_power the normalisation is fused into the deposit kernel)."""
    dtype = field.dtype.type
    if tot_weight is None:
        tot_weight = field.sum()
    norm = dtype(field.size / tot_weight)
    if inplace:
        field *= norm
        field -= dtype(1.0)
        return field
    return field * norm - dtype(1.0)


def get_field(pos, Lbox, nmesh, paste, w=None, d=0.0, nthread=MAX_THREADS, dtype=np.float32):
    """Overdensity mesh of the particles (:808-857).  `pos` is wrapped in place for TSC, like the reference."""
    if w is not None:
        assert pos.shape[0] == len(w)
    paste_u = paste.upper()
    if paste_u not in ('TSC', 'CIC'):
        raise ValueError(f'Unknown pasting method: {paste}')
    if paste_u == 'CIC':
        warnings.warn('Note that currently CIC pasting, unlike TSC, supports only a non-parallel implementation.')
    if np.dtype(dtype) == np.float64:     # float64 mesh: float64 accumulation and normalisation, cloud weights in the dtype of pos
        p, f64 = _pos_any(pos)
        field = np.empty((nmesh, nmesh, nmesh), dtype=np.float64)
        check(_lib.lib().abacus_field_f64(ptr(p), int(f64), C.c_int64(len(p)), ptr(_w_like(w, f64)), C.c_double(Lbox), int(nmesh),
                                          0 if paste_u == 'TSC' else 1, C.c_double(d), ptr(field)))
        return field
    if np.dtype(dtype) != np.float32:
        raise TypeError(f'mesh dtype {np.dtype(dtype)}: float32 or float64')
    p4 = _pos_f4(pos)
    field = np.empty((nmesh, nmesh, nmesh), dtype=np.float32)
    check(_lib.lib().abacus_field(ptr(p4), C.c_int64(len(p4)), ptr(_f4(w)), C.c_double(Lbox), int(nmesh),
                                  0 if paste_u == 'TSC' else 1, C.c_double(d), ptr(field)))
    return field


def _f4(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _pos_f4(pos):
    """the device path computes in float32 (the reference's default mesh dtype); float32 C-contiguous inputs are used
    in place (and wrapped in place), anything else is converted to a float32 copy"""
    if pos.dtype == np.float32 and pos.flags.c_contiguous and pos.flags.writeable:
        return pos
    return np.ascontiguousarray(pos, dtype=np.float32).copy()


def _pos_any(pos):
    """(array, is_float64): float64 positions keep their dtype (cloud weights in float64, analysis/tsc.py:400), anything else
    becomes float32; C-contiguous writeable inputs are used - and wrapped - in place"""
    if pos.dtype == np.float64:
        return _pos_f8(pos), True
    return _pos_f4(pos), False


def _w_like(w, f64):
    return None if w is None else np.ascontiguousarray(w, dtype=np.float64 if f64 else np.float32)


def _pos_f8(pos):
    """float64 C-contiguous positions are used (and wrapped) in place, anything else is converted to a float64 copy"""
    if pos.dtype == np.float64 and pos.flags.c_contiguous and pos.flags.writeable:
        return pos
    return np.ascontiguousarray(pos, dtype=np.float64).copy()


def get_field_fft(pos, Lbox, nmesh, paste, w, W, compensated, interlaced, nthread=MAX_THREADS, verbose=False,
                  dtype=np.float32):
    """delta_k of the particles as a complex64 (nmesh, nmesh, nmesh//2+1) array (:1001-1070); complex128 with
    dtype=np.float64 and interlaced=False - the reference's interlaced branch never sees `dtype` (:1048-1052: float32 meshes,
    complex64 spectrum whatever was asked), and neither does this one"""
    code = _paste_code(paste, ':')
    if np.dtype(dtype) not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise TypeError(f'mesh dtype {np.dtype(dtype)}: float32 or float64')
    if compensated:
        assert W is not None
    if np.dtype(dtype) == np.float64 and not interlaced:
        p, f64 = _pos_any(pos)
        out = np.empty((nmesh, nmesh, nmesh // 2 + 1), dtype=np.complex128)
        check(_lib.lib().abacus_field_fft_f64(ptr(p), int(f64), C.c_int64(len(p)), ptr(_w_like(w, f64)), C.c_double(Lbox), int(nmesh),
                                              code, ptr(_f4(W)) if compensated else None, ptr(out)))
        return out
    p = _pos_f4(pos)
    out = np.empty((nmesh, nmesh, nmesh // 2 + 1), dtype=np.complex64)
    check(_lib.lib().abacus_field_fft(ptr(p), C.c_int64(len(p)), ptr(_f4(w)), C.c_double(Lbox), int(nmesh), code,
                                      ptr(_f4(W)) if compensated else None, int(bool(interlaced)), ptr(out)))
    return out


def _alloc_outputs(Nk, Nmu, Np):
    return (np.zeros((Nk, Nmu), dtype=np.float32), np.zeros((Nk, Nmu), dtype=np.int64),
            np.zeros((Np, Nk), dtype=np.float32), np.zeros(Nk, dtype=np.int64),
            np.zeros((Nk, Nmu), dtype=np.float32))


def _pack(power, N_mode, binned_poles, N_mode_poles, k_avg, mu_bin_edges, squeeze_mu_axis):
    if squeeze_mu_axis and len(mu_bin_edges) == 2:
        power = power[:, 0]
        N_mode = N_mode[:, 0]
        k_avg = k_avg[:, 0]
    return dict(power=power, N_mode=N_mode, binned_poles=binned_poles, N_mode_poles=N_mode_poles, k_avg=k_avg)


def calc_pk_from_deltak(field_fft, Lbox, k_bin_edges, mu_bin_edges, field2_fft=None, poles=np.empty(0, 'i8'),
                        squeeze_mu_axis=True, nthread=MAX_THREADS):
    """Power spectrum of a given Fourier field with (k, mu) binning and optional multipoles (:730-805).
    Returns dict(power, N_mode, binned_poles, N_mode_poles, k_avg); power and binned_poles include L^3."""
    f1 = np.ascontiguousarray(field_fft, dtype=np.complex64)
    f2 = None if field2_fft is None else np.ascontiguousarray(field2_fft, dtype=np.complex64)
    nmesh = f1.shape[0]
    if f1.shape != (nmesh, nmesh, nmesh // 2 + 1) or (f2 is not None and f2.shape != f1.shape):
        raise ValueError('field_fft must have the rfftn shape (N, N, N//2+1)')
    ke = np.ascontiguousarray(k_bin_edges, dtype=np.float64)
    me = np.ascontiguousarray(mu_bin_edges, dtype=np.float64)
    pl = np.ascontiguousarray(poles, dtype=np.int64)
    outs = _alloc_outputs(len(ke) - 1, len(me) - 1, len(pl))
    check(_lib.lib().abacus_pk_from_deltak(ptr(f1), ptr(f2), int(nmesh), C.c_double(Lbox), ptr(ke), len(ke) - 1,
                                           ptr(me), len(me) - 1, ptr(pl), len(pl), *[ptr(o) for o in outs]))
    return _pack(*outs, me, squeeze_mu_axis)


def calc_power(pos, Lbox, kbins=None, mubins=None, k_max=None, logk=False, paste='TSC', nmesh=128,
               compensated=True, interlaced=True, w=None, pos2=None, w2=None, poles=None, squeeze_mu_axis=True,
               nthread=MAX_THREADS, dtype=np.float32):
    """3-D power spectrum of particle positions in a periodic box: (k, mu) wedges and optional Legendre
    multipoles.  Drop-in for abacusnbody/analysis/power_spectrum.py:1131-1319 (same arguments, same Table
    columns: k_min, k_max, k_mid, k_avg, power, N_mode[, poles, N_mode_poles][, mu_min, mu_max, mu_mid])."""
    if kbins is None:
        kbins = nmesh
    if k_max is None:
        k_max = np.pi * nmesh / Lbox
    return_mubins = mubins is not None
    if mubins is None:
        mubins = 1
    if np.dtype(dtype) not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise TypeError(f'mesh dtype {np.dtype(dtype)}: float32 or float64')

    meta = dict(Lbox=Lbox, logk=logk, paste=paste, nmesh=nmesh, compensated=compensated, interlaced=interlaced,
                poles=poles, nthread=nthread, N_pos=len(pos), is_weighted=w is not None, field_dtype=dtype,
                squeeze_mu_axis=squeeze_mu_axis)
    if pos2 is not None:
        meta['N_pos2'] = len(pos2)
        meta['is_weighted2'] = w2 is not None

    code = _paste_code(paste, '')
    W, poles_arr, kbins, mubins, ke, me = _power_setup(Lbox, nmesh, paste, compensated, interlaced, poles, k_max, kbins, mubins, logk)

    if np.dtype(dtype) == np.float64 and not interlaced:
        # float64 meshes and transform (csrc/gfft.hip in double precision); the interlaced branch of the reference ignores dtype
        # (get_field_fft :1048-1052) and so falls through to the float32 path below, like there
        f64 = pos.dtype == np.float64 or (pos2 is not None and pos2.dtype == np.float64)
        cast = _pos_f8 if f64 else _pos_f4
        p1 = cast(pos)
        p2 = None if pos2 is None else cast(pos2)
        outs = _alloc_outputs(len(ke) - 1, len(me) - 1, len(poles_arr))
        check(_lib.lib().abacus_power_f64(
            ptr(p1), int(f64), C.c_int64(len(p1)), ptr(_w_like(w, f64)), ptr(p2), C.c_int64(0 if p2 is None else len(p2)),
            ptr(_w_like(w2, f64)), C.c_double(Lbox), int(nmesh), code, ptr(_f4(W)), ptr(ke), len(ke) - 1, ptr(me), len(me) - 1,
            ptr(poles_arr), len(poles_arr), *[ptr(o) for o in outs]))
        return _power_table(outs, me, kbins, mubins, poles_arr, squeeze_mu_axis, return_mubins, meta)

    # the cloud weights are evaluated in the dtype of the positions (analysis/tsc.py:400): float64 positions go through the
    # float64 deposit (both sets then; the mesh and the transform are float32 either way)
    f64 = pos.dtype == np.float64 or (pos2 is not None and pos2.dtype == np.float64)
    cast = _pos_f8 if f64 else _pos_f4
    wcast = (lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)) if f64 else _f4
    p1 = cast(pos)
    p2 = None if pos2 is None else cast(pos2)
    outs = _alloc_outputs(len(ke) - 1, len(me) - 1, len(poles_arr))
    entry = _lib.lib().abacus_power_from_particles_f64 if f64 else _lib.lib().abacus_power_from_particles
    check(entry(
        ptr(p1), C.c_int64(len(p1)), ptr(wcast(w)), ptr(p2), C.c_int64(0 if p2 is None else len(p2)), ptr(wcast(w2)),
        C.c_double(Lbox), int(nmesh), code, ptr(_f4(W)), int(bool(interlaced)), ptr(ke), len(ke) - 1, ptr(me),
        len(me) - 1, ptr(poles_arr), len(poles_arr), *[ptr(o) for o in outs]))
    return _power_table(outs, me, kbins, mubins, poles_arr, squeeze_mu_axis, return_mubins, meta)


# BASELINE.json's north_star names the entry point `calc_power_spectrum()`; the reference itself only has `calc_power`
# (SURVEY.md section 0).  Same callable under both names.
calc_power_spectrum = calc_power


def _power_setup(Lbox, nmesh, paste, compensated, interlaced, poles, k_max, kbins, mubins, logk):
    W = get_W_compensated(Lbox, nmesh, paste, interlaced) if compensated else None
    poles_arr = np.asarray(poles or [], dtype=np.int64)
    kbins, mubins = get_k_mu_edges(Lbox, k_max, kbins, mubins, logk)
    ke = np.ascontiguousarray(kbins, dtype=np.float64)
    me = np.ascontiguousarray(mubins, dtype=np.float64)
    return W, poles_arr, kbins, mubins, ke, me


def _power_table(outs, me, kbins, mubins, poles_arr, squeeze_mu_axis, return_mubins, meta):
    P = _pack(*outs, me, squeeze_mu_axis)
    k_binc = (kbins[1:] + kbins[:-1]) * 0.5
    mu_binc = (mubins[1:] + mubins[:-1]) * 0.5
    res = dict(k_min=kbins[:-1], k_max=kbins[1:], k_mid=k_binc, k_avg=P['k_avg'], power=P['power'],
               N_mode=P['N_mode'])
    if len(poles_arr) > 0:
        res.update(poles=P['binned_poles'].T, N_mode_poles=P['N_mode_poles'])
    if return_mubins:
        res.update(mu_min=np.broadcast_to(mubins[:-1], res['power'].shape),
                   mu_max=np.broadcast_to(mubins[1:], res['power'].shape),
                   mu_mid=np.broadcast_to(mu_binc, res['power'].shape))
    return Table(res, meta=meta)


def calc_power_multi(columns, Lbox, kbins=None, mubins=None, k_max=None, logk=False, paste='TSC', nmesh=128,
                     compensated=True, interlaced=True, poles=None, squeeze_mu_axis=True):
    """Every auto and cross spectrum of several tracers whose galaxy columns are already in HBM - the loop of
    `AbacusHOD.compute_power` (abacusnbody/hod/abacus_hod.py:1400-1470) without its repeated work: each tracer's field is
    deposited and transformed ONCE (LRG x ELG: 2 deposits + FFTs instead of 4) and nothing crosses PCIe but the binned
    spectra.  `columns`: {tracer: (x, y, z)} of float64 `_lib.DeviceArray`s (`MockDict.device_xyz`); unweighted.
    Returns {(tracer_a, tracer_b): Table} for a <= b in dict order, each Table what
    `calc_power(pos_a, ..., pos2=pos_b)` returns."""
    if kbins is None:
        kbins = nmesh
    if k_max is None:
        k_max = np.pi * nmesh / Lbox
    return_mubins = mubins is not None
    if mubins is None:
        mubins = 1
    names = list(columns)
    if len(names) > 8:
        raise ValueError('calc_power_multi: at most 8 tracers')
    code = _paste_code(paste, '')
    W, poles_arr, kbins, mubins, ke, me = _power_setup(Lbox, nmesh, paste, compensated, interlaced, poles, k_max, kbins, mubins, logk)
    L = _lib.lib()
    for slot, tr in enumerate(names):
        x, y, z = columns[tr]
        check(L.abacus_power_field_soa64(slot, x.ptr, y.ptr, z.ptr, C.c_int64(len(x)), C.c_double(Lbox), int(nmesh), code,
                                         int(bool(interlaced))))
    out = {}
    for ia, ta in enumerate(names):
        for ib in range(ia, len(names)):
            tb = names[ib]
            outs = _alloc_outputs(len(ke) - 1, len(me) - 1, len(poles_arr))
            check(L.abacus_power_from_fields(ia, ib, ptr(_f4(W)), ptr(ke), len(ke) - 1, ptr(me), len(me) - 1, ptr(poles_arr),
                                             len(poles_arr), *[ptr(o) for o in outs]))
            meta = dict(Lbox=Lbox, logk=logk, paste=paste, nmesh=nmesh, compensated=compensated, interlaced=interlaced,
                        poles=poles, N_pos=len(columns[ta][0]), is_weighted=False, field_dtype=np.float32,
                        squeeze_mu_axis=squeeze_mu_axis)
            if ib != ia:
                meta.update(N_pos2=len(columns[tb][0]), is_weighted2=False)
            out[(ta, tb)] = _power_table(outs, me, kbins, mubins, poles_arr, squeeze_mu_axis, return_mubins, meta)
    check(L.abacus_power_fields_release())
    return out


# ---- the pieces of the chain as callables of their own (the reference exports them; calc_power runs them fused) -------
def bin_kmu(n1d, L, kedges, muedges, weights, poles=np.empty(0, 'i8'), dtype=np.float32, fourier=True, nthread=MAX_THREADS):
    """Mean and mode count in (k, mu) bins of an rfft-layout grid (n1d, n1d, n1d//2+1) or, fourier=False, of a real-space
    grid (n1d, n1d, n1d) (:150-300) -> (weighted_counts (Nk, Nmu), counts (Nk, Nmu) int64, weighted_counts_poles (Np, Nk),
    counts_poles (Nk,) int64, weighted_counts_k (Nk, Nmu)) - not multiplied by L^3, exactly as the reference returns them"""
    w = _grid_f4(weights)
    if int(n1d) != w.shape[0]:
        raise ValueError(f'n1d = {n1d} but the grid is {w.shape}')
    ke = np.ascontiguousarray(kedges, dtype=np.float64)
    me = np.ascontiguousarray(muedges, dtype=np.float64)
    po = np.ascontiguousarray(poles, dtype=np.int64)
    power, N_mode, bp, Nmp, k_avg = _alloc_outputs(len(ke) - 1, len(me) - 1, len(po))
    check(_lib.lib().abacus_bin_weights(ptr(w), int(n1d), int(w.shape[2]), C.c_double(L), int(bool(fourier)), ptr(ke), len(ke) - 1,
                                        ptr(me), len(me) - 1, ptr(po), len(po), C.c_double(1.0), ptr(power), ptr(N_mode), ptr(bp),
                                        ptr(Nmp), ptr(k_avg)))
    dt = np.dtype(dtype)
    return power.astype(dt, copy=False), N_mode, bp.astype(dt, copy=False), Nmp, k_avg.astype(dt, copy=False)


def _c64(a, name):
    a = np.asarray(a)
    if a.dtype != np.complex64:
        raise NotImplementedError(f'{name}: complex64 spectra only (float32 meshes), got {a.dtype}')
    return np.ascontiguousarray(a)


def get_raw_power(field_fft, field2_fft=None):
    """|field_fft|^2, or Re(conj(field_fft) field2_fft) (:707-727) -> float32 array of the same shape"""
    f1 = _c64(field_fft, 'get_raw_power')
    f2 = None if field2_fft is None else _c64(field2_fft, 'get_raw_power')
    if f2 is not None and f2.shape != f1.shape:
        raise ValueError('field_fft and field2_fft differ in shape')
    out = np.empty(f1.shape, dtype=np.float32)
    check(_lib.lib().abacus_raw_power(ptr(f1), ptr(f2), C.c_int64(f1.size), ptr(out)))
    return out


def shift_field_fft(field_fft, field_shift_fft, n1d, L, d, dtype=np.float32):
    """field_fft += field_shift_fft * exp(i (d / 2)(kx + ky + kz)); field_fft *= 0.5 / n1d^3, IN PLACE like the reference
    (:904-948: it returns nothing)"""
    if np.dtype(dtype) != np.float32 or not isinstance(field_fft, np.ndarray) or field_fft.dtype != np.complex64 or \
            not field_fft.flags.c_contiguous:
        raise NotImplementedError('shift_field_fft: C-contiguous complex64 spectra (float32 meshes) only')
    n1d = int(n1d)
    if field_fft.shape != (n1d, n1d, n1d // 2 + 1) or np.shape(field_shift_fft) != field_fft.shape:
        raise ValueError(f'expected two ({n1d}, {n1d}, {n1d // 2 + 1}) spectra')
    check(_lib.lib().abacus_shift_field_fft(ptr(field_fft), ptr(_c64(field_shift_fft, 'shift_field_fft')), n1d, C.c_double(L),
                                            C.c_double(d)))


def get_interlaced_field_fft(pos, Lbox, nmesh, paste, w, nthread=MAX_THREADS, verbose=False):
    """the two deposits half a cell apart, their transforms and shift_field_fft (:951-998), fused on the device -> complex64
    (nmesh, nmesh, nmesh//2+1), normalised by 0.5 / nmesh^3 like the reference's"""
    return get_field_fft(pos, Lbox, nmesh, paste, w, None, False, True, nthread=nthread, verbose=verbose)


# ---- ZCV-facing helpers (analysis/power_spectrum.py:303-660 of the reference) ------------------------------------
def _grid_f4(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 3 or a.shape[0] != a.shape[1]:
        raise ValueError('expected an (n, n, n//2+1) or (n, n, n) grid')
    return a


def bin_kppi(n1d, L, kedges, pimax, Npi, weights, dtype=np.float32, fourier=True, nthread=MAX_THREADS):
    """Mean and mode count in (k_perp, pi) bins of an rfft-layout grid (or, fourier=False, a real-space grid)
    (:303-412) -> (weighted_counts (Nk, Npi) float32, counts (Nk, Npi) int64)"""
    w = _grid_f4(weights)
    ke = np.ascontiguousarray(kedges, dtype=np.float64)
    Nk = len(ke) - 1
    mean = np.zeros((Nk, Npi), dtype=np.float32)
    counts = np.zeros((Nk, Npi), dtype=np.int64)
    check(_lib.lib().abacus_bin_kppi(ptr(w), int(n1d), int(w.shape[2]), C.c_double(L), ptr(ke), Nk, C.c_double(pimax),
                                     int(Npi), int(bool(fourier)), ptr(mean), ptr(counts)))
    return mean.astype(dtype, copy=False), counts


def project_3d_to_poles(k_bin_edges, raw_p3d, Lbox, poles):
    """Multipoles of a 3-D power spectrum given on the rfftn grid (:415-448) -> (binned_poles (Np, Nk), Npoles (Nk,))"""
    assert np.max(poles) <= 10, 'numba implementation works up to ell = 10'
    w = _grid_f4(raw_p3d)
    n = w.shape[0]
    ke = np.ascontiguousarray(k_bin_edges, dtype=np.float64)
    me = np.array([0.0, 1.0])
    po = np.ascontiguousarray(poles, dtype=np.int64)
    power, N_mode, bp, Nmp, k_avg = _alloc_outputs(len(ke) - 1, 1, len(po))
    check(_lib.lib().abacus_bin_weights(ptr(w), n, int(w.shape[2]), C.c_double(Lbox), 1, ptr(ke), len(ke) - 1, ptr(me), 1,
                                        ptr(po), len(po), C.c_double(float(Lbox) ** 3), ptr(power), ptr(N_mode), ptr(bp),
                                        ptr(Nmp), ptr(k_avg)))
    return bp, Nmp


def pk_to_xi(Pk, Lbox, r_bins, poles=[0, 2, 4]):
    """Correlation-function multipoles of a 3-D power spectrum (:620-660): irfftn on the device, then the multipoles of
    Xi in r bins -> (r_binc, binned_poles (Np, Nr), Npoles (Nr,))"""
    w = _grid_f4(Pk)
    n = w.shape[0]
    if w.shape[2] != n // 2 + 1:
        raise ValueError('Pk must have the rfftn shape (n, n, n//2+1)')
    rb = np.ascontiguousarray(r_bins, dtype=np.float64)
    po = np.ascontiguousarray(poles, dtype=np.int64)
    bp = np.zeros((len(po), len(rb) - 1), dtype=np.float32)
    Nmp = np.zeros(len(rb) - 1, dtype=np.int64)
    check(_lib.lib().abacus_pk_to_xi(ptr(w), n, C.c_double(Lbox), ptr(rb), len(rb) - 1, ptr(po), len(po), ptr(bp), ptr(Nmp)))
    return (rb[1:] + rb[:-1]) * 0.5, bp, Nmp


def expand_poles_to_3d(k_ell, P_ell, n1d, L, poles, dtype=np.float32):
    """3-D power spectrum on the fundamental modes from its multipoles (:451-505) -> (n1d, n1d, n1d//2+1)"""
    k_ell = np.ascontiguousarray(k_ell, dtype=np.float64)
    P_ell = np.ascontiguousarray(P_ell, dtype=np.float64).reshape(len(poles), len(k_ell))
    assert np.abs((k_ell[1] - k_ell[0]) - (k_ell[-1] - k_ell[-2])) < 1.0e-6
    po = np.ascontiguousarray(poles, dtype=np.int64)
    out = np.empty((n1d, n1d, n1d // 2 + 1), dtype=np.float32)
    check(_lib.lib().abacus_expand_poles_to_3d(ptr(k_ell), ptr(P_ell), len(k_ell), int(n1d), C.c_double(L), ptr(po), len(po),
                                               ptr(out)))
    return out.astype(dtype, copy=False)


def get_smoothing(n1d, L, R, dtype=np.float32):
    """Gaussian kernel exp(-k^2 R^2 / 2) on the rfftn grid (:527-577)"""
    out = np.empty((n1d, n1d, n1d // 2 + 1), dtype=np.float32)
    check(_lib.lib().abacus_get_smoothing(int(n1d), C.c_double(L), C.c_double(R), ptr(out)))
    return out.astype(dtype, copy=False)


def get_delta_mu2(delta, n1d, dtype_c=np.complex64, dtype_f=np.float32):
    """delta * mu^2 of a Fourier field on the rfftn grid (:580-617)"""
    d = np.ascontiguousarray(delta, dtype=np.complex64)
    out = np.empty_like(d)
    check(_lib.lib().abacus_get_delta_mu2(ptr(d), int(n1d), ptr(out)))
    return out.astype(dtype_c, copy=False)
