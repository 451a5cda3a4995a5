"""ctypes front-end of the CPU oracle (oracle/abacus_oracle.c) - TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product path (abacusutils_amd) never does.  Function names and
signatures follow the reference (abacusnbody/...; file:line cited per function);
the heavy loops are in C, the 3-D FFT is scipy.fft.rfftn (pocketfft), which is
the reference's own FFT (analysis/power_spectrum.py:12,980,986,1059).
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', str(_HERE)])


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / 'liboracle.so'
        src = _HERE / 'abacus_oracle.c'
        if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
            build()
        _LIB = C.CDLL(str(so))
    return _LIB


def cpu_quota():
    """CPUs' worth of time the cgroup grants this process (None: no limit)"""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return q / per if q > 0 else None
    except (OSError, ValueError):
        return None


def max_threads():
    """team size for the OpenMP / pocketfft legs: the OpenMP maximum, capped at twice the cgroup's CPU quota - a team far wider
    than the quota is throttled, not faster (GPU box of round 6: 256 logical CPUs in the affinity mask, cpu.max = 16 CPUs;
    calc_power at 512^3: 0.96 s with 32 threads, 2.0 s with 256; gen_gals of 4e6 halos: 4.6 ms with 128, 608 ms with 256)"""
    n = int(lib().oracle_max_threads())
    q = cpu_quota()
    return n if not q else max(1, min(n, int(round(2 * q))))


_D = C.c_double
_LRG_KEYS = ['logM_cut', 'logM1', 'sigma', 'alpha', 'kappa', 'alpha_c', 'alpha_s', 's', 's_v', 's_p', 's_r',
             'Acent', 'Asat', 'Bcent', 'Bsat', 'ic']
_ELG_KEYS = ['p_max', 'Q', 'logM_cut', 'kappa', 'sigma', 'logM1', 'alpha', 'gamma', 'A_s', 'alpha_c', 'alpha_s',
             's', 's_v', 's_p', 's_r', 'Acent', 'Asat', 'Bcent', 'Bsat', 'Ccent', 'Csat', 'ic',
             'logM1_EE', 'alpha_EE', 'logM1_EL', 'alpha_EL']
_QSO_KEYS = ['logM_cut', 'kappa', 'sigma', 'logM1', 'alpha', 'alpha_c', 'alpha_s', 's', 's_v', 's_p', 's_r',
             'Acent', 'Asat', 'Bcent', 'Bsat', 'ic']


class HodParams(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ('want_LRG', 'want_ELG', 'want_QSO', 'rsd', 'has_origin', 'enable_ranks', 'pad0', 'pad1')]
        + [('inv_velz2kms', _D), ('lbox', _D), ('origin', _D * 3)]
        + [('L_' + k, _D) for k in _LRG_KEYS]
        + [('E_' + k, _D) for k in _ELG_KEYS]
        + [('Q_' + k, _D) for k in _QSO_KEYS]
    )


def marshal_params(tracers, params, enable_ranks, rsd):
    """gen_gals parameter handling (hod/GRAND_HOD.py:1342-1475): z-evolution of
    logM_cut/logM1, defaults for the optional keys, inv_velz2kms."""
    p = HodParams()
    p.rsd = int(bool(rsd))
    p.enable_ranks = int(bool(enable_ranks))
    p.inv_velz2kms = 1 / params['velz2kms']
    p.lbox = params['Lbox']
    origin = params.get('origin', None)
    p.has_origin = int(origin is not None)
    if origin is not None:
        for i in range(3):
            p.origin[i] = float(origin[i])
    for tr, pre, keys in (('LRG', 'L_', _LRG_KEYS), ('ELG', 'E_', _ELG_KEYS), ('QSO', 'Q_', _QSO_KEYS)):
        if tr not in tracers:
            continue
        setattr(p, 'want_' + tr, 1)
        hod = dict(tracers[tr])
        delta_a = 1.0 / (1 + params['z']) - 1.0 / (1 + hod.get('z_pivot', params['z']))
        hod['logM_cut'] = hod['logM_cut'] + hod.get('logM_cut_pr', 0.0) * delta_a
        hod['logM1'] = hod['logM1'] + hod.get('logM1_pr', 0.0) * delta_a
        for k in ('Acent', 'Asat', 'Bcent', 'Bsat', 'Ccent', 'Csat'):
            hod.setdefault(k, 0.0)
        hod.setdefault('ic', 1.0)
        if tr == 'ELG':
            hod.setdefault('logM1_EE', hod['logM1'])
            hod.setdefault('alpha_EE', hod['alpha'])
            hod.setdefault('logM1_EL', hod['logM1'])
            hod.setdefault('alpha_EL', hod['alpha'])
        for k in keys:
            setattr(p, pre + k, float(hod[k]))  # KeyError for a missing required key, as the typed dict lookup
    return p


def _f8(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, t=C.c_void_p):
    return None if a is None else a.ctypes.data_as(t)


TRACERS = ('LRG', 'ELG', 'QSO')
COLS = ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass')


def _two_pass(fn, n, args_before, p, nthread):
    keep = np.empty(n, dtype=np.int8)
    counts = np.zeros(3, dtype=np.int64)
    fn(*args_before, C.byref(p), nthread, _ptr(keep), _ptr(counts), None, None)
    outs = [[np.empty(counts[t], dtype=np.float64) for _ in COLS] for t in range(3)]
    ids = [np.empty(counts[t], dtype=np.int64) for t in range(3)]
    outp = (C.c_void_p * 21)(*[a.ctypes.data for t in range(3) for a in outs[t]])
    idp = (C.c_void_p * 3)(*[a.ctypes.data for a in ids])
    fn(*args_before, C.byref(p), nthread, _ptr(keep), _ptr(counts), outp, idp)
    return keep, outs, ids


def gen_cent(halo_data, p, nthread):
    """hod/GRAND_HOD.py:139-414"""
    h = halo_data
    n = len(h['hmass'])
    arrs = [_f8(h['hpos']), _f8(h['hvel']), _f8(h['hmass']), np.ascontiguousarray(h['hid'], dtype=np.int64),
            _f8(h['hmultis']), _f8(h['hrandoms']), _f8(h['hveldev'])]
    opt = [(_f8(h[k]) if k in h else None) for k in ('hdeltac', 'hfenv', 'hshear')]
    args = [C.c_int64(n)] + [_ptr(a) for a in arrs] + [_ptr(a) for a in opt]
    return _two_pass(lib().oracle_gen_cent, n, args, p, nthread)


def gen_sats(particle_data, p, nthread, keep_cent_at_pinds):
    """hod/GRAND_HOD.py:825-1262"""
    s = particle_data
    n = len(s['phmass'])
    arrs = [_f8(s['ppos']), _f8(s['pvel']), _f8(s['phvel']), _f8(s['phmass']),
            np.ascontiguousarray(s['phid'], dtype=np.int64), _f8(s['pweights']), _f8(s['prandoms'])]
    opt = [(_f8(s[k]) if k in s else None) for k in ('pdeltac', 'pfenv', 'pshear')]
    ranks = [_f8(s[k]) for k in ('pranks', 'pranksv', 'pranksp', 'pranksr')]
    kc = np.ascontiguousarray(keep_cent_at_pinds, dtype=np.int8)
    args = [C.c_int64(n)] + [_ptr(a) for a in arrs + opt + ranks] + [_ptr(kc)]
    return _two_pass(lib().oracle_gen_sats, n, args, p, nthread)


def time_gen_gals(halo_data, particle_data, tracers, params, nthread, reps=3, enable_ranks=False, rsd=True):
    """CPU-baseline timing of the compiled kernels alone (bench.py `cpu_baseline`): arrays marshalled and output buffers
    allocated outside the timed region; one repetition = gen_cent (pass 1 + pass 2), the keep_cent[pinds] gather
    (hod/GRAND_HOD.py:1562), gen_sats (pass 1 + pass 2).  Returns (seconds per repetition [min], [mean], n_gal)."""
    import time
    p = marshal_params(tracers, params, enable_ranks, rsd)
    h, s = halo_data, particle_data
    nh, npart = len(h['hmass']), len(s['phmass'])
    harr = [_f8(h['hpos']), _f8(h['hvel']), _f8(h['hmass']), np.ascontiguousarray(h['hid'], dtype=np.int64),
            _f8(h['hmultis']), _f8(h['hrandoms']), _f8(h['hveldev'])]
    hopt = [(_f8(h[k]) if k in h else None) for k in ('hdeltac', 'hfenv', 'hshear')]
    parr = [_f8(s['ppos']), _f8(s['pvel']), _f8(s['phvel']), _f8(s['phmass']),
            np.ascontiguousarray(s['phid'], dtype=np.int64), _f8(s['pweights']), _f8(s['prandoms'])]
    popt = [(_f8(s[k]) if k in s else None) for k in ('pdeltac', 'pfenv', 'pshear')]
    ranks = [_f8(s[k]) for k in ('pranks', 'pranksv', 'pranksp', 'pranksr')]
    pinds = np.ascontiguousarray(s['pinds'], dtype=np.int64)
    keep_c, keep_s, kc = np.empty(nh, np.int8), np.empty(npart, np.int8), np.empty(npart, np.int8)
    cnt_c, cnt_s = np.zeros(3, np.int64), np.zeros(3, np.int64)
    L = lib()
    hargs = [C.c_int64(nh)] + [_ptr(a) for a in harr + hopt]
    L.oracle_gen_cent(*hargs, C.byref(p), nthread, _ptr(keep_c), _ptr(cnt_c), None, None)     # sizing pass
    L.oracle_gather_i8(_ptr(keep_c), _ptr(pinds), C.c_int64(npart), _ptr(kc), nthread)
    pargs = [C.c_int64(npart)] + [_ptr(a) for a in parr + popt + ranks] + [_ptr(kc)]
    L.oracle_gen_sats(*pargs, C.byref(p), nthread, _ptr(keep_s), _ptr(cnt_s), None, None)

    def bufs(cnt):
        outs = [[np.empty(cnt[t], np.float64) for _ in COLS] for t in range(3)]
        ids = [np.empty(cnt[t], np.int64) for t in range(3)]
        return outs, ids, (C.c_void_p * 21)(*[a.ctypes.data for t in range(3) for a in outs[t]]), \
            (C.c_void_p * 3)(*[a.ctypes.data for a in ids])
    oc, ic, ocp, icp = bufs(cnt_c)
    os_, is_, osp, isp = bufs(cnt_s)
    ts = []
    for _ in range(reps + 1):          # first repetition = warm-up
        t0 = time.perf_counter()
        L.oracle_gen_cent(*hargs, C.byref(p), nthread, _ptr(keep_c), _ptr(cnt_c), ocp, icp)
        L.oracle_gather_i8(_ptr(keep_c), _ptr(pinds), C.c_int64(npart), _ptr(kc), nthread)
        L.oracle_gen_sats(*pargs, C.byref(p), nthread, _ptr(keep_s), _ptr(cnt_s), osp, isp)
        ts.append(time.perf_counter() - t0)
    ts = ts[1:]
    return min(ts), sum(ts) / len(ts), int(cnt_c.sum() + cnt_s.sum())


def gen_gal_cat(halo_data, particle_data, tracers, params, Nthread=16, enable_ranks=False, rsd=True,
                return_keep=False):
    """gen_gal_cat -> gen_gals (hod/GRAND_HOD.py:1595-1724,1302-1592), particle-based path"""
    if not isinstance(rsd, bool):
        raise ValueError('Error: rsd has to be a boolean')
    p = marshal_params(tracers, params, enable_ranks, rsd)
    keep_c, out_c, id_c = gen_cent(halo_data, p, Nthread)
    keep_s, out_s, id_s = gen_sats(particle_data, p, Nthread, keep_c[particle_data['pinds']])
    mock = {}
    for tr in tracers:
        t = TRACERS.index(tr)
        d = {'Ncent': len(out_c[t][0])}
        for c, name in enumerate(COLS):
            d[name] = np.concatenate((out_c[t][c], out_s[t][c]))
        d['id'] = np.concatenate((id_c[t], id_s[t]))
        mock[tr] = d
    if return_keep:
        return mock, keep_c, keep_s
    return mock


# ---------------------------------------------------------------------------
# TSC / CIC / partition
# ---------------------------------------------------------------------------
def _suffix(a):
    return {np.dtype('f4'): 'f32', np.dtype('f8'): 'f64'}[a.dtype]


def wrap_inplace(pos, box):
    """analysis/tsc.py:219-226"""
    getattr(lib(), 'oracle_wrap_inplace_' + _suffix(pos))(_ptr(pos), C.c_int64(len(pos)), _D(box))


def tsc_scatter(positions, density, boxsize, weights=None, offset=0.0):
    """analysis/tsc.py:394-507 (3-D; serial, input order)"""
    assert positions.flags.c_contiguous and density.flags.c_contiguous and density.ndim == 3
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=positions.dtype)
    fn = getattr(lib(), f'oracle_tsc_scatter_{_suffix(positions)}_{_suffix(density)}')
    gx, gy, gz = density.shape
    fn(_ptr(positions), C.c_int64(len(positions)), _ptr(density), gx, gy, gz, _D(boxsize), _ptr(weights), _D(offset))


def cic_serial(positions, density, boxsize, weights=None):
    """analysis/cic.py:13-125"""
    assert positions.flags.c_contiguous and density.dtype == np.float32
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=positions.dtype)
    gx, gy, gz = density.shape
    getattr(lib(), 'oracle_cic_' + _suffix(positions))(
        _ptr(positions), C.c_int64(len(positions)), _ptr(density), gx, gy, gz, _D(boxsize), _ptr(weights))


def partition_parallel(pos, npartition, boxsize, weights=None, coord=0, nthread=-1, sort=False):
    """analysis/tsc.py:259-384 (sort=False only)"""
    assert not sort
    if nthread < 0:
        nthread = max_threads()
    pos = np.ascontiguousarray(pos)
    psort = np.empty_like(pos)
    starts = np.empty(npartition + 1, dtype=np.int64)
    wsort = None
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=pos.dtype)
        wsort = np.empty_like(weights)
    getattr(lib(), 'oracle_partition_' + _suffix(pos))(
        _ptr(pos), C.c_int64(len(pos)), int(npartition), _D(boxsize), _ptr(weights), int(coord), int(nthread),
        _ptr(psort), _ptr(starts), _ptr(wsort))
    return psort, starts, wsort


def default_npartition(n1d, nthread):
    """analysis/tsc.py:126-139"""
    if nthread > 1:
        if 2 * nthread >= n1d // 2:
            npartition = n1d // 2
            npartition = 2 * (npartition // 2)
            if npartition < n1d // 2:
                npartition = n1d // 3
        else:
            npartition = min(n1d // 3, 2 * nthread)
        npartition = 2 * (npartition // 2)
    else:
        npartition = 1
    return max(npartition, 1)


def tsc_parallel(pos, densgrid, box, weights=None, nthread=-1, wrap=True, npartition=None, offset=0.0):
    """analysis/tsc.py:10-206: wrap in place -> stripe partition -> even/odd stripe scatter"""
    if nthread < 0:
        nthread = max_threads()
    if isinstance(densgrid, (int, np.integer)):
        densgrid = (densgrid,) * 3
    user = not isinstance(densgrid, tuple)
    if not user:
        densgrid = np.zeros(densgrid, dtype=np.float32)
    n1d = densgrid.shape[0]
    if not npartition:
        npartition = default_npartition(n1d, nthread)
    if npartition > n1d // 3 and npartition != n1d // 2 and nthread > 1:
        raise ValueError(f'npartition {npartition} must be less than ngrid//3 = {n1d // 3} or equal to ngrid//2 = {n1d // 2}')
    if npartition > 1 and npartition % 2 != 0 and nthread > 1:
        raise ValueError(f'npartition {npartition} not divisible by 2')
    if wrap:
        wrap_inplace(pos, box)
    if npartition > 1:
        ppart, starts, wpart = partition_parallel(pos, npartition, box, weights=weights, nthread=nthread)
    else:
        ppart, wpart = pos, (None if weights is None else np.ascontiguousarray(weights, dtype=pos.dtype))
        starts = np.array([0, len(pos)], dtype=np.int64)
    assert densgrid.dtype == np.float32, 'the stripe scatter of the oracle is float32-grid only (tsc_scatter takes float64 grids)'
    fn = getattr(lib(), f'oracle_tsc_parallel_{_suffix(ppart)}_f32')
    gx, gy, gz = densgrid.shape
    # The reference admits npartition == n1d//2 (tsc.py:128-134,141): stripes two cells wide, while a TSC cloud
    # reaches 1.5 cells past its stripe on each side, so same-parity stripes can touch the same x-plane (a data
    # race in the reference; it needs two threads in the same cell at the same instant).  The oracle must be
    # deterministic: stripes narrower than three cells are scattered serially.
    scatter_threads = nthread if npartition <= max(n1d // 3, 1) else 1
    fn(_ptr(ppart), _ptr(starts), int(len(starts) - 1), _ptr(densgrid), gx, gy, gz, _D(box), _ptr(wpart),
       _D(offset), int(scatter_threads))
    return None if user else densgrid


# ---------------------------------------------------------------------------
# power spectrum
# ---------------------------------------------------------------------------
def get_k_mu_edges(Lbox, k_max, kbins, mubins, logk):
    """analysis/power_spectrum.py:663-704"""
    if isinstance(kbins, int):
        if logk:
            kbins = np.geomspace((1.0 - 1.0e-4) * 2.0 * np.pi / Lbox, k_max, kbins + 1)
        else:
            kbins = np.linspace(0.0, k_max, kbins + 1)
    if isinstance(mubins, int):
        mubins = np.linspace(0.0, 1.0, mubins + 1)
    return kbins, mubins


def get_W_compensated(Lbox, nmesh, paste, interlaced):
    """analysis/power_spectrum.py:1081-1128"""
    from scipy.fft import fftfreq
    d = Lbox / nmesh
    kN = np.pi / d
    k = (fftfreq(nmesh, d=d) * 2.0 * np.pi).astype(np.float32)
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


def get_field(pos, Lbox, nmesh, paste, w=None, d=0.0, nthread=1):
    """analysis/power_spectrum.py:808-857 (float32 mesh)"""
    field = np.zeros((nmesh, nmesh, nmesh), dtype=np.float32)
    paste = paste.upper()
    if paste == 'TSC':
        tsc_parallel(pos, field, Lbox, weights=w, nthread=nthread, offset=d)
    elif paste == 'CIC':
        cic_serial(np.ascontiguousarray(pos + d) if d != 0.0 else pos, field, Lbox, weights=w)
    else:
        raise ValueError(f'Unknown pasting method: {paste}')
    lib().oracle_normalize_field_f32(_ptr(field), C.c_int64(field.size), _D(len(pos)))
    return field


def get_field_fft(pos, Lbox, nmesh, paste, w, W, compensated, interlaced, nthread=1):
    """analysis/power_spectrum.py:1001-1070, 951-998"""
    from scipy.fft import rfftn
    L = lib()
    if interlaced:
        d = Lbox / nmesh
        f = rfftn(get_field(pos, Lbox, nmesh, paste, w, nthread=nthread), workers=nthread)
        fs = rfftn(get_field(pos, Lbox, nmesh, paste, w, d=0.5 * d, nthread=nthread), workers=nthread)
        assert f.dtype == np.complex64
        L.oracle_shift_field_fft(_ptr(f), _ptr(fs), int(nmesh), _D(Lbox), _D(d))
    else:
        field = get_field(pos, Lbox, nmesh, paste, w, nthread=nthread)
        inv_size = np.float32(1 / field.size)
        f = rfftn(field, overwrite_x=True, workers=nthread)
        L.oracle_scale_c64(_ptr(f), C.c_int64(f.size), C.c_float(inv_size))
    if compensated:
        L.oracle_compensate(_ptr(f), int(nmesh), _ptr(np.ascontiguousarray(W, dtype=np.float32)))
    return f


def get_raw_power(field_fft, field2_fft=None):
    """analysis/power_spectrum.py:707-727"""
    if field2_fft is not None:
        return (np.conj(field_fft) * field2_fft).real
    return np.abs(field_fft) ** 2


def shift_field_fft(field_fft, field_shift_fft, n1d, L, d):
    """analysis/power_spectrum.py:904-948, in place"""
    assert field_fft.dtype == np.complex64 and field_fft.flags.c_contiguous
    lib().oracle_shift_field_fft(_ptr(field_fft), _ptr(np.ascontiguousarray(field_shift_fft, dtype=np.complex64)), int(n1d), _D(L), _D(d))


def get_interlaced_field_fft(pos, Lbox, nmesh, paste, w, nthread=1):
    """analysis/power_spectrum.py:951-998"""
    return get_field_fft(pos, Lbox, nmesh, paste, w, None, False, True, nthread=nthread)


def bin_kmu(n1d, L, kedges, muedges, weights, poles=np.empty(0, 'i8'), fourier=True, accum64=False, nthread=1):
    """analysis/power_spectrum.py:150-300"""
    kedges = _f8(kedges)
    muedges = _f8(muedges)
    poles = np.ascontiguousarray(poles, dtype=np.int64)
    Nk, Nmu, Np = len(kedges) - 1, len(muedges) - 1, len(poles)
    # a configuration-space grid is (n1d, n1d, n1d): the loops visit k < n1d // 2 + 1 of every row (:232-236)
    weights = np.ascontiguousarray(np.asarray(weights, dtype=np.float32)[:, :, :n1d // 2 + 1])
    power = np.zeros((Nk, Nmu), dtype=np.float32)
    counts = np.zeros((Nk, Nmu), dtype=np.int64)
    bpoles = np.zeros((Np, Nk), dtype=np.float32)
    cpoles = np.zeros(Nk, dtype=np.int64)
    kavg = np.zeros((Nk, Nmu), dtype=np.float32)
    lib().oracle_bin_kmu(int(n1d), _D(L), _ptr(kedges), Nk, _ptr(muedges), Nmu, _ptr(weights), _ptr(poles), Np,
                         int(fourier), int(accum64), int(nthread), _ptr(power), _ptr(counts), _ptr(bpoles),
                         _ptr(cpoles), _ptr(kavg))
    return power, counts, bpoles, cpoles, kavg


def particle_cloud_tables(pos, w, Lbox, nmesh, paste='TSC', offset=0.0):
    """[P][3][nmesh] complex128: per particle and dimension, the discrete Fourier transform of its 1-D mass-assignment
    cloud on the mesh, sum_s w_s exp(-2 pi i m c_s / n) for every integer frequency index m.  The TSC weights are
    evaluated as _tsc_scatter does (analysis/tsc.py:400-451): in the DTYPE OF THE POSITIONS (`ftype`), with
    `inv_h = ftype(g / box)`, `p = (x + ftype(offset)) * inv_h`, `d = ftype(round(p)) - p` - at nmesh 2048 a float32 grid
    coordinate resolves 1.2e-4 of a cell, which moves single-particle window amplitudes at the 1e-4 level, so a float64
    evaluation is NOT what the reference deposits.  CIC (analysis/cic.py:29-71) is float64 whatever the dtype.  The
    particle weight multiplies the x table.  The 3-D transform of the deposited mesh is sum_p X_p[i] Y_p[j] Z_p[k]:
    exact, aliasing included."""
    pos = np.asarray(pos)
    ft = pos.dtype.type if pos.dtype in (np.float32, np.float64) else np.float64
    P = len(pos)
    w = np.ones(P) if w is None else np.asarray(w, dtype=np.float64)
    m = np.arange(nmesh)
    tabs = np.zeros((P, 3, nmesh), dtype=np.complex128)
    inv_h, off, HALF, P75 = ft(nmesh / Lbox), ft(offset), ft(0.5), ft(0.75)
    for p in range(P):
        for d in range(3):
            if paste.upper() == 'TSC':
                x = (ft(pos[p, d]) + off) * inv_h
                c = np.rint(x)                      # round half to even, like the reference's round()
                dd = ft(c) - x
                tm, tp, c = HALF + dd, HALF - dd, float(c)
                cells = [(c - 1, float(HALF * (tm * tm))), (c, float(P75 - dd * dd)), (c + 1, float(HALF * (tp * tp)))]
            else:
                x = (float(pos[p, d]) + offset) / Lbox * nmesh
                c = np.floor(x)
                f = x - c
                cells = [(c, 1 - f), (c + 1, f)]
            t = sum(ws * np.exp(-2j * np.pi * m * (cs % nmesh) / nmesh) for cs, ws in cells)
            tabs[p, d] = t * (w[p] if d == 0 else 1.0)
    return tabs


def pk_of_particles_analytic(pos, w, Lbox, nmesh, kedges, muedges, poles, paste='TSC', compensated=False, interlaced=False,
                             nthread=1):
    """calc_power (analysis/power_spectrum.py:1131-1319) of a HANDFUL of particles in closed form: delta_k = (1/N) sum_p
    [cloud transform], the interlaced combination (:904-948) as a second set of separable terms, the compensation window
    (:1081-1128) as a separable divisor, then bin_kmu's rule with float64 sums - nothing of mesh size is allocated, so the
    known answer at 2048^3 costs seconds.  Returns the columns of the Table as a dict."""
    pos = np.asarray(pos)                         # dtype kept: the cloud weights are evaluated in it, as the reference does
    P = len(pos)
    tabs = particle_cloud_tables(pos, w, Lbox, nmesh, paste)
    amp = 1.0 / P                                 # rho * (M / len(pos)) / M; the -1 only touches k = 0
    if interlaced:
        d = Lbox / nmesh
        sh = particle_cloud_tables(pos, w, Lbox, nmesh, paste, offset=0.5 * d)
        m = np.arange(nmesh)
        kk = np.where(m < nmesh // 2, m, m - nmesh)           # Nyquist takes the negative branch (:940-942)
        phase = np.exp(1j * np.pi * kk / nmesh)               # exp(i (d/2) k), k = kk * 2 pi / L
        sh = sh * phase[None, None, :]
        tabs = np.concatenate([tabs, sh])
        amp *= 0.5
    comp = None
    if compensated:
        W = get_W_compensated(Lbox, nmesh, paste, interlaced).astype(np.float64)
        comp = np.ascontiguousarray(np.stack([W, W, W]))
    kedges, muedges = _f8(kedges), _f8(muedges)
    poles_arr = np.ascontiguousarray(poles if poles is not None else [], dtype=np.int64)
    Nk, Nmu, Np = len(kedges) - 1, len(muedges) - 1, len(poles_arr)
    power = np.zeros((Nk, Nmu), dtype=np.float32)
    counts = np.zeros((Nk, Nmu), dtype=np.int64)
    bpoles = np.zeros((Np, Nk), dtype=np.float32)
    cpoles = np.zeros(Nk, dtype=np.int64)
    kavg = np.zeros((Nk, Nmu), dtype=np.float32)
    tri = np.ascontiguousarray(np.stack([tabs.real, tabs.imag], axis=-1))      # [T][3][n][2]
    lib().oracle_bin_kmu_separable(int(nmesh), _D(Lbox), _ptr(kedges), Nk, _ptr(muedges), Nmu, int(len(tabs)), _ptr(tri),
                                   _ptr(comp), _D(amp), _ptr(poles_arr), Np, int(nthread), _ptr(power), _ptr(counts),
                                   _ptr(bpoles), _ptr(cpoles), _ptr(kavg))
    # the k = 0 mode carries sum(w)/N - 1, not sum(w)/N (the window is 1 there): correct the bin that holds it (it is
    # binned iff kedges[0] <= 0); mu = 0 for the zero vector (:243)
    if kedges[0] <= 0 and counts[0, 0] > 0:
        d0 = (float(np.sum(w)) if w is not None else float(P)) / P
        fix = (d0 - 1.0) ** 2 - d0 ** 2
        power[0, 0] += np.float32(fix / counts[0, 0])
        for ip, ell in enumerate(poles_arr):
            pw = 1.0 if ell == 0 else (2 * ell + 1) * float(np.polynomial.legendre.legval(0.0, [0] * int(ell) + [1]))
            bpoles[ip, 0] += np.float32(fix * pw / cpoles[0])
    L3 = Lbox ** 3
    return dict(power=power * np.float32(L3), N_mode=counts, poles=(bpoles * np.float32(L3)).T, N_mode_poles=cpoles, k_avg=kavg)


def calc_pk_from_deltak(field_fft, Lbox, k_bin_edges, mu_bin_edges, field2_fft=None, poles=np.empty(0, 'i8'),
                        squeeze_mu_axis=True, nthread=1, accum64=False):
    """analysis/power_spectrum.py:730-805"""
    field_fft = np.ascontiguousarray(field_fft, dtype=np.complex64)
    raw = np.empty(field_fft.shape, dtype=np.float32)
    f2 = None if field2_fft is None else np.ascontiguousarray(field2_fft, dtype=np.complex64)
    lib().oracle_raw_power(_ptr(field_fft), _ptr(f2), C.c_int64(field_fft.size), _ptr(raw))
    nmesh = raw.shape[0]
    power, N_mode, binned_poles, N_mode_poles, k_avg = bin_kmu(
        nmesh, Lbox, k_bin_edges, mu_bin_edges, raw, poles, accum64=accum64, nthread=nthread)
    power *= Lbox**3
    if len(poles) > 0:
        binned_poles *= Lbox**3
    if squeeze_mu_axis and len(mu_bin_edges) == 2:
        power, N_mode, k_avg = power[:, 0], N_mode[:, 0], k_avg[:, 0]
    return dict(power=power, N_mode=N_mode, binned_poles=binned_poles, N_mode_poles=N_mode_poles, k_avg=k_avg)


def calc_power(pos, Lbox, kbins=None, mubins=None, k_max=None, logk=False, paste='TSC', nmesh=128,
               compensated=True, interlaced=True, w=None, pos2=None, w2=None, poles=None, squeeze_mu_axis=True,
               nthread=1, accum64=False):
    """analysis/power_spectrum.py:1131-1319; returns a plain dict with the Table's columns"""
    if kbins is None:
        kbins = nmesh
    if k_max is None:
        k_max = np.pi * nmesh / Lbox
    return_mubins = mubins is not None
    if mubins is None:
        mubins = 1
    W = get_W_compensated(Lbox, nmesh, paste, interlaced) if compensated else None
    f1 = get_field_fft(pos, Lbox, nmesh, paste, w, W, compensated, interlaced, nthread=nthread)
    f2 = None
    if pos2 is not None:
        f2 = get_field_fft(pos2, Lbox, nmesh, paste, w2, W, compensated, interlaced, nthread=nthread)
    poles = np.asarray(poles or [], dtype=np.int64)
    kbins, mubins = get_k_mu_edges(Lbox, k_max, kbins, mubins, logk)
    P = calc_pk_from_deltak(f1, Lbox, kbins, mubins, field2_fft=f2, poles=poles, squeeze_mu_axis=squeeze_mu_axis,
                            nthread=nthread, accum64=accum64)
    res = dict(k_min=kbins[:-1], k_max=kbins[1:], k_mid=(kbins[1:] + kbins[:-1]) * 0.5, k_avg=P['k_avg'],
               power=P['power'], N_mode=P['N_mode'])
    if len(poles) > 0:
        res.update(poles=P['binned_poles'].T, N_mode_poles=P['N_mode_poles'])
    if return_mubins:
        mu_binc = (mubins[1:] + mubins[:-1]) * 0.5
        res.update(mu_min=np.broadcast_to(mubins[:-1], res['power'].shape),
                   mu_max=np.broadcast_to(mubins[1:], res['power'].shape),
                   mu_mid=np.broadcast_to(mu_binc, res['power'].shape))
    return res


# ---------------------------------------------------------------------------
# pair counting (parity unpinned - Corrfunc is third party)
# ---------------------------------------------------------------------------
def paircount_cells(mode, x1, y1, z1, boxsize, bins, x2=None, y2=None, z2=None, pimax=0.0, npibins=0,
                    mu_max=1.0, nmubins=0, nthread=-1):
    """the counts of `paircount_brute` from a cell list (OpenMP over cells): the CPU baseline of bench.py's pair leg"""
    return paircount_brute(mode, x1, y1, z1, boxsize, bins, x2, y2, z2, pimax, npibins, mu_max, nmubins, nthread,
                           _fn='oracle_paircount_cells')


def paircount_brute(mode, x1, y1, z1, boxsize, bins, x2=None, y2=None, z2=None, pimax=0.0, npibins=0,
                    mu_max=1.0, nmubins=0, nthread=-1, _fn='oracle_paircount_brute'):
    """mode 'r' | 'rppi' | 'smu' ; float32 inputs as analysis/tpcf_corrfunc.py:134-139 casts them"""
    m = {'r': 0, 'rppi': 1, 'smu': 2}[mode]
    if nthread < 0:
        nthread = max_threads()
    f4 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    x1, y1, z1, x2, y2, z2 = map(f4, (x1, y1, z1, x2, y2, z2))
    bins = f4(bins)
    nb = len(bins) - 1
    nsub = 1 if m == 0 else (npibins if m == 1 else nmubins)
    out = np.zeros(nb * nsub, dtype=np.uint64)
    auto = x2 is None
    getattr(lib(), _fn)(m, int(auto), _ptr(x1), _ptr(y1), _ptr(z1), C.c_int64(len(x1)),
                                 _ptr(x2), _ptr(y2), _ptr(z2), C.c_int64(0 if auto else len(x2)),
                                 C.c_float(boxsize), _ptr(bins), nb, C.c_float(pimax), int(npibins),
                                 C.c_float(mu_max), int(nmubins), int(nthread), _ptr(out))
    return out


# ------------------------------------------------------------------------------------------------------------------
# NFW satellites: the deterministic part of gen_sats_nfw (hod/GRAND_HOD.py:634-706) - the Poisson MEAN per halo and
# tracer - restated in NumPy.  The draws themselves come from Numba's unseeded generators in the reference, so tests
# compare distributions (counts against these means, radii against the NFW_draw table, velocities against N(0, sig)).
def nfw_expected_counts(halo_data, tracers, params, keep_cent):
    from scipy.special import erfc
    p = marshal_params(tracers, params, False, True)
    m = _f8(halo_data['hmass'])
    z = np.zeros_like(m)
    dc, fe, sh = (_f8(halo_data.get(k, z)) for k in ('hdeltac', 'hfenv', 'hshear'))
    out = {}

    def n_sat_generic(M_h, M_cut, kappa, M_1, alpha, A_s=1.0):   # N_sat_generic / N_sat_elg (:45-65)
        x = M_h - kappa * M_cut
        with np.errstate(invalid='ignore'):
            return np.where(x < 0, 0.0, A_s * (np.maximum(x, 0) / M_1) ** alpha)

    if p.want_LRG:
        M1 = 10 ** (p.L_logM1 + p.L_Asat * dc + p.L_Bsat * fe)
        lc = p.L_logM_cut + p.L_Acent * dc + p.L_Bcent * fe
        ncen = 0.5 * erfc((lc - np.log10(m)) / (1.41421356 * p.L_sigma))
        out['LRG'] = n_sat_generic(m, 10 ** lc, p.L_kappa, M1, p.L_alpha) * ncen * p.L_ic   # n_sat_LRG_modified (:23-34)
    if p.want_ELG:
        M1 = 10 ** (p.E_logM1 + p.E_Asat * dc + p.E_Bsat * fe + p.E_Csat * sh)
        lc = p.E_logM_cut + p.E_Acent * dc + p.E_Bcent * fe + p.E_Ccent * sh
        alpha = np.full_like(m, p.E_alpha)
        kc = np.asarray(keep_cent)
        M1 = np.where(kc == 1, 10 ** (p.E_logM1_EL + p.E_Asat * dc + p.E_Bsat * fe), M1)
        alpha = np.where(kc == 1, p.E_alpha_EL, alpha)
        M1 = np.where(kc == 2, 10 ** (p.E_logM1_EE + p.E_Asat * dc + p.E_Bsat * fe), M1)
        alpha = np.where(kc == 2, p.E_alpha_EE, alpha)
        out['ELG'] = n_sat_generic(m, 10 ** lc, p.E_kappa, M1, alpha, p.E_A_s) * p.E_ic
    if p.want_QSO:
        M1 = 10 ** (p.Q_logM1 + p.Q_Asat * dc + p.Q_Bsat * fe)
        lc = p.Q_logM_cut + p.Q_Acent * dc + p.Q_Bcent * fe
        out['QSO'] = n_sat_generic(m, 10 ** lc, p.Q_kappa, M1, p.Q_alpha) * p.Q_ic
    return out


def compute_ngal_numpy(ball, tracers=None):
    """AbacusHOD.compute_ngal (hod/abacus_hod.py:861-1179) restated in NumPy: the reference's triple / quadruple sums over
    the weighted halo histograms `ball.halo_mass_func` (100^3) and `ball.halo_mass_func_wshear` (100^4), cell centres
    as abscissae.  `ball`: any object with logMbins, deltacbins, fenvbins, shearbins, z_mock and the two histograms."""
    import math
    if tracers is None:
        tracers = ball.tracers
    ngal_dict, fsat_dict = {}, {}
    erfc = np.vectorize(math.erfc)
    erf = np.vectorize(math.erf)
    logMs = 0.5 * (ball.logMbins[1:] + ball.logMbins[:-1])
    deltacs = 0.5 * (ball.deltacbins[1:] + ball.deltacbins[:-1])
    fenvs = 0.5 * (ball.fenvbins[1:] + ball.fenvbins[:-1])
    shears = 0.5 * (ball.shearbins[1:] + ball.shearbins[:-1])
    for etracer, hod in tracers.items():
        Delta_a = 1.0 / (1 + ball.z_mock) - 1.0 / (1 + hod.get('z_pivot', ball.z_mock))
        logM_cut = hod['logM_cut'] + hod.get('logM_cut_pr', 0) * Delta_a
        logM1 = hod['logM1'] + hod.get('logM1_pr', 0) * Delta_a
        Ac, As, Bc, Bs = (hod.get(k, 0) for k in ('Acent', 'Asat', 'Bcent', 'Bsat'))
        ic = hod.get('ic', 1)
        Mh = (10 ** logMs)[:, None, None]
        lc = logM_cut + Ac * deltacs[None, :, None] + Bc * fenvs[None, None, :]
        M1 = 10 ** (logM1 + As * deltacs[None, :, None] + Bs * fenvs[None, None, :])
        if etracer == 'LRG':
            ncent = 0.5 * erfc((lc - np.log10(Mh)) / (1.41421356 * hod['sigma']))
            base = Mh - hod['kappa'] * 10**lc
            nsat = np.where(base < 0, 0.0, (np.maximum(base, 0) / M1) ** hod['alpha'] * ncent)
            ngal_cent = np.sum(ball.halo_mass_func * ncent * ic)
            ngal_sat = np.sum(ball.halo_mass_func * nsat * ic)
        elif etracer == 'QSO':
            ncent = 0.5 * (1 + erf((np.log10(Mh) - lc) / 1.41421356 / hod['sigma']))
            base = Mh - hod['kappa'] * 10**lc
            nsat = np.where(base < 0, 0.0, (np.maximum(base, 0) / M1) ** hod['alpha'])
            ngal_cent = np.sum(ball.halo_mass_func * ncent * ic)
            ngal_sat = np.sum(ball.halo_mass_func * nsat * ic)
        elif etracer == 'ELG':
            Cc, Cs = hod.get('Ccent', 0), hod.get('Csat', 0)
            A_s = hod.get('A_s', 1)
            logM1_EE, alpha_EE = hod.get('logM1_EE', hod['logM1']), hod.get('alpha_EE', hod['alpha'])
            ngal_cent = ngal_sat = 0.0
            hmf = ball.halo_mass_func_wshear
            logMh = np.log10(Mh)
            for el, sh in enumerate(shears):  # one shear slice at a time keeps the temporaries at 100^3
                lce = lc + Cc * sh
                M1e = 10 ** (logM1 + As * deltacs[None, :, None] + Bs * fenvs[None, None, :] + Cs * sh)
                phi = 0.3989422804014327 / hod['sigma'] * np.exp(-((logMh - lce) ** 2) / 2 / hod['sigma'] ** 2)
                Phi = 0.5 * (1 + erf(hod['gamma'] * (logMh - lce) / hod['sigma'] / np.sqrt(2)))
                ncent = 2.0 * (hod['p_max'] - 1.0 / hod['Q']) * phi * Phi * ic
                base = Mh - hod['kappa'] * 10**lce
                nsat = np.where(base < 0, 0.0, A_s * (np.maximum(base, 0) / M1e) ** hod['alpha']) * ic
                M1c = 10 ** (logM1_EE + As * deltacs[None, :, None] + Bs * fenvs[None, None, :] + Cs * sh)
                nconf = np.where(base < 0, 0.0, A_s * (np.maximum(base, 0) / M1c) ** alpha_EE) * ic
                w = hmf[:, :, :, el]
                ngal_cent += np.sum(w * ncent)
                ngal_sat += np.sum(w * (nsat * (1 - ncent) + nconf * ncent))
        else:
            continue
        ngal_dict[etracer] = ngal_cent + ngal_sat
        fsat_dict[etracer] = ngal_sat / (ngal_cent + ngal_sat)
    return ngal_dict, fsat_dict



# ------------------------------------------------------------------------------------------------------------------
# ZCV-facing spectrum helpers (analysis/power_spectrum.py:303-660), NumPy restatements pinned by
# tests/golden/power_helpers.npz (outputs of the shimmed reference).
def _fold2(n):
    i = np.arange(n, dtype=np.int64)
    return np.where(i < n // 2, i * i, (i - n) ** 2)


def bin_kppi(n1d, L, kedges, pimax, Npi, weights, fourier=True):
    """bin_kppi (:303-412): mean of `weights` and mode counts in (k_perp, pi) bins.  Quirks kept: for every i the j loop
    BREAKS at the first j whose k_perp^2 reaches the last edge (the negative-frequency half behind it is never visited,
    :383-384); k_perp bins are (lo, hi] with bin 0 closed below; the kz loop breaks at the last pi edge."""
    kzlen = n1d // 2 + 1
    Nk = len(kedges) - 1
    dk = 2.0 * np.pi / L if fourier else L / n1d
    ke2 = ((np.asarray(kedges) / dk) ** 2).astype(np.float32)
    pe2 = ((np.linspace(0.0, pimax, Npi + 1) / dk) ** 2).astype(np.float32)
    f2 = _fold2(n1d)
    kp2 = (f2[:, None] + f2[None, :]).astype(np.float32)                 # (i, j)
    over = kp2 >= ke2[-1]
    visited = np.cumsum(over, axis=1) == 0                                # j before the first `break`
    use = visited & (kp2 >= ke2[0])
    bk = np.searchsorted(ke2[1:], kp2, side='left')                       # while kmag2 > kedges2[bk+1]: bk += 1
    kz2 = (np.arange(kzlen, dtype=np.int64) ** 2).astype(np.float32)
    zuse = kz2 < pe2[-1]
    bpi = np.searchsorted(pe2[1:], kz2, side='left')
    wz = np.where(np.arange(kzlen) == 0, 1, 2)
    counts = np.zeros((Nk, Npi), dtype=np.int64)
    wsum = np.zeros((Nk, Npi), dtype=np.float64)
    w = np.asarray(weights)[:, :, :kzlen].astype(np.float64)
    ii, jj = np.nonzero(use)
    for kz in np.nonzero(zuse)[0]:
        np.add.at(counts, (bk[ii, jj], bpi[kz]), wz[kz])
        np.add.at(wsum, (bk[ii, jj], bpi[kz]), wz[kz] * w[ii, jj, kz])
    mean = np.where(counts > 0, wsum / np.maximum(counts, 1), wsum)
    return mean.astype(np.float32), counts


def project_3d_to_poles(k_bin_edges, raw_p3d, Lbox, poles, nthread=1):
    """(:415-448): bin_kmu of a caller-supplied 3-D power with one mu bin, times L^3"""
    n = raw_p3d.shape[0]
    r = bin_kmu(n, Lbox, np.asarray(k_bin_edges, dtype=np.float64), np.array([0.0, 1.0]),
                np.ascontiguousarray(raw_p3d, dtype=np.float32), poles=np.asarray(poles, dtype=np.int64), accum64=True,
                nthread=nthread)
    return r[2] * np.float32(Lbox**3), r[3]


def pk_to_xi(Pk, Lbox, r_bins, poles=(0, 2, 4), nthread=1):
    """(:620-660): Xi = irfftn(Pk); multipoles of Xi in r bins (bin_kmu with fourier=False), times nmesh^3"""
    from scipy.fft import irfftn
    Xi = irfftn(Pk, workers=nthread).real
    n = Xi.shape[0]
    r_binc = (np.asarray(r_bins)[1:] + np.asarray(r_bins)[:-1]) * 0.5
    r = bin_kmu(n, Lbox, np.asarray(r_bins, dtype=np.float64), np.array([0.0, 1.0]),
                np.ascontiguousarray(Xi[:, :, : n // 2 + 1], dtype=np.float32), poles=np.asarray(poles, dtype=np.int64),
                fourier=False, accum64=True, nthread=nthread)
    return r_binc, r[2] * np.float32(n**3), r[3]


def _mu2_kmag2(n1d):
    f2 = _fold2(n1d)
    kz = np.arange(n1d // 2 + 1, dtype=np.int64)
    kmag2 = (f2[:, None, None] + f2[None, :, None] + (kz * kz)[None, None, :]).astype(np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        mu2 = np.where(kmag2 > 0, (kz * kz).astype(np.float32)[None, None, :] * (np.float32(1) / kmag2), np.float32(0))
    return mu2.astype(np.float32), kmag2


def get_delta_mu2(delta, n1d):
    """(:580-617)"""
    mu2, _ = _mu2_kmag2(n1d)
    return (delta * mu2).astype(np.complex64)


def get_smoothing(n1d, L, R):
    """(:527-577): exp(-k^2 R^2 / 2) in float32"""
    _, kmag2 = _mu2_kmag2(n1d)
    dk2 = np.float32(np.float32(2.0 * np.pi / L) ** 2)
    R2 = np.float32(R**2)
    return np.exp(-kmag2 * dk2 * R2 / np.float32(2.0)).astype(np.float32)


def expand_poles_to_3d(k_ell, P_ell, n1d, L, poles):
    """(:451-505): sum_l interp(P_l)(|k|) * P_l(mu) on the fundamental modes; linear_interp (:508-537) clamps at the ends"""
    from scipy.special import eval_legendre
    mu2, kmag2 = _mu2_kmag2(n1d)
    dk = np.float32(2.0 * np.pi / L)
    k_ell = np.asarray(k_ell).astype(np.float32)
    P_ell = np.asarray(P_ell).astype(np.float32)
    kk = np.sqrt(kmag2) * dk
    dx = k_ell[1] - k_ell[0]
    f = (kk - k_ell[0]) / dx
    fl = np.clip(f.astype(np.int64), 0, len(k_ell) - 2)
    out = np.zeros_like(kmag2, dtype=np.float32)
    mu = np.sqrt(mu2.astype(np.float64))
    for ip, ell in enumerate(poles):
        y = P_ell[ip]
        yd = y[fl] + (f - fl) * (y[fl + 1] - y[fl])
        yd = np.where(kk <= k_ell[0], y[0], np.where(kk >= k_ell[-1], y[-1], yd))
        out += (yd * (1.0 if ell == 0 else eval_legendre(int(ell), mu))).astype(np.float32)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# catalogue side: bit-unpacking and the local mass environment
# ---------------------------------------------------------------------------------------------------------------------
def unpack_rvint(intdata, boxsize, float_dtype=np.float32):
    """`_unpack_rvint` (abacusnbody/data/bitpacked.py:100-116): pos = (x >> 12) * (boxsize / 1e6), vel = ((x & 0xFFF) -
    2048) * (6000 / 2048); the shift is arithmetic (int32), products are int64 * float64, rounded once on store."""
    x = np.asarray(intdata, dtype=np.int32).reshape(-1, 3).astype(np.int64)
    pos = ((x >> 12).astype(np.float64) * (float(boxsize) / 1e6)).astype(float_dtype)
    vel = (((x & 0xFFF) - 2048).astype(np.float64) * (6000.0 / 2048)).astype(float_dtype)
    return pos, vel


def unpack_pids(packed, box=1.0, ppd=1, float_dtype=np.float32):
    """`_unpack_pids` (bitpacked.py:274-330), every field.  lagr_pos: uint64 * float_dtype(box / ppd) is float64
    arithmetic under both NumPy and Numba promotion; one rounding on store."""
    a = np.asarray(packed, dtype=np.uint64)
    ft = np.dtype(float_dtype).type
    inv_ppd, half = np.float64(ft(float(box) / ppd)), np.float64(ft(float(box) / 2))
    idx = np.stack([a & np.uint64(0x7FFF), (a & np.uint64(0x7FFF0000)) >> np.uint64(16),
                    (a & np.uint64(0x7FFF00000000)) >> np.uint64(32)], axis=1)
    d = ((a & np.uint64(0x07FE000000000000)) >> np.uint64(49)).astype(np.int64)
    return {'pid': (a & np.uint64(0x7FFF7FFF7FFF)).astype(np.int64),
            'lagr_idx': idx.astype(np.int16),
            'lagr_pos': (idx.astype(np.float64) * inv_ppd - half).astype(float_dtype),
            'tagged': ((a >> np.uint64(48)) & np.uint64(1)).astype(np.uint8),
            'density': (d * d).astype(float_dtype)}


def menv_brute(pos, mass, r_inner, r_outer, halo_lc, Lbox, mcut=1e11, chunk=512):
    """`do_Menv_from_tree` (abacusnbody/hod/menv.py:19-87) without the tree: all-pairs float64 distances (minimum image
    when periodic, like KDTree(boxsize=Lbox)), neighbours with d <= r, masses summed in float64."""
    pos = np.asarray(pos)
    mass = np.asarray(mass)
    if not halo_lc:
        pos = (pos + Lbox / 2.0) % Lbox
    p = pos.astype(np.float64)
    m = mass.astype(np.float64)
    n = len(p)
    ri = np.broadcast_to(np.asarray(r_inner, dtype=np.float64), (n,))
    ro = np.broadcast_to(np.asarray(r_outer, dtype=np.float64), (n,))
    out = np.zeros(n, dtype=np.float64)
    cen = np.nonzero(mass > mcut)[0]
    for s in range(0, len(cen), chunk):
        c = cen[s:s + chunk]
        d = np.abs(p[c, None, :] - p[None, :, :])
        if not halo_lc:
            d = np.where(d > 0.5 * Lbox, Lbox - d, d)
        d2 = (d * d).sum(axis=2)
        out[c] = ((d2 <= (ro[c] ** 2)[:, None]) * m[None, :]).sum(axis=1) - \
                 ((d2 <= (ri[c] ** 2)[:, None]) * m[None, :]).sum(axis=1)
    res = np.zeros_like(mass)
    res[cen] = out[cen]
    return res


def unpack_pack9(data, boxsize, velzspace_to_kms, float_dtype=np.float32):
    """`_unpack_pack9` (abacusnbody/data/pack9.py:59-123), vectorised: every record takes the state of the last header at
    or before it.  Typing as Numba compiles it (pinned by the reference's tests/ref_data/test_pack9.asdf): invcpd, the
    cell centres and pscale are float64 expressions rounded to `float_dtype`; `short * pscale + centre` and
    `short * vscale` are `float_dtype` arithmetic.  Records ahead of the first header decode to NaN."""
    F = np.dtype(float_dtype).type
    d = np.asarray(data, dtype=np.uint8).reshape(-1, 9)
    c = d.astype(np.int64)
    sh = np.stack([(c[:, 1] & 0x0F) | (c[:, 0] << 4), ((c[:, 1] & 0xF0) << 4) | c[:, 2],
                   (c[:, 4] & 0x0F) | (c[:, 3] << 4), ((c[:, 4] & 0xF0) << 4) | c[:, 5],
                   (c[:, 7] & 0x0F) | (c[:, 6] << 4), ((c[:, 7] & 0xF0) << 4) | c[:, 8]], axis=1) - 2048
    hdr = d[:, 0] == 0xFF
    hidx = np.maximum.accumulate(np.where(hdr, np.arange(len(d)), -1))
    H = sh[np.maximum(hidx, 0)]
    box, velz = F(boxsize), F(velzspace_to_kms)
    with np.errstate(divide='ignore', invalid='ignore'):
        invcpd = (1.0 / (H[:, 1] + 2000)).astype(F)
        csize = box * invcpd
        halfbox = np.float64(box) / 2
        vscale = ((H[:, 2] + 2000) * 0.0005).astype(F) * invcpd * velz
        cell = [(((H[:, 3 + k] + 2000.5) * csize.astype(np.float64)) - halfbox).astype(F) for k in range(3)]
        pscale = (0.0005 * csize.astype(np.float64)).astype(F)
        nohdr = hidx < 0
        for a in (vscale, pscale, *cell):
            a[nohdr] = np.nan
        pos = np.stack([sh[:, k].astype(F) * pscale + cell[k] for k in range(3)], axis=1)
        vel = np.stack([sh[:, 3 + k].astype(F) * vscale for k in range(3)], axis=1)
    part = ~hdr
    return pos[part], vel[part]


# ---- reseed stream (hod/abacus_hod.py:775-839): Philox4x32-10 + fixed float64 transforms --------------------------------
def philox4x32_10(ctr, key):
    """one block of Philox4x32-10 (C restatement); ctr (4,), key (2,) uint32 -> (4,) uint32"""
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().oracle_philox4x32_10(_ptr(c), _ptr(k), _ptr(out))
    return out


def philox4x32_10_py(ctr, key):
    """the same in pure Python straight from the published round function (Salmon et al. 2011, Random123): an
    independent check of the C restatement on the known-answer vectors"""
    c = [int(x) for x in ctr]
    k = [int(x) for x in key]
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
    return np.array(c, dtype=np.uint32)


def rs_log(x):
    lib().oracle_rs_log.restype = C.c_double
    return float(lib().oracle_rs_log(C.c_double(float(x))))


def rs_sincos2pi(t):
    s, c = C.c_double(0), C.c_double(0)
    lib().oracle_rs_sincos2pi(C.c_double(float(t)), C.byref(s), C.byref(c))
    return s.value, c.value


def reseed(seed, n_halo, n_part, hsigma3d=None, want_expvel=False, halo_index0=0, part_index0=0):
    """the arrays `run_hod(reseed=seed)` rewrites (:824-835), as the build's generator draws them:
    hrandoms (n_halo,), hveldev (n_halo, 3), prandoms (n_part,), all float64 holding float32-precision draws"""
    hr = np.zeros(n_halo, dtype=np.float64)
    hv = np.zeros((n_halo, 3), dtype=np.float64)
    pr = np.zeros(n_part, dtype=np.float64)
    sg = None if hsigma3d is None else _f8(hsigma3d)
    lib().oracle_reseed_halos(C.c_int64(n_halo), C.c_int64(halo_index0), C.c_uint64(int(seed) & (2**64 - 1)), _ptr(sg),
                              int(bool(want_expvel)), _ptr(hr), _ptr(hv))
    lib().oracle_reseed_particles(C.c_int64(n_part), C.c_int64(part_index0), C.c_uint64(int(seed) & (2**64 - 1)), _ptr(pr))
    return hr, hv, pr
