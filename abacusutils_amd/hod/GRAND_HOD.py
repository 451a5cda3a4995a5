"""MI355X drop-in for the reference's `abacusnbody.hod.GRAND_HOD` population entry points.

`gen_gal_cat` / `gen_gals` keep the reference signatures and return structure
(abacusnbody/hod/GRAND_HOD.py:1595-1724, 1302-1592); the central/satellite loops
(`gen_cent` :139-414, `gen_sats` :825-1262) and the concatenation
(`fast_concatenate` :1265-1299) run as HIP kernels behind include/abacus_hip.h
(`abacus_hod_*`).  `Nthread` is accepted and ignored (there are no host threads).

Device residency: `StagedCatalog` uploads the halo/particle subsample once and is
reused for any number of `populate` calls, which is what `AbacusHOD.staging()`
does on the host in the reference (hod/abacus_hod.py:193-197).  `gen_gal_cat`
called with plain dicts stages, populates and frees (stateless, like the
reference); `AbacusHOD.run_hod` keeps the staged catalog alive.
"""
import ctypes as C
import os
import warnings
from pathlib import Path

import numpy as np

from .. import _lib
from .._lib import HodArrays, HodParams, TRACER_KEYS, check, ptr

TRACERS = ('LRG', 'ELG', 'QSO')
COLS = ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass')

_HALO_REQ = ('hpos', 'hvel', 'hmass', 'hid', 'hmultis', 'hrandoms', 'hveldev')
_HALO_OPT = ('hdeltac', 'hfenv', 'hshear')
_PART_REQ = ('ppos', 'pvel', 'phvel', 'phmass', 'phid', 'pweights', 'prandoms')
_PART_OPT = ('pdeltac', 'pfenv', 'pshear', 'pranks', 'pranksv', 'pranksp', 'pranksr', 'pinds')


def marshal_params(tracers, params, enable_ranks, rsd):
    """Parameter handling of gen_gals (hod/GRAND_HOD.py:1342-1475) -> struct abacus_hod_params.

    z-evolution `logM_cut += logM_cut_pr * Delta_a`, `logM1 += logM1_pr * Delta_a` with
    `Delta_a = 1/(1+z) - 1/(1+z_pivot)` (:1362-1371); defaults `Acent..Csat = 0`, `ic = 1`,
    `logM1_EE/EL = logM1`, `alpha_EE/EL = alpha` (:1373-1379,1411-1421,1456-1460).  Keys without a
    default (`alpha_c`, `alpha_s`, `s`, ...) raise KeyError exactly as the typed-dict lookup would.
    """
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
    for tr in tracers.keys():
        if tr not in TRACER_KEYS:
            continue
        pre, keys = TRACER_KEYS[tr]
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
            setattr(p, pre + k, float(hod[k]))
    return p


def _as(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class StagedCatalog:
    """The halo + particle subsample resident in HBM (abacus_hod_stage).

    `halo_data` / `particle_data` are the dicts `AbacusHOD.staging()` builds
    (hod/abacus_hod.py:659-702).  float64 / int64 conversion happens here once.
    """

    def __init__(self, halo_data, particle_data):
        self._h = C.c_void_p()
        self.n_halo = len(halo_data['hmass'])
        self.n_part = len(particle_data['phmass']) if 'phmass' in particle_data else 0
        arrs = HodArrays()
        keep = []  # host copies must outlive the upload call

        vec3 = ('hpos', 'hvel', 'hveldev', 'ppos', 'pvel', 'phvel')

        def put(name, src, dtype, required):
            if name in src and src[name] is not None:
                a = _as(src[name], dtype)
                n = self.n_halo if name.startswith('h') else self.n_part
                if name == 'hveldev' and a.ndim == 1 and a.shape == (n,):
                    a = np.ascontiguousarray(np.repeat(a[:, None], 3, axis=1))   # the scalar form `reseed` accepts (:826-829)
                # a short array would be read past its end by the upload (the reference raises IndexError in its loops)
                want = (n, 3) if name in vec3 else (n,)
                if a.shape != want:
                    raise ValueError(f'{name} has shape {a.shape}, expected {want} (n_halo={self.n_halo}, n_part={self.n_part})')
                keep.append(a)
                setattr(arrs, name, a.ctypes.data)
            elif required:
                raise KeyError(name)

        arrs.n_halo = self.n_halo
        arrs.n_part = self.n_part
        for k in _HALO_REQ:
            put(k, halo_data, np.int64 if k == 'hid' else np.float64, True)
        for k in _HALO_OPT:
            put(k, halo_data, np.float64, False)
        if self.n_part or 'ppos' in particle_data:
            for k in _PART_REQ:
                put(k, particle_data, np.int64 if k == 'phid' else np.float64, True)
            for k in _PART_OPT:
                put(k, particle_data, np.int64 if k == 'pinds' else np.float64, False)
        check(_lib.lib().abacus_hod_stage(C.byref(arrs), 0, C.byref(self._h)))
        self.has_ranks = all(k in particle_data for k in ('pranks', 'pranksv', 'pranksp', 'pranksr'))
        self.counts = None

    def update(self, field, values):
        """re-upload `hrandoms`, `hveldev` or `prandoms` (the arrays `reseed` rewrites, hod/abacus_hod.py:824-835)"""
        a = _as(values, np.float64)
        check(_lib.lib().abacus_hod_update(self._h, field.encode(), ptr(a)))

    def reseed(self, seed, hsigma3d=None, want_expvel=False, halo_index0=0, part_index0=0):
        """rewrite hrandoms / hveldev / prandoms in HBM from the device Philox generator (abacus_hod_reseed);
        `hsigma3d` is staged on the first call"""
        if hsigma3d is not None and not getattr(self, '_sigma_set', False):
            a = _as(hsigma3d, np.float64)
            if len(a) != self.n_halo:
                raise ValueError('hsigma3d must have one value per halo')
            check(_lib.lib().abacus_hod_set_sigma3d(self._h, ptr(a), 0))
            self._sigma_set = True
        check(_lib.lib().abacus_hod_reseed(self._h, C.c_uint64(int(seed) & (2**64 - 1)), int(bool(want_expvel)),
                                           C.c_int64(halo_index0), C.c_int64(part_index0)))

    def fetch_field(self, field):
        """device -> host copy of `hrandoms` (N,), `hveldev` (N,3) or `prandoms` (Np,)"""
        n = {'hrandoms': self.n_halo, 'hveldev': 3 * self.n_halo, 'prandoms': self.n_part}[field]
        out = np.empty(n, dtype=np.float64)
        check(_lib.lib().abacus_hod_fetch_field(self._h, field.encode(), ptr(out)))
        return out.reshape(-1, 3) if field == 'hveldev' else out

    generation = 0    # bumped by every populate: the catalogue columns in HBM belong to the latest one only

    def populate(self, p):
        """decide + emit on the device; returns (Ncent[3], Nsat[3])"""
        self.generation += 1
        counts = (C.c_int64 * 6)()
        check(_lib.lib().abacus_hod_populate(self._h, C.byref(p), counts))
        self.counts = np.array(counts[:], dtype=np.int64)
        return self.counts[:3].copy(), self.counts[3:].copy()

    def populate_nfw(self, p, tracers, halo_data, NFW_draw, seed, halo_index0=0):
        """centrals as usual + NFW satellites (abacus_hod_populate_nfw); stages hsigma3d / hc / hrvir on first use"""
        if not getattr(self, '_sigma_set', False):
            check(_lib.lib().abacus_hod_set_sigma3d(self._h, ptr(_as(halo_data['hsigma3d'], np.float64)), 0))
            self._sigma_set = True
        if not getattr(self, '_profile_set', False):
            check(_lib.lib().abacus_hod_set_profile(self._h, ptr(_as(halo_data['hc'], np.float64)),
                                                    ptr(_as(halo_data['hrvir'], np.float64))))
            self._profile_set = True
        self.generation += 1
        nf = _lib.NfwParams()
        nf.seed = int(seed) & (2**64 - 1)
        for t, tr in enumerate(TRACERS):   # f_sigv defaults to 0 (hod/GRAND_HOD.py:1376,1418,1457)
            nf.f_sigv[t] = float(tracers.get(tr, {}).get('f_sigv', 0.0))
        elg = tracers.get('ELG', {})       # the reference takes these from the ELG dict for every tracer (:606-608)
        nf.exp_frac, nf.exp_scale = float(elg.get('exp_frac', 0.0)), float(elg.get('exp_scale', 1.0))
        nf.nfw_rescale = float(elg.get('nfw_rescale', 1.0))
        nf.halo_index0 = int(halo_index0)
        draw = _as(NFW_draw, np.float64)
        counts = (C.c_int64 * 6)()
        check(_lib.lib().abacus_hod_populate_nfw(self._h, C.byref(p), C.byref(nf), ptr(draw), C.c_int64(len(draw)), counts))
        self.counts = np.array(counts[:], dtype=np.int64)
        return self.counts[:3].copy(), self.counts[3:].copy()

    def populate_async(self, p):
        """enqueue only (bench): no host synchronisation, no result copy"""
        self.generation += 1
        check(_lib.lib().abacus_hod_populate_async(self._h, C.byref(p)))

    def wait_counts(self):
        counts = (C.c_int64 * 6)()
        check(_lib.lib().abacus_hod_counts(self._h, counts))
        self.counts = np.array(counts[:], dtype=np.int64)
        return self.counts

    def candidates(self):
        """diagnostic: (halos, particles) the last populate's filter handed to the exact float64 decision"""
        out = (C.c_int64 * 2)()
        check(_lib.lib().abacus_hod_candidates(self._h, out))
        return int(out[0]), int(out[1])

    def fetch(self, tracer):
        """device -> host copy of one tracer's catalog, in the reference's dict form (:1573-1589)"""
        t = TRACERS.index(tracer)
        n = int(self.counts[t] + self.counts[3 + t])
        block = _lib.pinned_empty((8, n), np.float64)   # one transfer into page-locked memory; the columns are views of it
        check(_lib.lib().abacus_hod_fetch_block(self._h, t, ptr(block), C.c_int64(n)))
        d = {'Ncent': int(self.counts[t])}
        for q, c in enumerate(COLS):
            d[c] = block[q]
        d['id'] = block[7].view(np.int64)
        return d

    def device_columns(self, tracer):
        """device pointers (x,y,z,vx,vy,vz,mass,id) of one tracer's catalog, valid until the next populate"""
        cols = (C.c_void_p * 8)()
        check(_lib.lib().abacus_hod_device_columns(self._h, TRACERS.index(tracer), cols))
        return list(cols)

    def fetch_keep(self):
        kc = np.empty(self.n_halo, dtype=np.int8)
        ks = np.empty(self.n_part, dtype=np.int8)
        check(_lib.lib().abacus_hod_fetch_keep(self._h, ptr(kc), ptr(ks)))
        return kc, ks

    def free(self):
        if self._h:
            _lib.lib().abacus_hod_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class LazyTracer(dict):
    """One tracer's catalogue with the reference's keys ('x' ... 'mass', 'id', 'Ncent') whose columns are still in HBM:
    the device-to-host copy (all eight columns, one transfer) happens at the first access of any of them.  An MCMC step
    that goes populate -> clustering on the device (`compute_power`, `compute_xirppi`, ...) never pays the copy.
    Opt-in (`AbacusHOD.lazy_columns = True`): the columns can only be materialised until the NEXT populate of the same
    staged catalogue rewrites them - accessing a stale, never-read catalogue raises instead of returning wrong galaxies.
    Behaves like a dict once touched; pickles / copies as a plain dict."""

    def __init__(self, staged, tracer, ncent):
        super().__init__(Ncent=int(ncent))
        import weakref
        self._staged, self._generation, self._tracer = weakref.ref(staged), staged.generation, tracer
        for c in COLS + ('id',):
            dict.__setitem__(self, c, None)          # placeholders: the keys exist from the start

    def _load(self):
        ref = self.__dict__.pop('_staged', None)
        if ref is None:
            return
        st = ref()
        if st is None or not st._h or st.generation != self._generation:
            self.__dict__['_staged'] = ref
            raise RuntimeError(f'the {self._tracer} catalogue of this mock was never read and a later run_hod() has replaced '
                               'it in HBM (AbacusHOD.lazy_columns = True): read the columns before the next run_hod(), or '
                               'switch lazy_columns off')
        full = st.fetch(self._tracer)
        for k, v in full.items():
            if k != 'Ncent':
                dict.__setitem__(self, k, v)

    def __getitem__(self, k):
        if k != 'Ncent':
            self._load()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        if k != 'Ncent':
            self._load()
        return dict.get(self, k, default)

    def __iter__(self):            # overriding the iteration makes dict(self) / {**self} go through __getitem__
        return dict.__iter__(self)

    def keys(self):
        return dict.keys(self)

    def items(self):
        self._load()
        return dict.items(self)

    def values(self):
        self._load()
        return dict.values(self)

    def pop(self, k, *default):
        if k != 'Ncent':
            self._load()
        return dict.pop(self, k, *default)

    # writes materialise the columns first: a caller's assignment must not be overwritten by a later load, and the
    # mock then no longer stands for what is in HBM (MockDict.device_xyz compares checksums of the loaded columns)
    def __setitem__(self, k, v):
        self._load()
        dict.__setitem__(self, k, v)

    def __delitem__(self, k):
        self._load()
        dict.__delitem__(self, k)

    def update(self, *a, **kw):
        self._load()
        dict.update(self, *a, **kw)

    def setdefault(self, k, default=None):
        self._load()
        return dict.setdefault(self, k, default)

    def clear(self):
        self.__dict__.pop('_staged', None)
        dict.clear(self)

    def popitem(self):
        self._load()
        return dict.popitem(self)

    def copy(self):
        self._load()
        return dict(dict.items(self))

    def __eq__(self, other):
        self._load()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        if '_staged' in self.__dict__:
            return f'<LazyTracer {self._tracer}: Ncent={dict.__getitem__(self, "Ncent")}, columns in HBM>'
        return dict.__repr__(self)

    def __reduce__(self):
        self._load()
        return (dict, (dict(dict.items(self)),))


class MockDict(dict):
    """The `mock_dict` of run_hod / gen_gal_cat - a plain dict of NumPy columns, exactly the reference's - that also
    remembers where the same columns still sit in HBM, so that the clustering step (compute_power, compute_xirppi, ...)
    can start from there instead of uploading them again.  `device_xyz(tracer)` returns None as soon as that is no
    longer safe: a later populate rewrote the device catalogue, the staged catalogue was freed, or the host columns were
    replaced or modified IN ANY ROW - a position-dependent checksum of the whole host column is compared with the same
    checksum of the device column (one pass over 3 n values on either side, cheaper than the upload it saves; nothing is
    computed when the mock is created).  Pickles and copies as a plain dict."""

    def _bind(self, staged):
        import weakref
        self._staged = weakref.ref(staged)
        self._generation = staged.generation
        return self

    def __reduce__(self):
        return (dict, ({k: (dict(v) if isinstance(v, dict) else v) for k, v in dict.items(self)},))

    def device_xyz(self, tracer):
        st = getattr(self, '_staged', lambda: None)()
        if st is None or not st._h or st.generation != self._generation or tracer not in self:
            return None
        d = self[tracer]
        lazy = isinstance(d, LazyTracer) and '_staged' in d.__dict__     # never copied to the host: nothing to compare
        n = int(st.counts[TRACERS.index(tracer)] + st.counts[3 + TRACERS.index(tracer)])
        if n == 0:
            return None
        cols = st.device_columns(tracer)
        if not lazy:
            try:
                for q, c in enumerate(('x', 'y', 'z')):
                    a = d[c]
                    if not isinstance(a, np.ndarray) or a.shape != (n,) or a.dtype != np.float64:
                        return None
                    if _lib.poshash_host(a) != _lib.poshash_device(cols[q], n):
                        return None
            except (KeyError, TypeError):
                return None
        return [_lib.DeviceArray.view(cols[q], np.float64, (n,)) for q in range(3)]


def gen_gals(halos_array, subsample, tracers, params, Nthread, enable_ranks, rsd, verbose, nfw, NFW_draw=None,
             staged=None, lazy=False):
    """hod/GRAND_HOD.py:1302-1592.  `staged`: a StagedCatalog to reuse (extension); otherwise the arrays are
    uploaded for this call only.  `lazy` (with `staged`): the columns stay in HBM until first read (LazyTracer)."""
    p = marshal_params(tracers, params, enable_ranks, rsd)
    own = staged is None
    if own:
        staged = StagedCatalog(halos_array, subsample)
    try:
        if enable_ranks and not staged.has_ranks:
            raise KeyError('pranks')
        if nfw:
            # NFW satellites (gen_sats_nfw, :522-822).  The reference draws them from Numba's unseeded per-thread
            # generators; here the Philox key comes from NumPy's global generator, so `np.random.seed` makes a run
            # reproducible.  Parity with the reference is statistical by construction.
            warnings.warn('NFW profile is unoptimized. It has different velocity bias. It does not support lightcone.')
            if NFW_draw is None:
                raise ValueError('nfw=True needs NFW_draw (the table of NFW radial draws in units of r_s)')
            for k in ('hsigma3d', 'hc', 'hrvir'):
                if k not in halos_array:
                    raise KeyError(k)
            seed = nfw if (isinstance(nfw, (int, np.integer)) and not isinstance(nfw, (bool, np.bool_))) else \
                int(np.random.randint(0, 2**62))
            ncent, nsat = staged.populate_nfw(p, tracers, halos_array, NFW_draw, seed)
        else:
            ncent, nsat = staged.populate(p)
        HOD_dict = {}
        for tracer in tracers:
            if tracer not in TRACERS:
                continue
            t = TRACERS.index(tracer)
            HOD_dict[tracer] = LazyTracer(staged, tracer, ncent[t]) if (lazy and not own and not verbose) else staged.fetch(tracer)
            if verbose:
                n = len(HOD_dict[tracer]['x'])
                print(tracer, 'number of galaxies ', n)
                print('satellite fraction ', (n - HOD_dict[tracer]['Ncent']) / n if n else 0.0)
        if not own:   # the catalogue stays in HBM behind a resident StagedCatalog: let the clustering step find it there
            HOD_dict = MockDict(HOD_dict)._bind(staged)
    finally:
        if own:
            staged.free()
    return HOD_dict


def _write_ecsv(path, cols, meta):
    """ECSV 1.0 table as `astropy.io.ascii.write(format='ecsv')` lays it out (hod/GRAND_HOD.py:1708-1722)"""
    names = list(cols.keys())
    with open(path, 'w') as f:
        f.write('# %ECSV 1.0\n# ---\n# datatype:\n')
        for n in names:
            dt = 'int64' if np.issubdtype(np.asarray(cols[n]).dtype, np.integer) else 'float64'
            f.write(f'# - {{name: {n}, datatype: {dt}}}\n')
        f.write('# meta: !!omap\n')
        for k, v in meta.items():
            if isinstance(v, (bool, np.bool_)):
                v = 'true' if v else 'false'
            elif isinstance(v, (np.floating, np.integer)):
                v = v.item()
            f.write(f'# - {{{k}: {v}}}\n')
        f.write('# schema: astropy-2.0\n')
        f.write(' '.join(names) + '\n')
        arrs = [np.asarray(cols[n]) for n in names]
        for i in range(len(arrs[0]) if arrs else 0):
            f.write(' '.join(repr(a[i].item()) for a in arrs) + '\n')


def gen_gal_cat(halo_data, particle_data, tracers, params, Nthread=16, enable_ranks=False, rsd=True, nfw=False,
                NFW_draw=None, write_to_disk=False, savedir='./', verbose=False, fn_ext=None, staged=None, lazy=False):
    """Drop-in for hod/GRAND_HOD.py:1595-1724: returns {tracer: {'x','y','z','vx','vy','vz','mass' (float64),
    'id' (int64), 'Ncent' (int)}} with centrals first; optionally writes `{tracer}s.dat` ECSV files."""
    if not isinstance(rsd, bool):
        raise ValueError('Error: rsd has to be a boolean')

    HOD_dict = gen_gals(halo_data, particle_data, tracers, params, Nthread, enable_ranks, rsd, verbose, nfw,
                        NFW_draw, staged=staged, lazy=lazy and not write_to_disk)

    if write_to_disk and tracers:
        rsd_string = '_rsd' if rsd else ''
        savedir = Path(savedir)
        outdir = savedir / ('galaxies' + rsd_string + (fn_ext or ''))
        os.makedirs(outdir, exist_ok=True)

    for tracer in tracers.keys():
        if not (verbose or write_to_disk):
            continue
        Ncent = HOD_dict[tracer]['Ncent']
        if verbose:
            n = len(HOD_dict[tracer]['x'])
            print('generated %ss:' % tracer, n, 'satellite fraction ', 1 - Ncent / n if n else 0.0)
        if write_to_disk:
            HOD_dict[tracer].pop('Ncent', None)  # the reference drops it from the returned dict too (:1707)
            meta = {'Ncent': Ncent, 'Gal_type': tracer, **tracers[tracer]}
            if params['chunk'] == -1:
                fn = outdir / f'{tracer}s.dat'
            else:
                fn = outdir / f'{tracer}s_chunk{params["chunk"]:d}.dat'
            _write_ecsv(fn, HOD_dict[tracer], meta)
    return HOD_dict
