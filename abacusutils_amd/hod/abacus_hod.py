"""MI355X drop-in for `abacusnbody.hod.abacus_hod.AbacusHOD` (reference: abacusnbody/hod/abacus_hod.py).

Same constructor, attributes and method signatures for the calls on the hot path:

    run_hod            (:706-859)   -> gen_gal_cat on the device-resident subsample (csrc/hod.hip)
    compute_power      (:1338-1472) -> calc_power (csrc/tsc.hip, power.hip)
    compute_xirppi / compute_wp / compute_multipole / compute_clustering  (:1181-1336, :1826-1885) -> csrc/pairs.hip
    compute_ngal       (:861-1179)  the same sums as one pass over the staged halos on the device (abacus_hod_ngal)

The halo/particle subsample is uploaded to HBM on the first `run_hod` and stays there (the reference keeps it in
host RAM across calls, :193-197); `reseed` rewrites the three random arrays in HBM with the device Philox generator
and mirrors them to the host dicts (:824-835).
`AbacusHOD.from_arrays` builds the object from in-memory arrays (tests, synthetic data); the regular constructor
runs `staging()` (needs h5py for the `halos_xcom_*`/`particles_xcom_*` files).
ZCV (`apply_zcv*`) is outside the hot path and not provided.
"""
import logging
import math
import time
from pathlib import Path

import numpy as np

from ..analysis.power_spectrum import calc_power
from ..analysis.tpcf_corrfunc import calc_multipole_fast, calc_wp_fast, calc_xirppi_fast
from .GRAND_HOD import StagedCatalog, gen_gal_cat

_PRIMARY_Z = [3.0, 2.5, 2.0, 1.7, 1.4, 1.1, 0.8, 0.5, 0.4, 0.3, 0.2, 0.1, 0.0]
_SECONDARY_Z = [0.15, 0.25, 0.35, 0.45, 0.575, 0.65, 0.725, 0.875, 0.95, 1.025, 1.175, 1.25, 1.325, 1.475, 1.55,
                1.625, 1.85, 2.25, 2.75, 3.0, 5.0, 8.0]


def _read_asdf_header(fn):
    """`header` mapping of an Abacus ASDF file without the asdf package: the tree is YAML ahead of the first block"""
    import re

    import yaml
    raw = open(fn, 'rb').read()
    end = raw.find(b'\n...')
    text = raw[: end if end >= 0 else len(raw)].decode('utf-8', errors='replace')
    text = re.sub(r'!\S+', '', text)  # drop YAML tags (core/asdf, core/ndarray, ...)
    text = '\n'.join(line for line in text.splitlines() if not line.startswith('%') and not line.startswith('#'))
    tree = yaml.safe_load(text)
    return tree['header']


def _i8(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _searchsorted(a, b):
    """`_searchsorted_parallel` (:588): np.searchsorted(a, b) on the device (abacus_searchsorted_i64)"""
    import ctypes as C

    from .. import _lib
    a, b = _i8(a), _i8(b)
    out = np.empty(len(b), dtype=np.int64)
    _lib.check(_lib.lib().abacus_searchsorted_i64(_lib.ptr(a), C.c_int64(len(a)), _lib.ptr(b), C.c_int64(len(b)), _lib.ptr(out)))
    return out


def _argsort_ids(ids):
    """the halo sort by id (:566-585) on the device (abacus_argsort_i64: stable radix sort)"""
    import ctypes as C

    from .. import _lib
    ids = _i8(ids)
    out = np.empty(len(ids), dtype=np.int64)
    _lib.check(_lib.lib().abacus_argsort_i64(_lib.ptr(ids), C.c_int64(len(ids)), _lib.ptr(out)))
    return out


def calc_fenv_opt(Menv, mbins, halosM):
    """global environment rank per mass bin (:1961-1970) on the device (abacus_fenv_rank)"""
    import ctypes as C

    from .. import _lib
    f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)   # noqa: E731
    Menv, mbins, halosM = f8(Menv), f8(mbins), f8(halosM)
    out = np.empty(len(Menv), dtype=np.float64)
    _lib.check(_lib.lib().abacus_fenv_rank(_lib.ptr(Menv), _lib.ptr(halosM), C.c_int64(len(Menv)), _lib.ptr(mbins), len(mbins),
                                         _lib.ptr(out)))
    return out


def _concat_tables(tabs):
    """rows of the slabs' subsample tables: HDF5 compound arrays or dicts of columns"""
    if isinstance(tabs[0], dict):
        return {k: np.concatenate([np.asarray(t[k]) for t in tabs]) for k in tabs[0]}
    return np.concatenate(tabs)


def _table_fields(tab):
    return tab.keys() if isinstance(tab, dict) else tab.dtype.fields.keys()


class AbacusHOD:
    """A multi-tracer HOD code for the AbacusSummit simulations (MI355X path)."""

    # Extensions of the reference class (both off = the reference's behaviour):
    # lazy_columns: run_hod() returns at once with the galaxy columns still in HBM; they are copied to NumPy at the first
    #   read (GRAND_HOD.LazyTracer) and only until the next run_hod() - an MCMC step that feeds the mock straight into
    #   compute_power / compute_xirppi / compute_wp / compute_multipole never pays the PCIe copy.
    # reseed_sync_host: after `run_hod(reseed=...)` copy the redrawn random columns back into halo_data / particle_data
    #   like the reference mutates them (True), or leave them on the device only (False).
    lazy_columns = False
    reseed_sync_host = True

    def __init__(self, sim_params, HOD_params, clustering_params=None, chunk=-1, n_chunks=1, skip_staging=False):
        self.logger = logging.getLogger('AbacusHOD')
        self.sim_name = sim_params['sim_name']
        self.sim_dir = sim_params['sim_dir']
        self.subsample_dir = sim_params['subsample_dir']
        self.z_mock = sim_params['z_mock']
        self.output_dir = sim_params.get('output_dir', './')
        self.halo_lc = sim_params.get('halo_lc', False)
        self.force_mt = sim_params.get('force_mt', False)
        self.local_env = sim_params.get('local_env', {})

        if self.halo_lc:
            ztype = 'lightcone'
        elif self.z_mock in _PRIMARY_Z:
            ztype = 'primary'
        elif self.z_mock in _SECONDARY_Z:
            ztype = 'secondary'
        else:
            raise Exception('illegal redshift')
        self.z_type = ztype
        self._init_hod(HOD_params, clustering_params)
        self.chunk = chunk
        self.n_chunks = n_chunks
        assert self.chunk < self.n_chunks, 'Total number of chunks needs to be larger than current chunk index'
        self._staged = None
        if not skip_staging:
            self.halo_data, self.particle_data, self.params, self.mock_dir = self.staging()
            self._build_mass_function()
        else:
            raise NotImplementedError('skip_staging=True needs abacusnbody.metadata (out of the hot-path scope); '
                                      'use AbacusHOD.from_arrays for in-memory data')

    def _init_hod(self, HOD_params, clustering_params):
        tracer_flags = HOD_params['tracer_flags']
        tracers = {}
        for key in tracer_flags.keys():
            if tracer_flags[key]:
                tracers[key] = HOD_params[key + '_params']
        self.tracers = tracers
        self.want_ranks = HOD_params.get('want_ranks', False)
        self.want_AB = HOD_params.get('want_AB', False)
        self.want_shear = HOD_params.get('want_shear', False)
        self.want_expvel = HOD_params.get('want_expvel', False)
        self.want_rsd = HOD_params['want_rsd']
        if clustering_params is not None:
            self.pimax = clustering_params.get('pimax', None)
            self.pi_bin_size = clustering_params.get('pi_bin_size', None)
            bin_params = clustering_params['bin_params']
            self.rpbins = np.logspace(bin_params['logmin'], bin_params['logmax'], bin_params['nbins'] + 1)
            self.clustering_type = clustering_params.get('clustering_type', None)

    @classmethod
    def from_arrays(cls, halo_data, particle_data, params, HOD_params, clustering_params=None, mock_dir='./',
                    z_type='primary'):
        """Build the object from the dictionaries `staging()` would produce (hod/abacus_hod.py:659-704)."""
        self = cls.__new__(cls)
        self.logger = logging.getLogger('AbacusHOD')
        self.sim_name = self.sim_dir = self.subsample_dir = None
        self.z_mock = params['z']
        self.output_dir = str(mock_dir)
        self.halo_lc = params.get('origin', None) is not None
        self.force_mt = False
        self.local_env = {}
        self.z_type = 'lightcone' if self.halo_lc else z_type
        self._init_hod(HOD_params, clustering_params)
        self.chunk, self.n_chunks = params.get('chunk', -1), 1
        self.halo_data, self.particle_data, self.params = halo_data, particle_data, params
        self.mock_dir = Path(mock_dir)
        self.lbox = params['Lbox']
        self._staged = None
        if self.want_AB:
            assert 'hfenv' in self.halo_data.keys()
            assert 'hdeltac' in self.halo_data.keys()
        if self.want_shear:
            assert 'hshear' in self.halo_data.keys()
        self._build_mass_function()
        return self

    @classmethod
    def from_prepared(cls, halo_tables, particle_tables, header, z_mock, HOD_params, clustering_params=None, env=None,
                      mock_dir='./', halo_lc=False):
        """Build the object straight from the subsample tables of prepare_sim (`prepare_slab_arrays` of
        abacusutils_amd/hod/prepare_sim.py, one pair per slab) - what `staging()` reads back from the HDF5 files, without the
        files.  header: BoxSize, ParticleMassHMsun, H0, VelZSpace_to_kms (+ LightConeOrigins) of the simulation; env: the
        concatenated (id, mass, Menv) of `slab_environment` over ALL slabs of the box (needed with want_AB, :595-657)."""
        self = cls.__new__(cls)
        self.logger = logging.getLogger('AbacusHOD')
        self.sim_name = self.sim_dir = self.subsample_dir = None
        self.z_mock = z_mock
        self.output_dir = str(mock_dir)
        self.halo_lc = bool(halo_lc)
        self.force_mt = False
        self.local_env = {}
        self.z_type = 'lightcone' if self.halo_lc else 'primary'
        self._init_hod(HOD_params, clustering_params)
        self.chunk, self.n_chunks = -1, 1
        params = {'z': z_mock, 'h': header['H0'] / 100.0, 'Lbox': header['BoxSize'], 'Mpart': header['ParticleMassHMsun'],
                  'velz2kms': header['VelZSpace_to_kms'] / header['BoxSize'], 'chunk': -1, 'numslabs': len(halo_tables),
                  'origin': np.array(header['LightConeOrigins']).reshape(-1, 3)[0] if self.halo_lc else None}
        if self.want_AB and not self.halo_lc and env is None:
            raise ValueError('want_AB needs the environment masses of the whole box (prepare_sim.slab_environment per slab)')
        self.halo_data, self.particle_data = self._assemble_staged(list(halo_tables), list(particle_tables), env, params, True)
        self.params = params
        self.mock_dir = Path(mock_dir)
        self.lbox = params['Lbox']
        self._staged = None
        self._build_mass_function()
        return self

    def _build_mass_function(self):
        """weighted halo histograms used by compute_ngal (:199-251)"""
        hd = self.halo_data
        n = len(hd['hmass'])
        self.logMbins = np.linspace(np.log10(np.min(hd['hmass'])), np.log10(np.max(hd['hmass'])), 101)
        self.deltacbins = np.linspace(-0.5, 0.5, 101)
        self.fenvbins = np.linspace(-0.5, 0.5, 101)
        self.shearbins = np.linspace(-0.5, 0.5, 101)
        cols = (np.log10(hd['hmass']), hd.get('hdeltac', np.zeros(n)), hd.get('hfenv', np.zeros(n)))
        self.halo_mass_func, _ = np.histogramdd(np.vstack(cols).T, bins=[self.logMbins, self.deltacbins, self.fenvbins],
                                                weights=hd['hmultis'])
        self._mass_func_wshear = None  # 100^4 histogram, built on first ELG compute_ngal

    @property
    def halo_mass_func_wshear(self):
        if self._mass_func_wshear is None:
            hd = self.halo_data
            n = len(hd['hmass'])
            cols = (np.log10(hd['hmass']), hd.get('hdeltac', np.zeros(n)), hd.get('hfenv', np.zeros(n)),
                    hd.get('hshear', np.zeros(n)))
            self._mass_func_wshear, _ = np.histogramdd(
                np.vstack(cols).T, bins=[self.logMbins, self.deltacbins, self.fenvbins, self.shearbins],
                weights=hd['hmultis'])
        return self._mass_func_wshear

    # ------------------------------------------------------------------------------------------------------
    def staging(self):
        """Load the halo+particle subsamples (hod/abacus_hod.py:253-704) into the float64 dictionaries the
        population kernels consume."""
        try:
            import h5py
        except ImportError as e:
            raise ImportError('AbacusHOD.staging() reads the prepare_sim HDF5 subsamples and needs h5py; '
                              'use AbacusHOD.from_arrays for in-memory data') from e
        output_dir = Path(self.output_dir)
        simname = Path(self.sim_name)
        sim_dir = Path(self.sim_dir)
        mock_dir = output_dir / simname / ('z%4.3f' % self.z_mock)
        subsample_dir = Path(self.subsample_dir) / simname / ('z%4.3f' % self.z_mock)
        if not (sim_dir / simname).exists():
            raise FileNotFoundError(f'Simulation directory {sim_dir / simname} not found.')
        if not subsample_dir.exists():
            raise FileNotFoundError(f'Subsample directory {subsample_dir} not found.')
        if self.halo_lc:
            halo_info_fns = [str(sim_dir / simname / ('z%4.3f' % self.z_mock) / 'lc_halo_info.asdf')]
        else:
            halo_info_fns = list((sim_dir / simname / 'halos' / ('z%4.3f' % self.z_mock) / 'halo_info').glob('*.asdf'))
        header = _read_asdf_header(halo_info_fns[0])

        params = {}
        params['z'] = self.z_mock
        params['h'] = header['H0'] / 100.0
        params['Lbox'] = header['BoxSize']
        params['Mpart'] = header['ParticleMassHMsun']
        params['velz2kms'] = header['VelZSpace_to_kms'] / params['Lbox']
        if self.halo_lc:
            params['origin'] = np.array(header['LightConeOrigins']).reshape(-1, 3)[0]
        else:
            params['origin'] = None
        n_chunks = self.n_chunks
        params['chunk'] = self.chunk
        chunk = 0 if self.chunk == -1 else self.chunk
        n_jump = int(np.ceil(len(halo_info_fns) / n_chunks))
        start = chunk * n_jump
        end = min((chunk + 1) * n_jump, len(halo_info_fns))
        params['numslabs'] = end - start
        self.lbox = header['BoxSize']

        def fnames(eslab):
            if ('ELG' not in self.tracers) and ('QSO' not in self.tracers) and (not self.force_mt):
                h = subsample_dir / ('halos_xcom_%d_seed600_abacushod_oldfenv' % eslab)
                p = subsample_dir / ('particles_xcom_%d_seed600_abacushod_oldfenv' % eslab)
            else:
                h = subsample_dir / ('halos_xcom_%d_seed600_abacushod_oldfenv_MT' % eslab)
                p = subsample_dir / ('particles_xcom_%d_seed600_abacushod_oldfenv_MT' % eslab)
            if self.want_ranks:
                p = str(p) + '_withranks'
            return str(h) + '_new.h5', str(p) + '_new.h5'

        with_parts = self.z_type in ('primary', 'lightcone')
        H, P = [], []
        for eslab in range(start, end):
            self.logger.info(f'Loading simulation slab {eslab}')
            hfn, pfn = fnames(eslab)
            with h5py.File(hfn, 'r') as f:
                H.append(f['halos'][:])
            if with_parts:
                with h5py.File(pfn, 'r') as f:
                    P.append(f['particles'][:])
        env = None
        if self.want_AB and (not self.halo_lc):
            ids, masses, menvs = [], [], []
            for eslab in range(len(halo_info_fns)):
                envfilename = subsample_dir / f'env_xcom_{eslab}_abacushod_localenv_new.h5'
                if not envfilename.exists():
                    raise FileNotFoundError(f'Missing env sidecar: {envfilename}')
                with h5py.File(envfilename, 'r') as fenv:
                    ids.append(fenv['id'][:].astype(np.int64))
                    masses.append(fenv['mass'][:])
                    menvs.append(fenv['Menv'][:])
            env = (np.concatenate(ids), np.concatenate(masses), np.concatenate(menvs))
        halo_data, particle_data = self._assemble_staged(H, P, env, params, with_parts)
        return halo_data, particle_data, params, mock_dir

    def _assemble_staged(self, H, P, env, params, with_parts):
        """the arrays of `halo_data` / `particle_data` from the subsample tables of the loaded slabs (:421-704).  H / P: one
        table per slab (HDF5 compound arrays, or dicts of columns as prepare_sim.prepare_slab_arrays returns them); env: the
        concatenated (id, mass, Menv) of the env sidecars of ALL slabs of the box, or None"""
        Hc = _concat_tables(H)
        f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        hpos, hvel = f8(Hc['x_L2com']), f8(Hc['v_L2com'])
        hmass = f8(Hc['N'] * params['Mpart'])
        hid = Hc['id'].astype(int)
        hmultis, hrandoms = f8(Hc['multi_halos']), f8(Hc['randoms'])
        vdev = Hc['randoms_exp'] if self.want_expvel else Hc['randoms_gaus_vrms']
        if vdev.ndim == 1:
            self.logger.warning('Warning: galaxy x, y velocity bias randoms not set, using z randoms instead. '
                                'x, y velocities may be unreliable.')
            vdev = np.concatenate((vdev, vdev, vdev)).reshape(-1, 3)
        hveldev = f8(vdev)
        hsigma3d = f8(Hc['sigmav3d_L2com'])
        hc = f8(Hc['r98_L2com'] / Hc['r25_L2com'])
        hrvir = f8(Hc['r98_L2com'])
        hdeltac = f8(Hc['deltac_rank']) if self.want_AB else None
        hfenv = f8(Hc['fenv_rank']) if self.want_AB else None
        hshear = f8(Hc['shear_rank']) if self.want_shear else None

        particle_data = {}
        if with_parts:
            Pc = _concat_tables(P)
            fields = _table_fields(Pc)
            ppos, pvel, phvel = f8(Pc['pos']), f8(Pc['vel']), f8(Pc['halo_vel'])
            phmass = f8(Pc['halo_mass'])
            phid = Pc['halo_id'].astype(int)
            pNp, psub, prandoms = f8(Pc['Np']), f8(Pc['downsample_halo']), f8(Pc['randoms'])
            pdeltac = f8(Pc['halo_deltac']) if self.want_AB else None
            pfenv = f8(Pc['halo_fenv']) if self.want_AB else None
            pshear = f8(Pc['halo_shear']) if self.want_shear else None
            ranks = {}
            if self.want_ranks:
                assert 'ranks' in fields and 'ranksv' in fields
                for src, dst in (('ranks', 'pranks'), ('ranksv', 'pranksv'), ('ranksp', 'pranksp'),
                                 ('ranksr', 'pranksr'), ('ranksc', 'pranksc')):
                    ranks[dst] = f8(Pc[src]) if src in fields else np.zeros(len(Pc))
        # sort halos by id, important for conformity (:566-585)
        if not np.all(hid[:-1] <= hid[1:]):
            self.logger.info('Sorting halos for conformity calculation.')
            s = _argsort_ids(hid)
            hpos, hvel, hmass, hid, hmultis, hrandoms, hveldev, hsigma3d, hc, hrvir = (
                a[s] for a in (hpos, hvel, hmass, hid, hmultis, hrandoms, hveldev, hsigma3d, hc, hrvir))
            if self.want_AB:
                hdeltac, hfenv = hdeltac[s], hfenv[s]
            if self.want_shear:
                hshear = hshear[s]
        assert np.all(hid[:-1] <= hid[1:])
        if with_parts:
            pweights = 1 / pNp / psub
            pinds = _searchsorted(hid, phid)

        # global environment ranking from the env sidecars (:595-657)
        if self.want_AB and (not self.halo_lc):
            mcut_env = self.local_env.get('mcut', 1e11)
            nbins_env = self.local_env.get('nbins', 100)
            env_id, env_mass, env_Menv = env
            mbins_env = np.logspace(np.log10(mcut_env), 15.5, nbins_env + 1)
            hfenv_full = calc_fenv_opt(env_Menv, mbins_env, env_mass)
            env_sort = _argsort_ids(env_id)
            env_id, hfenv_full = env_id[env_sort], hfenv_full[env_sort]
            hmatch = _searchsorted(env_id, hid)
            if not np.all(env_id[hmatch] == hid):
                raise RuntimeError('Failed to map global env sidecars onto staged halos by halo ID.')
            hfenv = hfenv_full[hmatch]
            if with_parts:
                if not np.all(hid[pinds] == phid):
                    raise RuntimeError('Particle-to-halo mapping pinds is inconsistent with phid.')
                pfenv = hfenv[pinds]

        halo_data = {'hpos': hpos, 'hvel': hvel, 'hmass': hmass, 'hid': hid, 'hmultis': hmultis,
                     'hrandoms': hrandoms, 'hveldev': hveldev, 'hsigma3d': hsigma3d, 'hc': hc, 'hrvir': hrvir}
        if with_parts:
            particle_data = {'ppos': ppos, 'pvel': pvel, 'phvel': phvel, 'phmass': phmass, 'phid': phid,
                             'pweights': pweights, 'prandoms': prandoms, 'pinds': pinds}
            npart = len(phmass)
            if self.want_ranks:
                particle_data.update(ranks)
            else:
                for k in ('pranks', 'pranksv', 'pranksp', 'pranksr', 'pranksc'):
                    particle_data[k] = np.ones(npart)
        if self.want_AB:
            halo_data['hdeltac'], halo_data['hfenv'] = hdeltac, hfenv
            if with_parts:
                particle_data['pdeltac'], particle_data['pfenv'] = pdeltac, pfenv
        if self.want_shear:
            halo_data['hshear'] = hshear
            if with_parts:
                particle_data['pshear'] = pshear
        return halo_data, particle_data

    # ------------------------------------------------------------------------------------------------------
    def restage(self):
        """Drop the device copy of the subsample (call after modifying halo_data / particle_data in place)."""
        if self._staged is not None:
            self._staged.free()
            self._staged = None

    def _device_catalog(self):
        if self._staged is None:
            self._staged = StagedCatalog(self.halo_data, self.particle_data)
        return self._staged

    def run_hod(self, tracers=None, want_rsd=True, want_nfw=False, NFW_draw=None, reseed=None, write_to_disk=False,
                Nthread=16, verbose=False, fn_ext=None):
        """Runs a custom HOD; returns `mock_dict` {tracer: {'x','y','z','vx','vy','vz','mass','id','Ncent'}}
        with centrals first (hod/abacus_hod.py:706-859)."""
        if tracers is None:
            tracers = self.tracers
        if self.z_type == 'secondary' and not want_nfw:
            raise RuntimeError('Secondary redshifts do not have particle pos/vel outputs and so only NFW profiles '
                               'are supported')
        if reseed:
            start = time.time()
            # The reference draws float32 streams from parallel_numpy_rng.MTGenerator(PCG64(reseed)) (:778-823), a
            # third-party generator that is not available here (stream parity unpinned; the reference only smoke-tests
            # this path).  Same distributions and dtypes from the device's counter-based Philox generator: the three
            # arrays are rewritten in HBM, then copied back so that `halo_data` / `particle_data` are mutated exactly
            # as the reference mutates them (:824-835).  `reseed_sync_host = False` skips the copy (MCMC loops).
            st = self._device_catalog()
            st.reseed(reseed, hsigma3d=self.halo_data['hsigma3d'], want_expvel=self.want_expvel)
            if getattr(self, 'reseed_sync_host', True):
                self.halo_data['hrandoms'] = st.fetch_field('hrandoms').astype(np.float32)    # float32 draws (:780)
                self.halo_data['hveldev'] = st.fetch_field('hveldev')                         # float32 * float64
                self.particle_data['prandoms'] = st.fetch_field('prandoms').astype(np.float32)
            self.logger.info(f'Randoms generated in elapsed time {time.time() - start:.2f} s.')

        start = time.time()
        mock_dict = gen_gal_cat(self.halo_data, self.particle_data, tracers, self.params, Nthread,
                                enable_ranks=self.want_ranks, rsd=want_rsd, nfw=want_nfw, NFW_draw=NFW_draw,
                                write_to_disk=write_to_disk, savedir=self.mock_dir, verbose=verbose, fn_ext=fn_ext,
                                staged=self._device_catalog(), lazy=getattr(self, 'lazy_columns', False))
        self.logger.info(f'HOD generated in elapsed time {time.time() - start:.2f} s.')
        return mock_dict

    # ------------------------------------------------------------------------------------------------------
    def _ngal_cells(self):
        """np.histogramdd's cell of every halo in the (logM, deltac, fenv, shear) histogram (:199-251): uint8 (N, 4),
        255 = outside the range (histogramdd drops those), plus the [4][100] table of cell centres"""
        hd = self.halo_data
        n = len(hd['hmass'])
        cols = (np.log10(hd['hmass']), hd.get('hdeltac', np.zeros(n)), hd.get('hfenv', np.zeros(n)),
                hd.get('hshear', np.zeros(n)))
        edges = (self.logMbins, self.deltacbins, self.fenvbins, self.shearbins)
        bins = np.empty((n, 4), dtype=np.uint8)
        for d, (x, e) in enumerate(zip(cols, edges)):
            idx = np.searchsorted(e, x, side='right') - 1
            idx[x == e[-1]] = len(e) - 2           # the last bin is closed on the right
            idx[(idx < 0) | (idx > len(e) - 2)] = 255
            bins[:, d] = idx
        centres = np.empty((4, len(self.logMbins) - 1))
        centres[0] = 10 ** (0.5 * (self.logMbins[1:] + self.logMbins[:-1]))   # Mh_temp = 10**logMs[i] (:1009)
        for d, e in enumerate(edges[1:], start=1):
            centres[d] = 0.5 * (e[1:] + e[:-1])
        return bins, centres

    def compute_ngal(self, tracers=None, Nthread=16):
        """Expected number of each tracer and its satellite fraction from the weighted halo histogram
        (hod/abacus_hod.py:861-1179), evaluated on the device as the identical sum over halos (abacus_hod_ngal)."""
        import ctypes as C

        from .. import _lib
        from .GRAND_HOD import TRACERS, marshal_params
        if tracers is None:
            tracers = self.tracers
        st = self._device_catalog()
        if not getattr(st, '_ngal_set', False):
            bins, centres = self._ngal_cells()
            _lib.check(_lib.lib().abacus_hod_set_ngal_bins(st._h, _lib.ptr(np.ascontiguousarray(bins)),
                                                           _lib.ptr(np.ascontiguousarray(centres)), centres.shape[1]))
            st._ngal_set = True
        known = {t: dict(tracers[t]) for t in tracers if t in TRACERS}
        for hod in known.values():               # keys gen_gals requires but compute_ngal never reads (:884-976)
            for k in ('alpha_c', 'alpha_s', 's', 's_v', 's_p', 's_r'):
                hod.setdefault(k, 0.0)
        if 'ELG' in known:
            # compute_ngal's own defaults (:925-948), which differ from gen_gals': `A_s` is optional here, and the
            # conformity parameters default to the RAW logM1 / alpha - the z-evolution (:1050-1051) never touches them
            hod = known['ELG']
            hod.setdefault('A_s', 1.0)
            for k, src in (('logM1_EE', 'logM1'), ('alpha_EE', 'alpha'), ('logM1_EL', 'logM1'), ('alpha_EL', 'alpha')):
                hod.setdefault(k, hod[src])
        p = marshal_params(known, dict(self.params, z=self.z_mock), False, True)
        out = (C.c_double * 6)()
        _lib.check(_lib.lib().abacus_hod_ngal(st._h, C.byref(p), out))
        ngal_dict, fsat_dict = {}, {}
        for etracer in known:
            t = TRACERS.index(etracer)
            ngal_cent, ngal_sat = out[t], out[3 + t]
            ngal_dict[etracer] = ngal_cent + ngal_sat
            fsat_dict[etracer] = ngal_sat / (ngal_cent + ngal_sat)
        return ngal_dict, fsat_dict

    # ------------------------------------------------------------------------------------------------------
    def compute_clustering(self, mock_dict, *args, **kwargs):
        """(:1181-1219)"""
        if self.clustering_type == 'xirppi':
            return self.compute_xirppi(mock_dict, *args, **kwargs)
        elif self.clustering_type == 'wp':
            return self.compute_wp(mock_dict, *args, **kwargs)
        elif self.clustering_type == 'multipole':
            return self.compute_multipole(mock_dict, *args, **kwargs)
        raise ValueError('clustering_type not implemented or not specified, use xirppi, wp, multipole')

    @staticmethod
    def _xyz(mock_dict, tracer):
        """the tracer's coordinate columns: still in HBM behind this object's staged catalogue when `mock_dict` is the
        untouched result of the latest run_hod (GRAND_HOD.MockDict.device_xyz), else the host arrays"""
        dev = mock_dict.device_xyz(tracer) if hasattr(mock_dict, 'device_xyz') else None
        if dev is not None:
            return dev
        return mock_dict[tracer]['x'], mock_dict[tracer]['y'], mock_dict[tracer]['z']

    def _pairs(self, mock_dict, fn):
        clustering = {}
        for i1, tr1 in enumerate(mock_dict.keys()):
            x1, y1, z1 = self._xyz(mock_dict, tr1)
            for i2, tr2 in enumerate(mock_dict.keys()):
                if i1 > i2:
                    continue  # cross-correlations are symmetric
                if i1 == i2:
                    clustering[tr1 + '_' + tr2] = fn(x1, y1, z1, None, None, None)
                else:
                    x2, y2, z2 = self._xyz(mock_dict, tr2)
                    if isinstance(x1, np.ndarray) != isinstance(x2, np.ndarray):   # one side only in HBM: both from the host
                        x1, y1, z1 = (mock_dict[tr1][c] for c in 'xyz')
                        x2, y2, z2 = (mock_dict[tr2][c] for c in 'xyz')
                    clustering[tr1 + '_' + tr2] = fn(x1, y1, z1, x2, y2, z2)
                    clustering[tr2 + '_' + tr1] = clustering[tr1 + '_' + tr2]
        return clustering

    def compute_xirppi(self, mock_dict, rpbins, pimax, pi_bin_size, Nthread=8):
        """xi(rp, pi) for every tracer pair (:1221-1279)"""
        return self._pairs(mock_dict, lambda x1, y1, z1, x2, y2, z2: calc_xirppi_fast(
            x1, y1, z1, rpbins, pimax, pi_bin_size, self.lbox, Nthread, x2=x2, y2=y2, z2=z2))

    def compute_wp(self, mock_dict, rpbins, pimax, pi_bin_size, Nthread=8):
        """wp(rp) for every tracer pair (:1826-1885)"""
        return self._pairs(mock_dict, lambda x1, y1, z1, x2, y2, z2: calc_wp_fast(
            x1, y1, z1, rpbins, pimax, self.lbox, Nthread, x2=x2, y2=y2, z2=z2))

    def compute_multipole(self, mock_dict, rpbins, pimax, sbins, nbins_mu, orders=[0, 2], Nthread=8):
        """wp concatenated with xi_l(s) (:1281-1336; like the reference, cross pairs use `rpbins` as s bins, :1313)"""
        def fn(x1, y1, z1, x2, y2, z2):
            sb = sbins if x2 is None else rpbins
            new_multi = calc_multipole_fast(x1, y1, z1, sb, self.lbox, Nthread, nbins_mu=nbins_mu, orders=orders,
                                            x2=x2, y2=y2, z2=z2)
            new_wp = calc_wp_fast(x1, y1, z1, rpbins, pimax, self.lbox, Nthread, x2=x2, y2=y2, z2=z2)
            return np.concatenate((new_wp, new_multi))
        return self._pairs(mock_dict, fn)

    def compute_power(self, mock_dict, nbins_k, nbins_mu, k_hMpc_max, logk, poles=[], paste='TSC', num_cells=550,
                      compensated=False, interlaced=False):
        r"""P(k, mu) and/or P_l(k) for every tracer pair (:1338-1472).  Returned keys: '{a}_{b}', '..._modes',
        '..._ell', '..._ell_modes', 'k_binc', 'mu_binc'."""
        Lbox = self.lbox
        clustering = {}
        # the untouched result of the latest run_hod, unweighted: every tracer's field from its columns in HBM, deposited
        # and transformed once, all pairs binned on the device (analysis.power_spectrum.calc_power_multi)
        dev = {tr: (mock_dict.device_xyz(tr) if hasattr(mock_dict, 'device_xyz') and 'w' not in mock_dict[tr] else None)
               for tr in mock_dict.keys()}
        if dev and all(v is not None for v in dev.values()) and len(dev) <= 8:
            from ..analysis.power_spectrum import calc_power_multi
            tabs = calc_power_multi(dev, Lbox, nbins_k, nbins_mu, k_hMpc_max, logk, paste, num_cells, compensated,
                                    interlaced, poles=poles)
            for (tr1, tr2), power in tabs.items():
                key = tr1 + '_' + tr2
                clustering[key] = power['power']
                clustering[key + '_modes'] = power['N_mode']
                clustering[key + '_ell'] = power['poles']
                clustering[key + '_ell_modes'] = power['N_mode_poles']
                if tr1 != tr2:
                    for suffix in ('', '_modes', '_ell', '_ell_modes'):
                        clustering[tr2 + '_' + tr1 + suffix] = clustering[key + suffix]
            clustering['k_binc'] = power['k_mid']
            clustering['mu_binc'] = power['mu_mid'][0]
            return clustering
        for i1, tr1 in enumerate(mock_dict.keys()):
            pos1 = np.stack((mock_dict[tr1]['x'], mock_dict[tr1]['y'], mock_dict[tr1]['z']), axis=1)
            w1 = mock_dict[tr1].get('w', None)
            for i2, tr2 in enumerate(mock_dict.keys()):
                if i1 > i2:
                    continue
                if i1 == i2:
                    power = calc_power(pos1, Lbox, nbins_k, nbins_mu, k_hMpc_max, logk, paste, num_cells, compensated,
                                       interlaced, w=w1, poles=poles)
                else:
                    pos2 = np.stack((mock_dict[tr2]['x'], mock_dict[tr2]['y'], mock_dict[tr2]['z']), axis=1)
                    w2 = mock_dict[tr2].get('w', None)
                    power = calc_power(pos1, Lbox, nbins_k, nbins_mu, k_hMpc_max, logk, paste, num_cells, compensated,
                                       interlaced, w=w1, pos2=pos2, w2=w2, poles=poles)
                key = tr1 + '_' + tr2
                clustering[key] = power['power']
                clustering[key + '_modes'] = power['N_mode']
                clustering[key + '_ell'] = power['poles']  # KeyError with poles=[] exactly like the reference (:1431)
                clustering[key + '_ell_modes'] = power['N_mode_poles']
                if i1 != i2:
                    rkey = tr2 + '_' + tr1
                    for suffix in ('', '_modes', '_ell', '_ell_modes'):
                        clustering[rkey + suffix] = clustering[key + suffix]
        clustering['k_binc'] = power['k_mid']
        clustering['mu_binc'] = power['mu_mid'][0]
        return clustering

    def apply_zcv(self, mock_dict, config, load_presaved=False):
        """signature of hod/abacus_hod.py:1474; Zel'dovich control variates (hod/zcv) are outside the MI355X hot-path scope"""
        raise NotImplementedError('Zel\'dovich control variates (hod/zcv) are outside the MI355X hot-path scope')

    def apply_zcv_xi(self, mock_dict, config, load_presaved=False):
        """signature of hod/abacus_hod.py:1663; see apply_zcv"""
        raise NotImplementedError('Zel\'dovich control variates (hod/zcv) are outside the MI355X hot-path scope')

    def gal_reader(self, output_dir=None, simname=None, sim_dir=None, z_mock=None, want_rsd=None, tracers=None):
        """Load `{tracer}s.dat` ECSV catalogs written by run_hod(write_to_disk=True) (:1887-1950)."""
        if want_rsd is None:
            want_rsd = self.want_rsd
        if tracers is None:
            tracers = self.tracers.keys()
        outdir = Path(self.mock_dir) / ('galaxies' + ('_rsd' if want_rsd else ''))
        mockdict = {}
        for tracer in tracers:
            header, rows, meta = None, [], {}
            for line in open(outdir / (tracer + 's.dat')):
                if line.startswith('#'):
                    if 'Ncent:' in line:
                        meta['Ncent'] = int(line.split('Ncent:')[1].strip(' }\n'))
                    continue
                if header is None:
                    header = line.split()
                    continue
                rows.append(line.split())
            cols = {}
            for j, name in enumerate(header):
                conv = int if name == 'id' else float
                cols[name] = np.array([conv(r[j]) for r in rows], dtype=np.int64 if name == 'id' else np.float64)
            cols.update(meta)
            mockdict[tracer] = cols
        return mockdict
