#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE.  Needs /root/reference (absent on the GPU box); only its
outputs - small data fixtures - are committed.  The reference's numba-decorated
functions run as plain Python through the identity shim in oracle/shim/
(numba is not importable here).

What is captured
  hod_mini.npz / hod_lc.npz   the reference's own test fixtures
        (tests/ref_hod/**: HDF5 subsamples -> staging()-layout arrays,
        galaxies_rsd/*.dat ECSV -> expected columns), plus a check that the
        shimmed reference reproduces the ECSV exactly.
  hod_synth_*.npz             gen_gal_cat / gen_cent outputs of the reference on
        seeded synthetic halos (abacusutils_amd.synth) for the branches its
        tests do not pin (QSO, ranks, AB, shear, conformity, light cone, ...).
  tsc_ref.npz                 tests/ref_tsc/*.asdf grids (blosc-decoded), sparse.
  tsc_cases.npz               _tsc_scatter / tsc_parallel / partition_parallel
        outputs for dtype / weights / offset / anisotropic cases (2-D grids raise in
        the reference itself: 3 indices on a 2-D array, tsc.py:471).
  power_cases.npz             calc_power for all paste x compensated x interlaced
        modes, cross spectra, poles, logk; bin_kmu / calc_pk_from_deltak.

  power_exports.npz           bin_kmu, get_raw_power, shift_field_fft, get_interlaced_field_fft on seeded inputs
  power_f64.npz               get_field / get_field_fft / calc_power with dtype=np.float64
  power_helpers.npz           bin_kppi, project_3d_to_poles, pk_to_xi, expand_poles_to_3d, get_smoothing,
        get_delta_mu2 on seeded 16^3 / 21^3 inputs.

  ngal.npz                    AbacusHOD.compute_ngal (abacus_hod.py:861-1179) on a 12-cell-per-dimension histogram of
        seeded synthetic halos, three parameter cases (defaults, assembly bias + z-evolution, evolving ELG with the
        conformity defaults).
  pair_wrappers.npz           calc_xirppi_fast / calc_wp_fast / calc_multipole_fast / tpcf_multipole with a brute-force
        stand-in for Corrfunc's counters (pins the wrapper arithmetic, not Corrfunc).

  prepare_sim.npz             prepare_sim.prepare_slab (hod/prepare_sim.py:296-1052) on seeded synthetic slabs (periodic box and
                              halo light cones) with stand-ins
        for its file layer (CompaSOHaloCatalog -> synth tables, h5py -> capture): three configurations (MT / LRG-only,
        ranks, assembly bias with the padded Menv, shear).

usage: python oracle/make_golden.py [hod] [tsc] [power] [helpers] [catalog] [sweep] [ngal] [pairs] [prepare]
"""
import ctypes
import os
import re
import struct
import subprocess
import sys
import tempfile
import types
import warnings
from pathlib import Path

import numpy as np

# the goldens record the reference's behaviour under NumPy >= 2 scalar promotion (NEP 50): prepare_sim's perihelion
# iteration and a few float32 expressions of the spectrum code promote differently under NumPy 1.x (csrc/prepare.hip: prep_ranks)
assert int(np.__version__.split('.')[0]) >= 2, 'golden vectors are generated under NumPy >= 2'

REPO = Path(__file__).resolve().parent.parent
REF = Path('/root/reference')
GOLD = REPO / 'tests' / 'golden'
H5PY_PYTHON = '/opt/conda/bin/python3.9'
LIBBLOSC = '/opt/conda/lib/libblosc.so.1'

sys.path.insert(0, str(REPO / 'oracle' / 'shim'))
sys.path.insert(0, str(REPO))


def import_reference():
    """abacusnbody/__init__.py imports a generated version.py that is not in the
    tree -> register a bare package object pointing at the reference directory."""
    pkg = types.ModuleType('abacusnbody')
    pkg.__path__ = [str(REF / 'abacusnbody')]
    sys.modules['abacusnbody'] = pkg
    import abacusnbody.hod.GRAND_HOD as G
    import abacusnbody.analysis.tsc as T
    import abacusnbody.analysis.power_spectrum as P
    import abacusnbody.analysis.cic as C
    return G, T, P, C


# ----------------------------------------------------------------------------
# fixture readers
# ----------------------------------------------------------------------------
def read_h5(fn, dset):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'a.npy')
        subprocess.check_call([H5PY_PYTHON, str(REPO / 'oracle' / '_h5dump.py'), str(fn), dset, out])
        return np.load(out)


def read_ecsv(fn):
    """ECSV -> dict of float64 columns + int64 id, and the Ncent meta value."""
    ncent = None
    rows = []
    header = None
    for line in open(fn):
        if line.startswith('#'):
            m = re.search(r'Ncent: (\d+)', line)
            if m:
                ncent = int(m.group(1))
            continue
        if header is None:
            header = line.split()
            continue
        rows.append(line.split())
    cols = {}
    for j, name in enumerate(header):
        if name == 'id':
            cols[name] = np.array([int(r[j]) for r in rows], dtype=np.int64)
        else:
            cols[name] = np.array([float(r[j]) for r in rows], dtype=np.float64)
    return cols, ncent


_ASDF_DT = {'float32': 'f4', 'float64': 'f8', 'int32': 'i4', 'uint64': 'u8', 'int64': 'i8', 'uint32': 'u4',
            'int16': 'i2', 'int8': 'i1', 'uint8': 'u1', 'bool8': 'u1'}


def _asdf_blocks(raw):
    """decompressed bytes of every binary block of an ASDF file written with the reference's 'blsc' extension
    (abacusnbody/data/asdf.py:81-93,128-181): block header, then chunks of [!I nbytes][blosc frame]; plain blocks too"""
    lib = ctypes.CDLL(LIBBLOSC)
    blocks = []
    p = raw.find(b'\xd3BLK')
    while p >= 0 and raw[p:p + 4] == b'\xd3BLK':
        p += 4
        (hsize,) = struct.unpack('>H', raw[p:p + 2])
        flags, comp, alloc, used, dsize = struct.unpack('>I4sQQQ', raw[p + 2:p + 2 + 32])
        p = p + 2 + hsize
        end = p + used
        if comp == b'blsc':
            out = bytearray()
            q = p
            while q < end:
                (n,) = struct.unpack('!I', raw[q:q + 4])
                q += 4
                frame = raw[q:q + n]
                q += n
                nbytes, cbytes, bs = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
                lib.blosc_cbuffer_sizes(frame, ctypes.byref(nbytes), ctypes.byref(cbytes), ctypes.byref(bs))
                buf = ctypes.create_string_buffer(nbytes.value)
                r = lib.blosc_decompress(frame, buf, ctypes.c_size_t(nbytes.value))
                assert r == nbytes.value
                out += buf.raw
            assert len(out) == dsize
            blocks.append(bytes(out))
        else:
            assert comp == b'\0\0\0\0', comp
            blocks.append(raw[p:p + dsize])
        p = p + alloc
    return blocks


def read_asdf_arrays(fn):
    """{name: ndarray} of every ndarray in the YAML tree of an ASDF file (key name, or the column `name:` of a table)"""
    raw = open(fn, 'rb').read()
    tree = raw[: raw.index(b'\xd3BLK')].decode('latin1')
    blocks = _asdf_blocks(raw)
    out = {}
    for m in re.finditer(r'(\w+): !core/ndarray-[\d.]+\s*\n((?:[ \t]+\w+: .*\n)+)', tree):
        body = m.group(2)
        src = int(re.search(r'source: (\d+)', body).group(1))
        shape = [int(x) for x in re.search(r'shape: \[([\d, ]+)\]', body).group(1).split(',')]
        dt = re.search(r'datatype: (\w+)', body).group(1)
        bo = re.search(r'byteorder: (\w+)', body).group(1)
        nm = re.search(r'\n\s+name: (\w+)', '\n' + body) or re.search(r'^\s+name: (\w+)', tree[m.end():m.end() + 80])
        name = m.group(1) if m.group(1) != 'data' or nm is None else nm.group(1)
        dtype = np.dtype(_ASDF_DT[dt]).newbyteorder('<' if bo == 'little' else '>')
        off = re.search(r'offset: (\d+)', body)
        off = int(off.group(1)) if off else 0
        cnt = int(np.prod(shape))      # a view of a larger base array is stored as the base block + offset
        assert 'strides' not in body
        out[name] = np.frombuffer(blocks[src], dtype=dtype, count=cnt, offset=off).reshape(shape).astype(dtype.newbyteorder('='))
    return out


def read_asdf_blsc(fn, key):
    """one named array of an ASDF file (see read_asdf_arrays)"""
    return read_asdf_arrays(fn)[key]


# ----------------------------------------------------------------------------
# HOD
# ----------------------------------------------------------------------------
def stage_from_h5(dirn, nslab, Mpart, Lbox, velzspace_to_kms, origin, z):
    """Builds halo_data / particle_data / params the way staging() does
    (hod/abacus_hod.py:289-298,355-383,423-552,566-588,659-702) for
    want_AB=True, want_shear=False, want_ranks=False, MT files."""
    halos = [read_h5(dirn / f'halos_xcom_{i}_seed600_abacushod_oldfenv_MT_new.h5', 'halos') for i in range(nslab)]
    parts = [read_h5(dirn / f'particles_xcom_{i}_seed600_abacushod_oldfenv_MT_new.h5', 'particles') for i in range(nslab)]
    H = np.concatenate(halos)
    Pt = np.concatenate(parts)
    nh, npart = len(H), len(Pt)
    hpos = np.empty((nh, 3)); hpos[:] = H['x_L2com']
    hvel = np.empty((nh, 3)); hvel[:] = H['v_L2com']
    hmass = np.empty(nh); hmass[:] = H['N'] * Mpart
    hid = np.empty(nh, dtype=int); hid[:] = H['id'].astype(int)
    hmultis = np.empty(nh); hmultis[:] = H['multi_halos']
    hrandoms = np.empty(nh); hrandoms[:] = H['randoms']
    hveldev = np.empty((nh, 3)); hveldev[:] = H['randoms_gaus_vrms']
    hsigma3d = np.empty(nh); hsigma3d[:] = H['sigmav3d_L2com']
    hc = np.empty(nh); hc[:] = H['r98_L2com'] / H['r25_L2com']
    hrvir = np.empty(nh); hrvir[:] = H['r98_L2com']
    hdeltac = np.empty(nh); hdeltac[:] = H['deltac_rank']
    hfenv = np.empty(nh); hfenv[:] = H['fenv_rank']
    ppos = np.empty((npart, 3)); ppos[:] = Pt['pos']
    pvel = np.empty((npart, 3)); pvel[:] = Pt['vel']
    phvel = np.empty((npart, 3)); phvel[:] = Pt['halo_vel']
    phmass = np.empty(npart); phmass[:] = Pt['halo_mass']
    phid = np.empty(npart, dtype=int); phid[:] = Pt['halo_id'].astype(int)
    pNp = np.empty(npart); pNp[:] = Pt['Np']
    psub = np.empty(npart); psub[:] = Pt['downsample_halo']
    prandoms = np.empty(npart); prandoms[:] = Pt['randoms']
    pdeltac = np.empty(npart); pdeltac[:] = Pt['halo_deltac']
    pfenv = np.empty(npart); pfenv[:] = Pt['halo_fenv']
    if not np.all(hid[:-1] <= hid[1:]):
        s = np.argsort(hid)
        hpos, hvel, hmass, hid, hmultis, hrandoms, hveldev = (a[s] for a in (hpos, hvel, hmass, hid, hmultis, hrandoms, hveldev))
        hsigma3d, hc, hrvir, hdeltac, hfenv = (a[s] for a in (hsigma3d, hc, hrvir, hdeltac, hfenv))
    pweights = 1 / pNp / psub
    pinds = np.searchsorted(hid, phid).astype(np.int64)
    halo_data = dict(hpos=hpos, hvel=hvel, hmass=hmass, hid=hid, hmultis=hmultis, hrandoms=hrandoms,
                     hveldev=hveldev, hsigma3d=hsigma3d, hc=hc, hrvir=hrvir, hdeltac=hdeltac, hfenv=hfenv)
    particle_data = dict(ppos=ppos, pvel=pvel, phvel=phvel, phmass=phmass, phid=phid, pweights=pweights,
                         prandoms=prandoms, pinds=pinds, pdeltac=pdeltac, pfenv=pfenv)
    for k in ('pranks', 'pranksv', 'pranksp', 'pranksr', 'pranksc'):
        particle_data[k] = np.ones(npart)
    params = dict(z=z, h=0.6736, Lbox=Lbox, Mpart=Mpart, velz2kms=velzspace_to_kms / Lbox,
                  origin=None if origin is None else np.array(origin, dtype=np.float64), chunk=-1, numslabs=nslab)
    return halo_data, particle_data, params


def pack_inputs(halo_data, particle_data, params):
    d = {}
    for k, v in halo_data.items():
        d['h.' + k] = v
    for k, v in particle_data.items():
        d['p.' + k] = v
    for k, v in params.items():
        if v is None:
            continue
        d['params.' + k] = np.asarray(v)
    return d


def pack_mock(prefix, mock):
    d = {}
    for tr, cols in mock.items():
        for k, v in cols.items():
            d[f'{prefix}.{tr}.{k}'] = np.asarray(v)
    return d


def checksum(halo_data, particle_data):
    """order-sensitive float64 checksum of the synthetic inputs (detects a
    change of numpy's Generator streams between build container and GPU box)"""
    s = 0.0
    for d in (halo_data, particle_data):
        for k in sorted(d):
            a = np.asarray(d[k], dtype=np.float64).ravel()
            s += float(np.dot(a, np.cos(np.arange(a.size) * 0.001)))
    return s


def gen_hod(G):
    import yaml
    from abacusutils_amd import synth

    cfg = yaml.safe_load(open(REF / 'tests' / 'abacus_hod.yaml'))
    tracers = {'LRG': cfg['HOD_params']['LRG_params'], 'ELG': cfg['HOD_params']['ELG_params']}

    # --- the reference's own fixtures -----------------------------------
    for name, dirn, nslab, Mpart, Lbox, vz, origin, z in (
        ('hod_mini', REF / 'tests/ref_hod/Mini_N64_L32/z0.000', 3, 1.088239739e10, 32.0, 3200.0, None, 0.0),
        ('hod_lc', REF / 'tests/ref_hod/AbacusSummit_base_c000_ph001-abridged/z2.250', 1,
         2109081520.453063, 2000.0, 208774.9025637363, (-990.0, -990.0, -990.0), 2.25),
    ):
        hd, pd, params = stage_from_h5(dirn, nslab, Mpart, Lbox, vz, origin, z)
        out = pack_inputs(hd, pd, params)
        mock = G.gen_gal_cat(hd, pd, tracers, params, Nthread=4, enable_ranks=False, rsd=True)
        for tr in ('LRG', 'ELG'):
            cols, ncent = read_ecsv(dirn / 'galaxies_rsd' / f'{tr}s.dat')
            assert ncent == mock[tr]['Ncent'], (name, tr, ncent, mock[tr]['Ncent'])
            for k in cols:
                if k == 'id':
                    np.testing.assert_array_equal(cols[k], mock[tr][k])
                else:
                    np.testing.assert_allclose(cols[k], mock[tr][k], rtol=1e-14, atol=0)
                out[f'expect.{tr}.{k}'] = cols[k]
            out[f'expect.{tr}.Ncent'] = np.int64(ncent)
            print(name, tr, 'N', len(cols['x']), 'Ncent', ncent, 'shimmed reference == ECSV fixture')
        # also keep what the shimmed reference returned (bitwise target for the oracle)
        out.update(pack_mock('shim', mock))
        np.savez_compressed(GOLD / f'{name}.npz', **out)

    # --- synthetic cases: branches the reference tests do not pin --------
    rich = {
        'LRG': dict(synth.LRG_PARAMS, alpha_c=0.3, alpha_s=0.8, Acent=0.1, Asat=-0.2, Bcent=-0.05, Bsat=0.15,
                    s=0.1, s_v=-0.1, s_p=0.05, s_r=0.2, logM_cut=12.6, logM1=13.6, z_pivot=0.8,
                    logM_cut_pr=0.2, logM1_pr=-0.1),
        'ELG': dict(synth.ELG_PARAMS, alpha_c=0.2, alpha_s=1.1, Acent=0.1, Asat=0.1, Bcent=0.05, Bsat=-0.1,
                    Ccent=0.07, Csat=-0.04, logM1_EE=13.0, alpha_EE=0.9, logM1_EL=13.2, alpha_EL=1.1,
                    s=0.2, s_v=0.1, s_p=-0.1, s_r=0.0, logM1=13.0),
        'QSO': dict(synth.QSO_PARAMS, alpha_c=0.5, alpha_s=0.9, Acent=-0.1, Asat=0.2, Bcent=0.1, Bsat=0.0,
                    s=-0.1, s_v=0.0, s_p=0.1, s_r=0.1, logM1=13.0),
    }
    cases = [
        # name, tracers, ranks, rsd, origin, nh, np
        ('lrg', {'LRG': dict(synth.LRG_PARAMS, logM_cut=12.6, logM1=13.6)}, False, True, None),
        ('all_rich', rich, False, True, None),
        ('all_rich_ranks', rich, True, True, None),
        ('all_rich_norsd', rich, False, False, None),
        ('all_rich_lc', rich, True, True, (-990.0, -990.0, -990.0)),
        ('elg_only', {'ELG': rich['ELG']}, False, True, None),
        ('qso_only', {'QSO': rich['QSO']}, False, True, None),
        ('lrg_qso', {'LRG': rich['LRG'], 'QSO': rich['QSO']}, True, True, None),
    ]
    NH, NP, SEED = 20000, 30000, 600
    for name, trs, ranks, rsd, origin in cases:
        hd, pd, params = synth.synth_hod_inputs(NH, NP, seed=SEED, with_ranks=ranks, origin=origin)
        mock = G.gen_gal_cat(hd, pd, trs, params, Nthread=3, enable_ranks=ranks, rsd=rsd)
        out = pack_mock('expect', mock)
        out['meta.nh'] = np.int64(NH)
        out['meta.np'] = np.int64(NP)
        out['meta.seed'] = np.int64(SEED)
        out['meta.ranks'] = np.bool_(ranks)
        out['meta.rsd'] = np.bool_(rsd)
        out['meta.checksum'] = np.float64(checksum(hd, pd))
        if origin is not None:
            out['meta.origin'] = np.array(origin)
        import json
        out['meta.tracers'] = np.array(json.dumps(trs))
        print('hod_synth_' + name, {t: (len(m['x']), m['Ncent']) for t, m in mock.items()})
        np.savez_compressed(GOLD / f'hod_synth_{name}.npz', **out)


# ----------------------------------------------------------------------------
# TSC
# ----------------------------------------------------------------------------
def sparse(grid):
    flat = grid.ravel()
    idx = np.flatnonzero(flat).astype(np.int32)
    return idx, flat[idx]


def gen_tsc(T, C):
    out = {}
    for ng in (10, 256):
        for pre, key in (('tsc', 'pydens'), ('nbodykit_tsc', 'mesh')):
            g = read_asdf_blsc(REF / 'tests' / 'ref_tsc' / f'{pre}_ngrid{ng}.asdf', key)
            assert g.shape == (ng, ng, ng)
            idx, val = sparse(g)
            out[f'{pre}_ngrid{ng}.idx'] = idx
            out[f'{pre}_ngrid{ng}.val'] = val
            print(pre, ng, g.dtype, 'sum', g.sum(dtype='f8'), 'nnz', len(idx))
    np.savez_compressed(GOLD / 'tsc_ref.npz', **out)

    # check the shimmed reference against its own saved grid (tests/test_tsc.py:105-136)
    box = 123.0
    rng = np.random.default_rng(234)
    pos = rng.random((10000, 3), dtype='f4').astype('f8') * box
    w = rng.random((10000,), dtype='f4').astype('f8')
    d10 = np.zeros((10, 10, 10), dtype=np.float32)
    T._tsc_scatter(pos, d10, box, w)
    idx, val = out['tsc_ngrid10.idx'], out['tsc_ngrid10.val']
    ref = np.zeros(1000, dtype=np.float32); ref[idx] = val
    assert np.allclose(d10.ravel(), ref, rtol=1e-4, atol=1e-5)
    print('shimmed _tsc_scatter == ref_tsc/tsc_ngrid10.asdf (max abs diff %.3g)' % np.abs(d10.ravel() - ref).max())

    cases = {}
    N = 3000
    rng = np.random.default_rng(77)
    base = rng.random((N, 3), dtype='f4')
    wts = rng.random(N, dtype='f4')
    warnings.simplefilter('ignore')
    for name, dtype, shape, useW, offset, boxs in (
        ('f4_w', 'f4', (12, 12, 12), True, 0.0, 50.0),
        ('f4_now', 'f4', (12, 12, 12), False, 0.0, 50.0),
        ('f8_w', 'f8', (12, 12, 12), True, 0.0, 50.0),
        ('f4_offset', 'f4', (16, 16, 16), True, 0.5 * 50.0 / 16, 50.0),
        ('f4_aniso', 'f4', (8, 12, 20), True, 0.0, 50.0),
        ('f8_grid64', 'f8', (9, 9, 9), False, 0.0, 7.0),
    ):
        p = (base.astype(dtype) * boxs).astype(dtype)
        ww = wts.astype(dtype) if useW else None
        gdt = np.float64 if name == 'f8_grid64' else np.float32
        g = np.zeros(shape, dtype=gdt)
        T._tsc_scatter(p, g, boxs, ww, offset)
        cases[f'{name}.grid'] = g
        cases[f'{name}.box'] = np.float64(boxs)
        cases[f'{name}.offset'] = np.float64(offset)
    # tsc_parallel end to end incl. in-place wrap of out-of-box particles and accumulation
    p = ((base - np.float32(0.3)) * np.float32(1.6) * np.float32(50.0)).astype('f4')  # spills both sides
    p0 = p.copy()
    g = np.full((12, 12, 12), 0.25, dtype=np.float32)
    r = T.tsc_parallel(p, g, 50.0, weights=wts, nthread=1)
    assert r is None
    cases['parallel_wrap.pos_in'] = p0
    cases['parallel_wrap.pos_out'] = p
    cases['parallel_wrap.grid'] = g
    cases['base'] = base
    cases['wts'] = wts
    # partition_parallel (stable counting sort)
    pp = (base * np.float32(50.0)).astype('f4')
    ps, st, ws = T.partition_parallel(pp, 7, 50.0, weights=wts, nthread=1)
    cases['partition.psort'] = ps
    cases['partition.starts'] = st
    cases['partition.wsort'] = ws
    # CIC (cic_serial) - float64 math into a float32 grid
    g = np.zeros((12, 12, 12), dtype=np.float32)
    C.cic_serial((base * np.float32(50.0)).astype('f4'), g, 50.0, weights=wts)
    cases['cic_f4_w.grid'] = g
    np.savez_compressed(GOLD / 'tsc_cases.npz', **cases)
    print('tsc_cases written')


# ----------------------------------------------------------------------------
# power spectrum
# ----------------------------------------------------------------------------
def gen_power(P):
    from abacusutils_amd import synth
    warnings.simplefilter('ignore')
    L = 500.0
    N = 20000
    pos = synth.synth_positions(N, L, seed=300, clustered=True)
    pos2 = synth.synth_positions(N // 2, L, seed=301, clustered=True)
    rng = np.random.default_rng(5)
    w = (0.5 + rng.random(N, dtype='f4')).astype('f4')
    out = {'meta.L': np.float64(L), 'meta.N': np.int64(N)}

    def store(name, tab):
        for k in ('k_avg', 'power', 'N_mode', 'poles', 'N_mode_poles', 'k_mid', 'mu_mid'):
            if k in tab:
                out[f'{name}.{k}'] = np.asarray(tab[k])

    nmesh = 32
    for paste in ('TSC', 'CIC'):
        for comp in (False, True):
            for inter in (False, True):
                name = f'{paste}_c{int(comp)}_i{int(inter)}'
                tab = P.calc_power(pos.copy(), L, kbins=12, mubins=4, k_max=np.pi * nmesh / L + 1e-6, paste=paste,
                                   nmesh=nmesh, compensated=comp, interlaced=inter, poles=[0, 2, 4], nthread=1)
                store(name, tab)
                print(name, 'P[1:4,0]', np.asarray(tab['power'])[1:4, 0])
    # weights, cross, logk, squeeze, default bins
    tab = P.calc_power(pos.copy(), L, kbins=10, mubins=None, paste='TSC', nmesh=nmesh, compensated=True,
                       interlaced=False, w=w, poles=[0, 2], nthread=1)
    store('TSC_weights_squeeze', tab)
    tab = P.calc_power(pos.copy(), L, kbins=9, mubins=3, paste='TSC', nmesh=nmesh, compensated=True,
                       interlaced=True, pos2=pos2.copy(), poles=[0, 2, 4], nthread=1)
    store('TSC_cross', tab)
    tab = P.calc_power(pos.copy(), L, kbins=8, mubins=2, logk=True, paste='TSC', nmesh=nmesh, compensated=False,
                       interlaced=False, nthread=1)
    store('TSC_logk', tab)
    tab = P.calc_power(pos.copy(), L, paste='TSC', nmesh=24, compensated=True, interlaced=True, nthread=1)
    store('TSC_defaults_n24', tab)
    # odd mesh
    tab = P.calc_power(pos.copy(), L, kbins=7, mubins=2, paste='TSC', nmesh=27, compensated=True, interlaced=True,
                       poles=[0, 2], nthread=1)
    store('TSC_odd27', tab)

    # calc_pk_from_deltak / bin_kmu on a seeded complex field (zcv call pattern)
    n = 20
    rng = np.random.default_rng(9)
    f1 = (rng.standard_normal((n, n, n // 2 + 1)) + 1j * rng.standard_normal((n, n, n // 2 + 1))).astype(np.complex64)
    f2 = (rng.standard_normal((n, n, n // 2 + 1)) + 1j * rng.standard_normal((n, n, n // 2 + 1))).astype(np.complex64)
    ke, me = P.get_k_mu_edges(L, np.pi * n / L, 6, 3, False)
    r = P.calc_pk_from_deltak(f1, L, ke, me, field2_fft=f2, poles=np.array([0, 2, 4, 6]), nthread=1)
    out['deltak.f1'] = f1
    out['deltak.f2'] = f2
    for k, v in r.items():
        out[f'deltak.cross.{k}'] = np.asarray(v)
    r = P.calc_pk_from_deltak(f1, L, ke, me, poles=np.array([], dtype='i8'), squeeze_mu_axis=False, nthread=1)
    for k, v in r.items():
        out[f'deltak.auto.{k}'] = np.asarray(v)
    # window functions
    for paste in ('TSC', 'CIC'):
        for inter in (False, True):
            out[f'W.{paste}_i{int(inter)}'] = P.get_W_compensated(L, 32, paste, inter)
    np.savez_compressed(GOLD / 'power_cases.npz', **out)
    print('power_cases written')


def gen_helpers(P):
    """ZCV-facing spectrum helpers: bin_kppi, project_3d_to_poles, pk_to_xi, expand_poles_to_3d,
    get_smoothing, get_delta_mu2 of the reference on seeded inputs (plain-Python loops under the shim: small meshes)"""
    warnings.simplefilter('ignore')
    L = 400.0
    out = {'meta.L': np.float64(L)}
    for n in (16, 21):
        kz = n // 2 + 1
        rng = np.random.default_rng(40 + n)
        p3d = (rng.random((n, n, kz), dtype=np.float32) * 100).astype(np.float32)
        out[f'n{n}.p3d'] = p3d
        ke = np.linspace(0.0, np.pi * n / L * 0.9, 7)
        out[f'n{n}.kedges'] = ke
        # bin_kppi in Fourier space and in configuration space
        # pimax beyond the largest kz: below it the reference indexes past its pi edges (unchecked under Numba) before it breaks
        wc, c = P.bin_kppi(n, L, ke, pimax=np.pi * n / L * 1.01, Npi=5, weights=p3d, nthread=1)
        out[f'n{n}.kppi.mean'], out[f'n{n}.kppi.counts'] = wc, c
        xi = rng.standard_normal((n, n, n)).astype(np.float32)
        re = np.linspace(0.0, L / 3, 6)
        out[f'n{n}.xi'], out[f'n{n}.redges'] = xi, re
        wc, c = P.bin_kppi(n, L, re, pimax=L / 2 * 1.01, Npi=4, weights=xi, fourier=False, nthread=1)
        out[f'n{n}.rppi.mean'], out[f'n{n}.rppi.counts'] = wc, c
        # multipoles of a 3-D power spectrum
        bp, npo = P.project_3d_to_poles(ke, p3d, L, [0, 2, 4])
        out[f'n{n}.p2poles.poles'], out[f'n{n}.p2poles.N'] = bp, npo
        # correlation-function multipoles
        rb, xp, nr = P.pk_to_xi(p3d.copy(), L, re, poles=[0, 2, 4])
        out[f'n{n}.pk2xi.r'], out[f'n{n}.pk2xi.poles'], out[f'n{n}.pk2xi.N'] = rb, xp, nr
        # multipoles -> 3-D grid
        k_ell = np.linspace(0.01, 0.3, 12)
        P_ell = (rng.random((3, 12)) * 1e3).astype(np.float64)
        out[f'n{n}.expand.k_ell'], out[f'n{n}.expand.P_ell'] = k_ell, P_ell
        out[f'n{n}.expand.Pk'] = P.expand_poles_to_3d(k_ell, P_ell, n, L, np.array([0, 2, 4]))
        out[f'n{n}.smoothing'] = P.get_smoothing(n, L, 7.5)
        d = (rng.standard_normal((n, n, kz)) + 1j * rng.standard_normal((n, n, kz))).astype(np.complex64)
        out[f'n{n}.delta'] = d
        out[f'n{n}.delta_mu2'] = P.get_delta_mu2(d, n)
    np.savez_compressed(GOLD / 'power_helpers.npz', **out)
    print('power_helpers written')


def mock_digest(mock):
    """{tracer: (N, Ncent, sha256 of the eight columns' bytes)} of a gen_gal_cat result"""
    import hashlib
    out = {}
    for tr, cols in mock.items():
        h = hashlib.sha256()
        for c in ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass'):
            h.update(np.ascontiguousarray(cols[c], dtype=np.float64).tobytes())
        h.update(np.ascontiguousarray(cols['id'], dtype=np.int64).tobytes())
        out[tr] = (len(cols['x']), int(cols['Ncent']), np.frombuffer(h.digest(), dtype=np.uint8).copy())
    return out


def gen_sweep(G):
    """40 seeded random parameter sets (tests/sweep.py: assembly bias, conformity, ranks, velocity bias, tracer subsets,
    light-cone RSD) through the shimmed reference; the catalogues are kept as digests (counts + SHA-256 of the columns),
    the inputs are regenerated from the seeds by the tests."""
    sys.path.insert(0, str(REPO / 'tests'))
    from sweep import sweep_case
    out = {}
    for seed in range(40):
        hd, pd, params, tracers, ranks, rsd = sweep_case(seed)
        out[f'case{seed}.checksum'] = np.float64(checksum(hd, pd))
        mock = G.gen_gal_cat({k: v.copy() for k, v in hd.items()}, {k: v.copy() for k, v in pd.items()}, tracers, params,
                             Nthread=1, enable_ranks=ranks, rsd=rsd, write_to_disk=False)
        for tr, (n, nc, sha) in mock_digest(mock).items():
            out[f'case{seed}.{tr}.n'], out[f'case{seed}.{tr}.ncent'], out[f'case{seed}.{tr}.sha'] = np.int64(n), np.int64(nc), sha
        print('sweep', seed, {tr: (len(m['x']), m['Ncent']) for tr, m in mock.items()})
    np.savez_compressed(GOLD / 'hod_sweep.npz', **out)
    print('hod_sweep written')


def gen_catalog():
    """Catalogue side: the reference's unpack_rvint / unpack_pids on the Mini_N64_L32 subsample
    files (tests/Mini_N64_L32/halos/z0.000/{halo,field}_{rv,pid}_A) and do_Menv_from_tree on the Mini halos and on
    seeded synthetic halos (periodic and open geometry)."""
    # abacusnbody/data/__init__.py configures astropy's IERS tables at import: bypass it with a bare package object
    dpkg = types.ModuleType('abacusnbody.data')
    dpkg.__path__ = [str(REF / 'abacusnbody' / 'data')]
    sys.modules['abacusnbody.data'] = dpkg
    import abacusnbody.data.bitpacked as B
    import abacusnbody.hod.menv as M
    base = REF / 'tests' / 'Mini_N64_L32' / 'halos' / 'z0.000'
    out = {}
    rv = np.concatenate([read_asdf_blsc(base / f'{k}_rv_A' / f'{k}_rv_A_{i:03d}.asdf', 'rvint')
                         for k in ('halo', 'field') for i in range(3)])
    # a few extreme words: sign bit set (negative positions), all-ones / all-zeros velocity fields
    rv = np.concatenate([rv, np.array([[-2147483648, 2147483647, -1], [0, 4095, -4096]], dtype=np.int32)])
    out['rvint.in'] = rv
    for ft, tag in ((np.float32, 'f4'), (np.float64, 'f8')):
        pos, vel = B.unpack_rvint(rv.copy(), 32.0, float_dtype=ft)
        out[f'rvint.pos.{tag}'], out[f'rvint.vel.{tag}'] = pos, vel
    pid = np.concatenate([read_asdf_blsc(base / f'{k}_pid_A' / f'{k}_pid_A_{i:03d}.asdf', 'packedpid')
                          for k in ('halo', 'field') for i in range(3)])
    pid = np.concatenate([pid, np.array([0xFFFFFFFFFFFFFFFF, 0, 0x07FE000000000000, 1 << 48], dtype=np.uint64)])
    out['pids.in'] = pid
    for ft, tag in ((np.float32, 'f4'), (np.float64, 'f8')):
        r = B.unpack_pids(pid.copy(), box=32.0, ppd=64, pid=True, lagr_pos=True, tagged=True, density=True,
                          lagr_idx=True, float_dtype=ft)
        for k, v in r.items():
            out[f'pids.{k}.{tag}'] = v
    print('rvint', rv.shape, 'pids', pid.shape)
    # local mass environment: the Mini halos as prepare_sim passes them (x_L2com f32, N * Mpart, r98 f32; :603-612)
    sub = REF / 'tests' / 'ref_hod' / 'Mini_N64_L32' / 'z0.000'
    H = np.concatenate([read_h5(sub / f'halos_xcom_{i}_seed600_abacushod_oldfenv_MT_new.h5', 'halos') for i in range(3)])
    Mpart = 1.088239739e10
    cases = {'mini': dict(pos=np.ascontiguousarray(H['x_L2com']), mass=H['N'] * Mpart, r_inner=np.ascontiguousarray(H['r98_L2com']),
                          r_outer=5.0, halo_lc=False, Lbox=32.0, mcut=1e11)}
    rng = np.random.default_rng(77)
    n = 4000
    cen = rng.random((40, 3)) * 300 - 150
    p = (cen[rng.integers(0, 40, n)] + rng.standard_normal((n, 3)) * 6).astype(np.float32)
    p = ((p + 150) % 300 - 150).astype(np.float32)
    m = 10 ** (10.5 + rng.exponential(0.5, n))
    cases['synth_periodic'] = dict(pos=p, mass=m, r_inner=(0.2 + rng.random(n)).astype(np.float32), r_outer=5.0,
                                   halo_lc=False, Lbox=300.0, mcut=1e11)
    cases['synth_router_array'] = dict(pos=p.astype(np.float64), mass=m.astype(np.float32), r_inner=0.5,
                                       r_outer=3.0 + 4 * rng.random(n), halo_lc=False, Lbox=300.0, mcut=3e10)
    cases['synth_lightcone'] = dict(pos=(p * np.float32(3) + np.float32(500)).astype(np.float32), mass=m,
                                    r_inner=(0.2 + rng.random(n)).astype(np.float32), r_outer=12.0, halo_lc=True,
                                    Lbox=2000.0, mcut=1e11)
    for name, c in cases.items():
        got = M.do_Menv_from_tree(c['pos'], c['mass'], r_inner=c['r_inner'], r_outer=c['r_outer'], halo_lc=c['halo_lc'],
                                  Lbox=c['Lbox'], nthread=1, mcut=c['mcut'])
        for k, v in c.items():
            out[f'menv.{name}.{k}'] = np.asarray(v)
        out[f'menv.{name}.Menv'] = got
        print('menv', name, 'centres', int((c['mass'] > c['mcut']).sum()), 'nonzero', int((got != 0).sum()), got.dtype)
    # ---- outputs of the REAL (Numba-compiled) reference held by its own tests (tests/test_data.py:258-326):
    #      read_asdf of field_rv_A_000 / field_pid_A_000 -> ref_data/test_read_asdf.asdf; of L0_pack9/slab000 and
    #      L0_pack9_pid/slab000 -> ref_data/test_pack9.asdf, test_pack9_pid.asdf.  Inputs and expected outputs are copied
    #      into the fixture file; BoxSize 32, ppd 64, VelZSpace_to_kms 3200 from the file headers.
    T = REF / 'tests'
    real = read_asdf_arrays(T / 'ref_data' / 'test_read_asdf.asdf')
    out['real.rvint.in'] = read_asdf_arrays(base / 'field_rv_A' / 'field_rv_A_000.asdf')['rvint']
    out['real.rvint.pos'], out['real.rvint.vel'] = real['pos'], real['vel']
    out['real.pids.in'] = read_asdf_arrays(base / 'field_pid_A' / 'field_pid_A_000.asdf')['packedpid']
    assert np.array_equal(out['real.pids.in'], real['aux'])
    for k in ('pid', 'lagr_pos', 'lagr_idx', 'tagged', 'density'):
        out[f'real.pids.{k}'] = real[k]
    sl = T / 'Mini_N64_L32' / 'slices' / 'z0.000'
    p9 = read_asdf_arrays(sl / 'L0_pack9' / 'slab000.L0.pack9.asdf')['pack9'].view(np.uint8)
    r9 = read_asdf_arrays(T / 'ref_data' / 'test_pack9.asdf')
    out['real.pack9.in'], out['real.pack9.pos'], out['real.pack9.vel'] = p9, r9['pos'], r9['vel']
    hdr = open(sl / 'L0_pack9' / 'slab000.L0.pack9.asdf', 'rb').read(20000).decode('latin1')
    out['real.pack9.box'] = np.float64(re.search(r'BoxSize: ([\d.eE+-]+)', hdr).group(1))
    out['real.pack9.velz'] = np.float64(re.search(r'VelZSpace_to_kms: ([\d.eE+-]+)', hdr).group(1))
    rp = read_asdf_arrays(T / 'ref_data' / 'test_pack9_pid.asdf')
    out['real.pack9pid.in'] = rp['aux']
    for k in ('pid', 'lagr_pos', 'lagr_idx', 'tagged', 'density'):
        out[f'real.pack9pid.{k}'] = rp[k]
    # pack9 cannot run under the identity shim: `c[0] << 4` on NumPy uint8 scalars wraps, where Numba widens to 64 bit
    # (pack9.py:128-130) - so float64 pack9 has no golden; a second input file is kept for HIP-vs-oracle comparisons
    out['pack9.field.in'] = read_asdf_arrays(sl / 'field_pack9' / 'slab002.field.pack9.asdf')['pack9'].view(np.uint8)
    print('pack9 real', p9.shape, '->', r9['pos'].shape)
    np.savez_compressed(GOLD / 'catalog_cases.npz', **out)
    print('catalog_cases written')


# ----------------------------------------------------------------------------
# compute_ngal and the pair-count wrappers: the reference's abacus_hod.py / tpcf_corrfunc.py import third-party
# modules that are absent here (asdf, h5py, parallel_numpy_rng, Corrfunc).  None of them is used by the functions
# captured below, except Corrfunc's counters, for which a stand-in with Corrfunc's calling convention is backed by the
# oracle's brute-force float32 counter (so the golden vectors pin the WRAPPER arithmetic: casts, pi regrouping,
# analytic RR, xi, wp, multipoles - not Corrfunc's kernels, which stay unpinned).
# ----------------------------------------------------------------------------
def _brute_corrfunc():
    from oracle import oracle as O

    def result(n, bins, nsub):
        res = np.zeros((len(bins) - 1) * nsub, dtype=[('rmin', 'f8'), ('rmax', 'f8'), ('npairs', 'u8')])
        res['npairs'] = n
        return res

    def second(autocorr, X2, Y2, Z2):
        return (None, None, None) if autocorr else (X2, Y2, Z2)

    def DDrppi(autocorr, nthreads, binfile=None, pimax=None, X1=None, Y1=None, Z1=None, X2=None, Y2=None, Z2=None,
               periodic=True, boxsize=None, max_cells_per_dim=None, verbose=False):
        assert periodic
        x2, y2, z2 = second(autocorr, X2, Y2, Z2)
        n = O.paircount_brute('rppi', X1, Y1, Z1, float(boxsize), binfile, x2, y2, z2, pimax=float(pimax),
                              npibins=int(pimax), nthread=8)
        return result(n, binfile, int(pimax))

    def DDsmu(autocorr, nthreads, binfile, mu_max, nmu_bins, X1, Y1, Z1, X2=None, Y2=None, Z2=None, periodic=True,
              boxsize=None, max_cells_per_dim=None, verbose=False):
        assert periodic
        x2, y2, z2 = second(autocorr, X2, Y2, Z2)
        n = O.paircount_brute('smu', X1, Y1, Z1, float(boxsize), binfile, x2, y2, z2, mu_max=float(mu_max),
                              nmubins=int(nmu_bins), nthread=8)
        return result(n, binfile, int(nmu_bins))

    return DDrppi, DDsmu


def import_reference_hod():
    """abacusnbody.hod.abacus_hod / analysis.tpcf_corrfunc with inert stand-ins for their absent imports"""
    for name in ('asdf', 'h5py', 'parallel_numpy_rng', 'Corrfunc', 'Corrfunc.theory'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['parallel_numpy_rng'].MTGenerator = None
    sys.modules['Corrfunc.theory'].DDrppi, sys.modules['Corrfunc.theory'].DDsmu = _brute_corrfunc()
    import abacusnbody.analysis.tpcf_corrfunc as TP
    import abacusnbody.hod.abacus_hod as AH
    return AH, TP


NGAL_CASES = {
    'defaults': None,   # filled from synth.*_PARAMS
    'ab_zpivot': {'LRG': dict(Acent=0.3, Asat=-0.2, Bcent=0.1, Bsat=0.4, logM_cut_pr=0.5, logM1_pr=-0.3, z_pivot=0.8, ic=0.9),
                  'ELG': dict(Acent=0.2, Bsat=-0.3, Ccent=0.5, Csat=0.25, logM1_EE=13.0, alpha_EE=0.8),
                  'QSO': dict(Bcent=-0.4, Asat=0.3, ic=0.7, logM_cut_pr=-0.2, z_pivot=1.1)},
    # the conformity defaults of compute_ngal are the RAW logM1 / alpha although logM1 itself evolves (:925-948, 1050)
    'elg_evolving_conformity_defaults': {'ELG': dict(logM1_pr=0.6, logM_cut_pr=0.2, z_pivot=0.9, Asat=0.15)},
}


def gen_ngal(nbin=12, nhalo=30000):
    import contextlib
    import io
    import json

    from abacusutils_amd import synth
    AH, _ = import_reference_hod()
    hd, _, params = synth.synth_hod_inputs(nhalo, 10, seed=77)
    ball = AH.AbacusHOD.__new__(AH.AbacusHOD)
    ball.z_mock = params['z']
    n = len(hd['hmass'])
    # the histograms of AbacusHOD.__init__ (:200-251) on an `nbin`-cell grid per dimension (the reference hard-codes 100,
    # 10^8 cells for ELG: hours of pure-Python loops under the shim; the kernels take the grid from the edge arrays)
    ball.logMbins = np.linspace(np.log10(np.min(hd['hmass'])), np.log10(np.max(hd['hmass'])), nbin + 1)
    ball.deltacbins = np.linspace(-0.5, 0.5, nbin + 1)
    ball.fenvbins = np.linspace(-0.5, 0.5, nbin + 1)
    ball.shearbins = np.linspace(-0.5, 0.5, nbin + 1)
    cols = [np.log10(hd['hmass']), hd['hdeltac'], hd['hfenv'], hd['hshear']]
    ball.halo_mass_func, _ = np.histogramdd(np.vstack(cols[:3]).T, bins=[ball.logMbins, ball.deltacbins, ball.fenvbins],
                                            weights=hd['hmultis'])
    ball.halo_mass_func_wshear, _ = np.histogramdd(
        np.vstack(cols).T, bins=[ball.logMbins, ball.deltacbins, ball.fenvbins, ball.shearbins], weights=hd['hmultis'])
    base = {'LRG': synth.LRG_PARAMS, 'ELG': synth.ELG_PARAMS, 'QSO': synth.QSO_PARAMS}
    out = dict(nbin=nbin, z=params['z'], hmass=hd['hmass'], hdeltac=hd['hdeltac'], hfenv=hd['hfenv'], hshear=hd['hshear'],
               hmultis=hd['hmultis'])
    cases = {}
    for name, over in NGAL_CASES.items():
        tracers = {t: dict(base[t], **(over or {}).get(t, {})) for t in (over or base)}
        if name == 'elg_evolving_conformity_defaults':
            for k in ('logM1_EE', 'logM1_EL', 'alpha_EE', 'alpha_EL', 'A_s'):
                tracers['ELG'].pop(k, None)
        with contextlib.redirect_stdout(io.StringIO()):      # compute_ngal prints `newngal` for ELG (:949)
            ngal, fsat = AH.AbacusHOD.compute_ngal(ball, tracers, Nthread=1)
        cases[name] = tracers
        for t in tracers:
            out[f'{name}.{t}.ngal'] = np.float64(ngal[t])
            out[f'{name}.{t}.fsat'] = np.float64(fsat[t])
            print('ngal', name, t, ngal[t], fsat[t])
    out['cases_json'] = np.array(json.dumps(cases))
    np.savez_compressed(GOLD / 'ngal.npz', **out)
    print('ngal.npz written')


def gen_pair_wrappers():
    """calc_xirppi_fast / calc_wp_fast / calc_multipole_fast / tpcf_multipole of the reference (tpcf_corrfunc.py:17-372)
    on seeded clustered points, auto and cross, float64 inputs (the wrappers cast)"""
    import contextlib
    import io
    _, TP = import_reference_hod()
    rng = np.random.default_rng(4242)
    L = 120.0
    n1, n2 = 1800, 1300
    centres = rng.random((60, 3)) * L
    a = (centres[rng.integers(0, 60, n1)] + rng.normal(0, 3.0, (n1, 3))) % L
    b = np.concatenate([(centres[rng.integers(0, 60, n2 // 2)] + rng.normal(0, 5.0, (n2 // 2, 3))) % L,
                        rng.random((n2 - n2 // 2, 3)) * L])
    rpbins = np.logspace(-0.5, 1.3, 8)
    sbins = np.linspace(0.5, 25.0, 9)
    pimax, pi_bin_size, nmu = 24, 4, 10
    out = dict(L=L, a=a, b=b, rpbins=rpbins, sbins=sbins, pimax=pimax, pi_bin_size=pi_bin_size, nbins_mu=nmu)
    with contextlib.redirect_stdout(io.StringIO()):
        for tag, second in (('auto', {}), ('cross', dict(x2=b[:, 0], y2=b[:, 1], z2=b[:, 2]))):
            out[f'{tag}.xirppi'] = TP.calc_xirppi_fast(a[:, 0], a[:, 1], a[:, 2], rpbins, pimax, pi_bin_size, L, 4, **second)
            out[f'{tag}.wp'] = TP.calc_wp_fast(a[:, 0], a[:, 1], a[:, 2], rpbins, pimax, L, 4, **second)
            out[f'{tag}.multipole'] = TP.calc_multipole_fast(a[:, 0], a[:, 1], a[:, 2], sbins, L, 4, nbins_mu=nmu,
                                                            orders=[0, 2, 4], **second)
    xi = rng.normal(size=(6, 12))
    mub = np.linspace(0, 1, 13)
    out['tpcf.xi'], out['tpcf.mu_bins'] = xi, mub
    for ell in (0, 1, 2, 4):
        out[f'tpcf.l{ell}'] = TP.tpcf_multipole(xi, mub, order=ell)
    for k in ('auto.xirppi', 'auto.wp', 'auto.multipole', 'cross.wp'):
        print(k, out[k].dtype, out[k].shape, float(np.ravel(out[k])[0]))
    np.savez_compressed(GOLD / 'pair_wrappers.npz', **out)
    print('pair_wrappers.npz written')


# ----------------------------------------------------------------------------
# prepare_sim.prepare_slab (hod/prepare_sim.py:296-1052): the REFERENCE's function run on seeded synthetic slabs.  Its
# file layer is replaced by stand-ins: a CompaSOHaloCatalog that hands over the tables of
# abacusutils_amd.synth.synth_compaso_slabs (the CompaSO / ASDF readers are out of scope), an h5py that keeps what
# create_dataset receives.  Everything between the loader and the writer - halo down-sampling, padded Menv, concentration /
# shear ranks, per-halo particle selection, satellite ranks, host columns, random columns - is the reference's own code
# consuming NumPy's global generator in its own order.
# ----------------------------------------------------------------------------
PREPARE_CASES = {
    # name: (slab, MT, want_ranks, want_AB, want_shear)
    'mt_ab': (1, True, False, True, False),            # the configuration of the reference's own fixture (tests/abacus_hod.yaml)
    'lrg_ranks_ab': (0, False, True, True, False),
    # (want_shear without want_AB fails in the reference itself: :784 touches deltac_rank, which only want_AB defines)
    'mt_ranks_ab_shear': (2, True, True, True, True),
}
PREPARE_SYNTH = dict(numslabs=3, n_halo=1500, seed=900, lbox=300.0)
# halo light cones (halo_lc=True, :362-433 loader keys, :474-616 environment with the edge correction by randoms):
# name: (geometry of synth.synth_lightcone_slab, slab index i (enters the seeds), MT, want_ranks)
PREPARE_LC_CASES = {
    'lc_octant': ('octant', 0, True, False),      # three origins, as the base boxes
    'lc_centre': ('centre', 2, False, True),      # one observer in the middle of the box, as the huge boxes
}
PREPARE_LC_SYNTH = dict(n_halo=2500, seed=950, lbox=300.0)


def prepare_shearmark(ndim=16, seed=5):
    return np.random.default_rng(seed).random((ndim, ndim, ndim))


def gen_prepare():
    from astropy.table import Table
    from abacusutils_amd import synth
    slabs, header = synth.synth_compaso_slabs(**PREPARE_SYNTH)
    captured = {}

    lc = {}     # the light-cone slab the stand-in catalogue serves next (its file name carries no slab index)

    class FakeCat:
        def __init__(self, slabname, subsamples=None, fields=None, cleaned=True, filter_func=None, **kw):
            self.halo_lc = str(slabname).endswith('lc_halo_info.asdf')
            if self.halo_lc:
                self.header = dict(lc['header'])
                rename = {'id': 'index_halo', 'x_L2com': 'pos_interp', 'v_L2com': 'vel_interp', 'N': 'N_interp'}   # (:370-373)
                self.halos = Table({rename.get(k, k): v.copy() for k, v in lc['slab']['halos'].items()})
                if subsamples:
                    self.subsamples = Table({k: v.copy() for k, v in lc['slab']['parts'].items()})
                return
            i = int(re.search(r'halo_info_(\d+)\.asdf', str(slabname)).group(1))
            self.header = dict(header)
            H = Table({k: v.copy() for k, v in slabs[i]['halos'].items()})
            if filter_func is not None:
                H = H[filter_func(H)]
            self.halos = H
            if subsamples:
                self.subsamples = Table({k: v.copy() for k, v in slabs[i]['parts'].items()})

    class FakeH5File:
        def __init__(self, fn, mode='r'):
            self.fn = str(fn)

        def create_dataset(self, name, data=None):
            captured[(os.path.basename(self.fn), name)] = {k: np.asarray(v) for k, v in data.items()} if isinstance(data, dict) else np.asarray(data)

        def close(self):
            pass

    for name in ('asdf', 'h5py'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['h5py'].File = FakeH5File
    dpkg = types.ModuleType('abacusnbody.data')
    dpkg.__path__ = [str(REF / 'abacusnbody' / 'data')]
    sys.modules['abacusnbody.data'] = dpkg
    fake = types.ModuleType('abacusnbody.data.compaso_halo_catalog')
    fake.CompaSOHaloCatalog = FakeCat
    sys.modules['abacusnbody.data.compaso_halo_catalog'] = fake
    ra = types.ModuleType('abacusnbody.data.read_abacus')
    ra.read_asdf = None
    sys.modules['abacusnbody.data.read_abacus'] = ra
    import abacusnbody.hod.prepare_sim as PS
    out = {'meta.header_json': np.array(__import__('json').dumps(header)), 'meta.synth_json': np.array(__import__('json').dumps(PREPARE_SYNTH))}
    with tempfile.TemporaryDirectory() as td:
        for case, (i, MT, want_ranks, want_AB, want_shear) in PREPARE_CASES.items():
            captured.clear()
            shear = prepare_shearmark() if want_shear else None
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                PS.prepare_slab(i, td, '/sim', 'Synth', 0.5, 'primary', {}, MT, want_ranks, want_AB, want_shear, shear, True, 600,
                                halo_lc=False, nthread=1, overwrite=1, mcut=1e11, rad_outer=10, numslabs=PREPARE_SYNTH['numslabs'])
            for (fn, dset), val in captured.items():
                kind = 'env' if fn.startswith('env_') else dset
                if isinstance(val, dict):
                    for k, v in val.items():
                        out[f'{case}.{kind}.{k}'] = v
                else:
                    out[f'{case}.{kind}.{dset}'] = val
            nk = len(out[f'{case}.halos.id'])
            print(case, 'halos kept', nk, 'particles kept', len(out[f'{case}.particles.pos']),
                  'env' if f'{case}.env.Menv' in out else 'no env')
        # light cones: the corrected environment masses are an intermediate of the reference (the argument of its
        # calc_fenv_opt, :618) - recorded on the way through, next to the tables it writes
        ranker = PS.calc_fenv_opt
        seen = {}

        def recording_ranker(Menv, mbins, halosM):
            seen['Menv'] = np.array(Menv, dtype=np.float64)
            return ranker(Menv, mbins, halosM)

        PS.calc_fenv_opt = recording_ranker
        out['meta.lc_synth_json'] = np.array(__import__('json').dumps(PREPARE_LC_SYNTH))
        for case, (geometry, i, MT, want_ranks) in PREPARE_LC_CASES.items():
            captured.clear()
            seen.clear()
            lc['slab'], lc['header'] = synth.synth_lightcone_slab(geometry=geometry, **PREPARE_LC_SYNTH)
            with contextlib.redirect_stdout(io.StringIO()):
                PS.prepare_slab(i, td, '/sim', 'Synth', 0.5, 'lightcone', {}, MT, want_ranks, True, False, None, True, 600,
                                halo_lc=True, nthread=1, overwrite=1, mcut=1e11, rad_outer=10, numslabs=1)
            for (fn, dset), val in captured.items():
                for k, v in val.items():
                    out[f'{case}.{dset}.{k}'] = v
            out[f'{case}.Menv_corrected'] = seen['Menv']
            print(case, 'halos kept', len(out[f'{case}.halos.id']), 'particles kept', len(out[f'{case}.particles.pos']),
                  'Menv > 0:', int((seen['Menv'] > 0).sum()))
        PS.calc_fenv_opt = ranker
    np.savez_compressed(GOLD / 'prepare_sim.npz', **out)
    print('prepare_sim written', os.path.getsize(GOLD / 'prepare_sim.npz') // 1024, 'KiB')


def gen_exports(P):
    """the pieces of the calc_power chain the reference exports as functions of their own: bin_kmu, get_raw_power,
    shift_field_fft, get_interlaced_field_fft (analysis/power_spectrum.py:150-300, 707-727, 904-998) on seeded inputs"""
    warnings.simplefilter('ignore')
    sys.path.insert(0, str(REPO))
    from abacusutils_amd import synth
    L = 300.0
    out = {'meta.L': np.float64(L)}
    for n in (16, 21):
        kz = n // 2 + 1
        rng = np.random.default_rng(90 + n)
        f1 = (rng.standard_normal((n, n, kz)) + 1j * rng.standard_normal((n, n, kz))).astype(np.complex64)
        f2 = (rng.standard_normal((n, n, kz)) + 1j * rng.standard_normal((n, n, kz))).astype(np.complex64)
        out[f'n{n}.f1'], out[f'n{n}.f2'] = f1, f2
        out[f'n{n}.raw_auto'] = P.get_raw_power(f1)
        out[f'n{n}.raw_cross'] = P.get_raw_power(f1, f2)
        ke = np.linspace(0.0, np.pi * n / L * 0.95, 8)
        me = np.linspace(0.0, 1.0, 4)
        out[f'n{n}.kedges'], out[f'n{n}.muedges'] = ke, me
        res = P.bin_kmu(n, L, ke, me, out[f'n{n}.raw_auto'], poles=np.array([0, 2, 4]), nthread=1)
        for name, a in zip(('power', 'N_mode', 'poles', 'N_mode_poles', 'k_avg'), res):
            out[f'n{n}.kmu.{name}'] = a
        xi = rng.standard_normal((n, n, n)).astype(np.float32)
        re = np.linspace(0.0, L / 3, 6)
        out[f'n{n}.xi'], out[f'n{n}.redges'] = xi, re
        res = P.bin_kmu(n, L, re, np.array([0.0, 0.5, 1.0]), xi, poles=np.array([0, 2]), fourier=False, nthread=1)
        for name, a in zip(('power', 'N_mode', 'poles', 'N_mode_poles', 'k_avg'), res):
            out[f'n{n}.rmu.{name}'] = a
        a = f1.copy()
        P.shift_field_fft(a, f2, n, L, L / n)
        out[f'n{n}.shifted'] = a
        pos = synth.synth_positions(3000, L, seed=70 + n, clustered=True)
        w = (0.5 + rng.random(3000, dtype=np.float32)).astype(np.float32)
        out[f'n{n}.w'] = w
        out[f'n{n}.il_tsc'] = P.get_interlaced_field_fft(pos.copy(), L, n, 'TSC', None, nthread=1)
        out[f'n{n}.il_cic_w'] = P.get_interlaced_field_fft(pos.copy(), L, n, 'CIC', w, nthread=1)
    np.savez_compressed(GOLD / 'power_exports.npz', **out)
    print('power_exports written')


def gen_power64(P):
    """calc_power / get_field / get_field_fft of the reference with dtype=np.float64 (float64 mesh and transform; its
    interlaced branch ignores dtype, :1048-1052), float32 and float64 positions, TSC and CIC, cross spectrum"""
    warnings.simplefilter('ignore')
    sys.path.insert(0, str(REPO))
    from abacusutils_amd import synth
    L, N = 500.0, 20000
    out = {'meta.L': np.float64(L), 'meta.N': np.int64(N)}
    pos = synth.synth_positions(N, L, seed=300, clustered=True)
    pos2 = synth.synth_positions(N // 2, L, seed=301, clustered=True)
    w = (0.5 + np.random.default_rng(5).random(N, dtype=np.float32)).astype(np.float32)
    out['w'] = w
    for n in (24, 30, 21):
        out[f'n{n}.field_tsc'] = P.get_field(pos.copy(), L, n, 'TSC', w, dtype=np.float64)
        out[f'n{n}.field_cic_p8'] = P.get_field(pos.astype(np.float64), L, n, 'CIC', None, dtype=np.float64)
        W = P.get_W_compensated(L, n, 'TSC', False)
        out[f'n{n}.fft_tsc_comp'] = P.get_field_fft(pos.copy(), L, n, 'TSC', w, W, True, False, dtype=np.float64)
        out[f'n{n}.fft_tsc_p8'] = P.get_field_fft(pos.astype(np.float64), L, n, 'TSC', None, None, False, False, dtype=np.float64)
        il = P.get_field_fft(pos.copy(), L, n, 'TSC', None, W, True, True, dtype=np.float64)
        out[f'n{n}.fft_interlaced_dtype'] = np.array(str(il.dtype))
        for name, kw in (('auto', dict(compensated=True, interlaced=False, w=w)),
                         ('cross', dict(compensated=False, interlaced=False, pos2=pos2.copy())),
                         ('interlaced', dict(compensated=True, interlaced=True))):
            tab = P.calc_power(pos.copy(), L, kbins=10, mubins=3, k_max=np.pi * n / L + 1e-6, paste='TSC', nmesh=n, poles=[0, 2, 4],
                               dtype=np.float64, **kw)
            for c in ('power', 'N_mode', 'poles', 'N_mode_poles', 'k_avg'):
                out[f'n{n}.{name}.{c}'] = np.asarray(tab[c])
    np.savez_compressed(GOLD / 'power_f64.npz', **out)
    print('power_f64 written', os.path.getsize(GOLD / 'power_f64.npz') // 1024, 'KiB')


def gen_mini_prepare():
    """the reference-HELD pin of prepare_sim (tests/test_hod.py:89-100): its Mini_N64_L32 subsample files (all three slabs) as
    arrays -> prepare_mini.npz; the reference reader's own catalogues of that simulation (tests/ref_data/test_halos_clean.asdf,
    test_halos_unclean.asdf, test_subsamples_clean.asdf, test_subsamples_unclean.asdf: outputs of the real CompaSOHaloCatalog,
    tests/test_data.py:29-158) for the columns the MI355X reader unpacks -> compaso_mini.npz; and the simulation's input files
    this needs (halo_info, halo_rv / halo_pid A and B, cleaned_halo_info, cleaned_rvpid: the reference's test DATA) copied to
    tests/golden/Mini_N64_L32/ in the reference's directory layout"""
    import shutil
    sys.path.insert(0, str(REPO))
    from abacusutils_amd.data.asdf import AsdfFile
    sub = REF / 'tests' / 'ref_hod' / 'Mini_N64_L32' / 'z0.000'
    out = {}
    for i in range(3):
        for kind, dset in (('halos', 'halos'), ('particles', 'particles')):
            a = read_h5(sub / f'{kind}_xcom_{i}_seed600_abacushod_oldfenv_MT_new.h5', dset)
            for name in a.dtype.names:
                out[f's{i}.{kind}.{name}'] = np.ascontiguousarray(a[name])
    np.savez_compressed(GOLD / 'prepare_mini.npz', **out)
    cols = ['id', 'npstartA', 'npstartB', 'npoutA', 'npoutB', 'N', 'x_L2com', 'v_L2com', 'sigmav3d_L2com', 'r100_L2com', 'r25_L2com',
            'r50_L2com', 'r90_L2com', 'r98_L2com', 'x_com', 'v_com', 'sigmav3d_com', 'r10_com', 'rvcirc_max_L2com', 'SO_radius',
            'SO_central_particle', 'SO_central_density', 'vcirc_max_L2com', 'meanSpeed_com', 'N_merge', 'is_merged_to', 'haloindex']
    out = {}
    for tag in ('clean', 'unclean'):
        af = AsdfFile(REF / 'tests' / 'ref_data' / f'test_halos_{tag}.asdf')
        for c in cols:
            if c in af.names():
                out[f'halos_{tag}.{c}'] = af.array(c)
        af = AsdfFile(REF / 'tests' / 'ref_data' / f'test_subsamples_{tag}.asdf')
        for c in af.names():
            out[f'subsamples_{tag}.{c}'] = af.array(c)
    np.savez_compressed(GOLD / 'compaso_mini.npz', **out)
    dst = GOLD / 'Mini_N64_L32'
    if dst.exists():
        shutil.rmtree(dst)
    src = REF / 'tests' / 'Mini_N64_L32' / 'halos' / 'z0.000'
    for d in ('halo_info', 'halo_rv_A', 'halo_rv_B', 'halo_pid_A', 'halo_pid_B'):
        (dst / 'Mini_N64_L32' / 'halos' / 'z0.000' / d).mkdir(parents=True)
        for f in sorted((src / d).glob('*.asdf')):
            shutil.copyfile(f, dst / 'Mini_N64_L32' / 'halos' / 'z0.000' / d / f.name)
    csrc = REF / 'tests' / 'cleaning' / 'Mini_N64_L32' / 'z0.000'
    for d in ('cleaned_halo_info', 'cleaned_rvpid'):
        (dst / 'cleaning' / 'Mini_N64_L32' / 'z0.000' / d).mkdir(parents=True)
        for f in sorted((csrc / d).glob('*.asdf')):
            shutil.copyfile(f, dst / 'cleaning' / 'Mini_N64_L32' / 'z0.000' / d / f.name)
    shutil.copyfile(REF / 'tests' / 'abacus_hod.yaml', dst / 'abacus_hod.yaml')
    for f in dst.rglob('*'):
        if f.is_file():
            f.chmod(0o644)
    print('prepare_mini / compaso_mini written;', sum(f.stat().st_size for f in dst.rglob('*') if f.is_file()) // 1024, 'KiB of input files')


if __name__ == '__main__':
    which = sys.argv[1:] or ['hod', 'tsc', 'power', 'helpers', 'exports', 'catalog', 'sweep', 'ngal', 'pairs', 'prepare', 'mini', 'power64']
    G, T, P, C = import_reference()
    GOLD.mkdir(parents=True, exist_ok=True)
    if 'hod' in which:
        gen_hod(G)
    if 'tsc' in which:
        gen_tsc(T, C)
    if 'power' in which:
        gen_power(P)
    if 'helpers' in which:
        gen_helpers(P)
    if 'exports' in which:
        gen_exports(P)
    if 'mini' in which:
        gen_mini_prepare()
    if 'power64' in which:
        gen_power64(P)
    if 'catalog' in which:
        gen_catalog()
    if 'sweep' in which:
        gen_sweep(G)
    if 'ngal' in which:
        gen_ngal()
    if 'pairs' in which:
        gen_pair_wrappers()
    if 'prepare' in which:
        gen_prepare()
