"""CompaSO halo catalogues for the HOD pipeline (reference: abacusnbody/data/compaso_halo_catalog.py:48-420, 600-1370).

`CompaSOHaloCatalog(path, cleaned=True, subsamples=..., fields=..., filter_func=...)` with the reference's argument
meaning for what `prepare_sim` and `AbacusHOD` ask of it: one halo_info slab file (or a redshift directory: all slabs),
the "cleaned" catalogue (halos merged away have N = 0, the others take the particles of what merged into them:
`N_total`, `npstart{A,B}_merge`, `npout{A,B}_merge` of cleaned_halo_info and the particles of cleaned_rvpid), unit
conversion of the halo columns, and the subsample particles re-indexed so that a halo's original particles are followed by
the ones it gained - `npstartA` / `npoutA` then index `subsamples`.  Halo light-cone catalogues (one lc_halo_info.asdf +
lc_pid_rv.asdf, already cleaned) are read as well.  The files are decoded by `abacusutils_amd.data.asdf` (no `asdf`
package); the rvint -> (pos, vel) unpacking of the particles runs on the device (`abacusutils_amd.data.bitpacked`).

Not rebuilt: the long tail of halo columns (eigenvectors, sigmar / sigman, ...), `passthrough`, field particles, the
B-subsample PIDs' Lagrangian unpacking beyond `bitpacked.unpack_pids`: outside the MI355X hot-path scope."""
import re
from pathlib import Path

import numpy as np

from . import bitpacked
from .asdf import AsdfFile

__all__ = ['CompaSOHaloCatalog', 'HaloTable']

INT16SCALE = 32000.0
_RAW = {'id', 'npstartA', 'npstartB', 'npoutA', 'npoutB', 'ntaggedA', 'ntaggedB', 'N', 'L2_N', 'L0_N'}
_CLEAN = {'N_total', 'N_merge', 'npstartA_merge', 'npstartB_merge', 'npoutA_merge', 'npoutB_merge', 'is_merged_to', 'haloindex',
          'haloindex_mainprog', 'N_mainprog', 'v_L2com_mainprog', 'vcirc_max_L2com_mainprog', 'sigmav3d_L2com_mainprog'}
_LC_RAW = {'index_halo', 'pos_avg', 'vel_avg', 'redshift_interp', 'N_interp', 'N', 'npstartA', 'npoutA'}


class HaloTable(dict):
    """columns of equal length; `t[mask]` / `t[slice]` select rows of every column (what prepare_sim does with the astropy
    Table of the reference), `t['name']` a column, `len(t)` the number of rows; `.meta` the header"""

    def __init__(self, cols=None, meta=None):
        super().__init__(cols or {})
        self.meta = dict(meta or {})

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict.__getitem__(self, k)
        return HaloTable({n: v[k] for n, v in dict.items(self)}, self.meta)

    def __len__(self):
        for v in dict.values(self):
            return len(v)
        return 0

    @property
    def colnames(self):
        return list(dict.keys(self))

    def rename_column(self, a, b):
        self[b] = dict.pop(self, a)

    def remove_column(self, a):
        dict.pop(self, a)

    def add_column(self, col, name=None, copy=False):
        self[name] = np.array(col) if copy else col


class CompaSOHaloCatalog:
    def __init__(self, path, cleaned=True, subsamples=False, convert_units=True, unpack_bits=False, fields='DEFAULT_FIELDS',
                 verbose=False, cleandir=None, filter_func=None, halo_lc=None, passthrough=False, **kwargs):
        if passthrough or kwargs:
            raise NotImplementedError(f'CompaSOHaloCatalog: unsupported arguments {["passthrough"] * bool(passthrough) + list(kwargs)}')
        path = Path(path)
        self.halo_lc = self._is_path_halo_lc(path) if halo_lc is None else bool(halo_lc)
        self.cleaned = True if self.halo_lc else bool(cleaned)
        load_clean = self.cleaned and not self.halo_lc           # light cones already incorporate the cleaning
        self.groupdir, halo_fns, clean_fns, self.clean_rvpid_dir, self.superslab_inds = \
            self._setup_file_paths(path, load_clean, cleandir, self.halo_lc)
        self.load_AB, self.load_pidrv = self._setup_load_subsamples(subsamples)
        self.filter_func = filter_func
        self.convert_units = convert_units
        afs = [AsdfFile(f) for f in halo_fns]
        cafs = [AsdfFile(f) for f in clean_fns]
        self.header = dict(afs[0].header)
        self.header['cleaned_halos'] = self.cleaned
        if cafs:
            for k in ('TimeSliceRedshiftsPrev', 'NumTimeSliceRedshiftsPrev'):
                if k in cafs[0].header:
                    self.header[k] = cafs[0].header[k]
        if fields == 'DEFAULT_FIELDS':
            fields = 'all'
        self.fields = self._field_list(fields, afs[0], load_clean)
        self.halos, N_halo_per_file = self._read_halo_info(afs, cafs, load_clean)
        self.subsamples = HaloTable()
        self.numhalos = N_halo_per_file
        if self.halo_lc:
            if self.load_AB:
                af = AsdfFile(Path(self.groupdir) / 'lc_pid_rv.asdf')
                for w in self.load_pidrv:
                    if w in ('pos', 'vel', 'pid'):
                        self.subsamples[w] = af.array(w)
                if 'pid' in self.subsamples and unpack_bits:
                    raise NotImplementedError('unpack_bits of a halo light-cone catalogue')
        elif self.load_AB:
            self._load_subsamples(N_halo_per_file, load_clean, unpack_bits)
        if load_clean:
            self.halos.rename_column('N_total', 'N')

    # ---- paths (compaso_halo_catalog.py:310-418) ----------------------------------------------------------------------
    @staticmethod
    def _is_path_halo_lc(path):
        path = Path(path)
        return 'halo_light_cones' in str(path) or any(path.glob('lc_*.asdf'))

    def _setup_file_paths(self, path, cleaned, cleandir, halo_lc):
        if halo_lc:
            if path.is_file():
                return path.parent, [path], [], None, [0]
            return path, [path / 'lc_halo_info.asdf'], [], None, [0]
        if path.is_file():
            groupdir = path.parents[1]
            halo_fns = [path]
        else:
            groupdir = path
            halo_fns = sorted((path / 'halo_info').glob('halo_info_*.asdf'))
            if not halo_fns:
                raise FileNotFoundError(f'no halo_info files under {path}')
        inds = [int(re.search(r'_(\d+)\.asdf$', f.name).group(1)) for f in halo_fns]
        clean_fns, clean_rvpid_dir = [], None
        if cleaned:
            if not cleandir:
                for p in Path(groupdir).resolve().parents:
                    if (p / 'cleaning').is_dir():
                        cleandir = p / 'cleaning'
                        break
                else:
                    raise FileNotFoundError(f'Could not find cleaning info dir, searching upwards from {groupdir}. To load the '
                                            'uncleaned catalog, use `cleaned=False`.')
            cleandir = Path(cleandir)
            g = Path(groupdir).resolve()
            relpath = (g.parents[1] / g.name).relative_to(cleandir.resolve().parent)   # SimName/z0.000 (halos/ dropped)
            if (cleandir / relpath / 'cleaned_halo_info').is_dir():
                cinfo, clean_rvpid_dir = cleandir / relpath / 'cleaned_halo_info', cleandir / relpath / 'cleaned_rvpid'
            else:
                cinfo = clean_rvpid_dir = cleandir / relpath
            clean_fns = [cinfo / f'cleaned_halo_info_{i:03d}.asdf' for i in inds]
            for fn in clean_fns:
                if not fn.is_file():
                    raise FileNotFoundError(f'Cleaning info not found. File path was: "{fn}". To load the uncleaned catalog, '
                                            'use `cleaned=False`.')
        return groupdir, halo_fns, clean_fns, clean_rvpid_dir, inds

    @staticmethod
    def _setup_load_subsamples(load_subsamples):
        """(:433-512) -> (['A', 'B'] subset, ['pos', 'vel', 'pid'] subset)"""
        if load_subsamples is False or load_subsamples is None:
            return [], []
        if load_subsamples is True:
            load_subsamples = dict(A=True, B=True, rv=True, pid=True)
        s = dict(load_subsamples)
        if 'rv' in s and ('pos' in s or 'vel' in s):
            raise ValueError('Cannot pass `rv` and `pos` or `vel` in `load_subsamples`.')
        for k in s:
            if k not in ('A', 'B', 'rv', 'pid', 'pos', 'vel'):
                raise ValueError(f'Unrecognized keys in `load_subsamples`: {[k]}')
        AB = [k for k in 'AB' if s.get(k)]
        pidrv = [k for k in s if k in ('pid', 'pos', 'vel', 'rv') and s.get(k)]
        if pidrv and not AB:
            AB = ['A']
        elif AB and not pidrv:
            pidrv = ['rv']
        if 'rv' in pidrv:
            pidrv.remove('rv')
            pidrv += ['pos', 'vel']
        return AB, pidrv

    # ---- halo columns (compaso_halo_catalog.py:514-1000) ----------------------------------------------------------------
    def _field_list(self, fields, af, cleaned):
        if isinstance(fields, str):
            if fields != 'all':
                fields = [fields]
            elif self.halo_lc:
                fields = [f for f in ('N', 'N_interp', 'npstartA', 'npoutA', 'index_halo', 'pos_interp', 'vel_interp',
                                      'redshift_interp') if f in af.names() or f.endswith('_interp')]
            else:
                fields = ['id', 'npstartA', 'npstartB', 'npoutA', 'npoutB', 'N', 'x_L2com', 'v_L2com', 'sigmav3d_L2com',
                          'r100_L2com', 'r25_L2com', 'r50_L2com', 'r90_L2com', 'r98_L2com', 'x_com', 'v_com', 'sigmav3d_com']
        fields = list(dict.fromkeys(fields))
        if cleaned:                                   # (:561-597) N is replaced by N_total; the merge indexing rides along
            fields = [f for f in fields if f != 'N']
            fields += ['N_total']
        for AB in self.load_AB:
            for f in (f'npstart{AB}', f'npout{AB}'):
                if f not in fields:
                    fields.append(f)
            if cleaned:
                fields += [f'npstart{AB}_merge', f'npout{AB}_merge']
        return list(dict.fromkeys(fields))

    def _load_field(self, name, raw, craw):
        """one user-facing column from the raw columns of a slab (the loaders of :806-946)"""
        box = self.header['BoxSize'] if self.convert_units else 1.0
        kms = self.header['VelZSpace_to_kms'] if self.convert_units else 1.0
        if self.halo_lc:
            m = re.fullmatch(r'(pos|vel)_interp', name)
            if m:                                     # averaged where an average exists (:905-918)
                avail = np.any(np.atleast_2d(raw('pos_avg')), axis=1)
                return np.where(avail[:, None], raw(m[1] + '_avg'), raw(name))
            if name == 'origin':
                return raw(name) % 3
            if name in _LC_RAW:
                return raw(name)
        if name in _CLEAN:
            return craw(name)
        if name in _RAW:
            return raw(name)
        m = re.fullmatch(r'(?:r\d{1,2}|rvcirc_max)(?P<com>_(?:L2)?com)', name)
        if m:
            return raw(name + '_i16') * raw('r100' + m['com']) / INT16SCALE * box
        if re.fullmatch(r'(x|r100)_(?:L2)?com', name) or re.fullmatch(r'SO(?:_L2max)?(?:_central_particle|_radius)', name):
            return raw(name) * box
        if re.fullmatch(r'(v|sigmav3d|meanSpeed|sigmav3d_r50|meanSpeed_r50|vcirc_max)_(?:L2)?com', name):
            return raw(name) * kms
        if re.fullmatch(r'SO(?:_L2max)?_central_density', name):
            return raw(name)
        raise KeyError(f'Don\'t know how to load halo field "{name}" (not among the columns the MI355X build unpacks)')

    def _read_halo_info(self, afs, cafs, cleaned):
        per_file, n_per = [], []
        for i, af in enumerate(afs):
            caf = cafs[i] if cleaned else None
            cache = {}

            def raw(k, af=af, cache=cache):
                if k not in cache:
                    cache[k] = af.array(k)
                return cache[k]

            def craw(k, caf=caf, cache=cache):
                if k not in cache:
                    cache[k] = caf.array(k)
                return cache[k]

            cols = {f: np.asarray(self._load_field(f, raw, craw)) for f in self.fields}
            t = HaloTable(cols, self.header)
            if self.filter_func is not None:
                if cleaned:
                    t.rename_column('N_total', 'N')      # the user's filter sees 'N' (:757-760)
                t = t[np.asarray(self.filter_func(t), dtype=bool)]
                if cleaned:
                    t.rename_column('N', 'N_total')
            per_file.append(t)
            n_per.append(len(t))
        cols = {f: np.concatenate([t[f] for t in per_file]) for f in self.fields}
        return HaloTable(cols, self.header), np.array(n_per, dtype=np.int64)

    # ---- subsample particles (compaso_halo_catalog.py:1031-1370) --------------------------------------------------------
    def _load_subsamples(self, N_halo_per_file, cleaned, unpack_bits):
        H = self.halos
        nh = len(H)
        file_off = np.concatenate([[0], np.cumsum(N_halo_per_file)]).astype(np.int64)
        cleaned_mask = (H['N_total'] == 0) if cleaned else None
        want_rv = 'pos' in self.load_pidrv or 'vel' in self.load_pidrv
        want_pid = 'pid' in self.load_pidrv
        offset = 0
        gathered_rv, gathered_pid, new_start = [], [], {}
        clean_afs = [AsdfFile(self.clean_rvpid_dir / f'cleaned_rvpid_{i:03d}.asdf') for i in self.superslab_inds] if cleaned else []
        for AB in self.load_AB:
            npout = H[f'npout{AB}'].astype(np.int64)
            if cleaned:                               # merged-away halos keep no particles of their own (:1052-1060)
                npout[cleaned_mask] = 0
                H[f'npout{AB}'] = npout.astype(np.uint32)
                nmerge = H[f'npout{AB}_merge'].astype(np.int64)
            else:
                nmerge = np.zeros(nh, dtype=np.int64)
            tot = npout + nmerge
            start = offset + np.concatenate([[0], np.cumsum(tot)])
            new_start[AB] = start
            offset = int(start[-1])
            for i, ind in enumerate(self.superslab_inds):
                sl = slice(file_off[i], file_off[i + 1])
                rs, rl = H[f'npstart{AB}'][sl].astype(np.int64), npout[sl]
                cs = H[f'npstart{AB}_merge'][sl].astype(np.int64) if cleaned else None
                cl = nmerge[sl]
                # read index of every output particle: halo by halo, the original particles, then the ones merged in; the
                # cleaned file's rows sit behind the slab's own in one concatenated array
                for kind, stash in (('rv', gathered_rv), ('pid', gathered_pid)):
                    if (kind == 'rv' and not want_rv) or (kind == 'pid' and not want_pid):
                        continue
                    col = {'rv': 'rvint', 'pid': 'packedpid'}[kind]
                    a = AsdfFile(Path(self.groupdir) / f'halo_{kind}_{AB}' / f'halo_{kind}_{AB}_{ind:03d}.asdf').array(col)
                    if cleaned:
                        c = clean_afs[i].array(f'{col}_{AB}')
                        src = np.concatenate([a, c.astype(a.dtype, copy=False)]) if len(c) else a
                    else:
                        src = a
                    idx = _zipper_index(rs, rl, cs, cl, len(a))
                    stash.append(src[idx])
        n_sub = offset
        if want_rv:
            rv = np.concatenate(gathered_rv) if gathered_rv else np.empty((0, 3), dtype=np.int32)
            assert len(rv) == n_sub
            pos, vel = bitpacked.unpack_rvint(rv, self.header['BoxSize'])
            if 'pos' in self.load_pidrv:
                self.subsamples['pos'] = pos
            if 'vel' in self.load_pidrv:
                self.subsamples['vel'] = vel
        if want_pid:
            packed = np.concatenate(gathered_pid) if gathered_pid else np.empty(0, dtype=np.uint64)
            assert len(packed) == n_sub
            which = unpack_bits if unpack_bits not in (True, False) else (bitpacked.PID_FIELDS if unpack_bits else ['pid'])
            if isinstance(which, str):
                which = [which]
            self.subsamples.update(bitpacked.unpack_pids(packed, box=self.header['BoxSize'], ppd=self.header['ppd'],
                                                         **{f: True for f in which}))
        for AB in self.load_AB:                       # the new indexing replaces the files' (:1346-1370)
            H[f'npstart{AB}'] = new_start[AB][:-1].astype(np.uint64)
            H[f'npout{AB}'] = np.diff(new_start[AB]).astype(np.uint32)
            if cleaned:
                H.remove_column(f'npstart{AB}_merge')
                H.remove_column(f'npout{AB}_merge')

    def __repr__(self):
        return (f'CompaSO Halo Catalog\n====================\n{self.header.get("SimName")} @ z={self.header.get("Redshift", 0):.5g}\n'
                f'{len(self.halos)} halos, {len(self.halos.colnames)} fields; {len(self.subsamples)} subsample particles')


def _zipper_index(rs, rl, cs, cl, nslab):
    """gather index into [slab rows ‖ cleaned rows]: for every halo its rl original rows from rs, then its cl rows from cs"""
    tot = rl + (cl if cl is not None else 0)
    n = int(tot.sum())
    if n == 0:
        return np.empty(0, dtype=np.int64)
    wstart = np.concatenate([[0], np.cumsum(tot)[:-1]])
    h = np.repeat(np.arange(len(rl)), tot)               # halo of every output row
    k = np.arange(n) - wstart[h]                         # rank inside the halo
    orig = k < rl[h]
    idx = np.where(orig, rs[h] + k, 0)
    if cs is not None:
        idx = np.where(orig, idx, nslab + cs[h] + (k - rl[h]))
    return idx.astype(np.int64)
