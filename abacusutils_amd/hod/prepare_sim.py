"""Subsample preparation on the MI355X: the data-parallel core of the reference's `prepare_sim.prepare_slab`
(abacusnbody/hod/prepare_sim.py:296-1052), between "the CompaSO loader handed over the slab's `halos` and `parts` tables" and
"the two HDF5 datasets are written".

What runs where
  device   subsample_halos + halo mask + particle targets (`abacus_prepare_halo_factors`); the per-halo particle selection,
           new halo offsets, kept-particle list, Np and the five satellite rank columns (`abacus_prepare_particles`); the
           concentration / shear / (light-cone) environment ranks per mass bin (`abacus_fenv_rank`); the environment masses
           (`do_Menv_from_tree`).
  host     drawing the random numbers and gathering the kept rows in the `rng='numpy'` form; file I/O (optional, needs h5py).
           With `rng=<seed>` the whole slab goes through HBM once (`abacus_prepare_slab`: inputs uploaded once or used in place as
           device arrays, kept rows gathered on the device, every output column copied out once into page-locked memory).

Random numbers.  The reference consumes NumPy's global legacy generator in a fixed order (:349-350 seed, :449 halo mask,
one `np.random.choice(replace=False)` per kept halo :163/:172, :984-996 halo randoms, :1029 particle randoms).
`rng='numpy'` reproduces exactly that - including the serial per-halo `choice` loop, which stays on the host - so a run
seeded like the reference agrees with it value for value (tests/test_prepare_gpu.py against the reference's own prepare_slab).
`rng=<int seed>` is the scalable form: every draw - the halo mask, the per-halo selection (counter-based Philox keys, no
per-halo serial work), the random columns of both tables (abacus_prepare_randoms) - is made on the device as a function of
(seed, global halo / particle index); same distributions and dtypes, another stream, restated by oracle/prepare_oracle.py.

Light cones (`halo_lc=True`): the environment of halos within `rad_outer` of the survey boundary is corrected by the
fraction of their annulus that lies inside it, measured with randoms (:474-616).  `lightcone_environment` draws the randoms
on the host from `default_rng(seed)` in the reference's order (so a run seeded like the reference counts the same points)
and counts them around the edge halos on the device (`abacus_menv` with unit-mass randoms; the reference queries a KD-tree).
"""
import ctypes as C

import numpy as np

from .. import _lib
from .menv import do_Menv_from_tree

__all__ = ['subsample_halos', 'prepare_slab_arrays', 'slab_environment', 'lightcone_environment', 'lightcone_edge_norm',
           'rank_in_mass_bins', 'reference_seed', 'save_subsample']

NBINS = 100          # mass bins of the rank columns (:454)
RANK_COLUMNS = ('ranks', 'ranksv', 'ranksp', 'ranksr', 'ranksc')


def reference_seed(newseed, i):
    """seed NumPy's global generator like prepare_slab does for slab i (:349-351); returns the light-cone randoms seed"""
    seeder = np.random.default_rng(newseed + i)
    np.random.seed(seeder.integers(0, 2**32 - 1))
    return seeder.integers(0, 2**32 - 1)


def _halo_factors(N, Mpart, MT, u=None, pnum=None):
    N = np.ascontiguousarray(N, dtype=np.uint32)
    n = len(N)
    p = np.empty(n, dtype=np.float64)
    mask = np.empty(n, dtype=np.uint8) if u is not None else None
    nt = np.empty(n, dtype=np.int32) if pnum is not None else None
    u8 = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
    pn = None if pnum is None else np.ascontiguousarray(pnum, dtype=np.int64)
    _lib.check(_lib.lib().abacus_prepare_halo_factors(_lib.ptr(N), C.c_int64(n), C.c_double(Mpart), int(bool(MT)), _lib.ptr(u8),
                                                      _lib.ptr(pn), _lib.ptr(p), _lib.ptr(mask), _lib.ptr(nt)))
    return p, (None if mask is None else mask.astype(bool)), nt


def subsample_halos(m, MT, Mpart=None):
    """fraction of halos kept as a function of mass (:83-108), evaluated on the device.  `m` are masses; halos are counted in
    whole particles, so pass `Mpart` for exact agreement with `halos['N'] * Mpart`; without it the masses are used as given"""
    m = np.ascontiguousarray(m, dtype=np.float64)
    if Mpart is None:            # the reference's signature: the formula on the float64 masses themselves
        p = np.empty(len(m), dtype=np.float64)
        _lib.check(_lib.lib().abacus_prepare_halo_factors_mass(_lib.ptr(m), C.c_int64(len(m)), int(bool(MT)), _lib.ptr(p)))
        return p
    return _halo_factors(np.rint(m / Mpart).astype(np.uint32), float(Mpart), MT)[0]


def rank_in_mass_bins(values, masses, mbins):
    """per mass bin (strictly inside both edges), the rank of `values` rescaled to [-0.5, 0.5]; halos alone in a bin or on an
    edge get 0 (deltac_rank :762-773, shear_rank :776-796, calc_fenv_opt :283-293): abacus_fenv_rank on the device"""
    v = np.ascontiguousarray(values, dtype=np.float64)
    m = np.ascontiguousarray(masses, dtype=np.float64)
    e = np.ascontiguousarray(mbins, dtype=np.float64)
    out = np.zeros(len(v), dtype=np.float64)
    if len(v):
        _lib.check(_lib.lib().abacus_fenv_rank(_lib.ptr(v), _lib.ptr(m), C.c_int64(len(v)), _lib.ptr(e), len(e), _lib.ptr(out)))
    return out


def _periodic_dx(x, x0, Lbox):
    return ((x - x0 + 0.5 * Lbox) % Lbox) - 0.5 * Lbox


def slab_environment(i, central, neighbours, numslabs, Lbox, Mpart, rad_outer=10, mcut=1e11):
    """raw environment masses of slab i's halos with the halos of the neighbouring slabs within `rad_outer` of its x-range
    as padding (the periodic-box branch, :622-745).  central / neighbours: halo tables (dicts with x_L2com, N, r98_L2com, id);
    `neighbours` lists the slabs (i - d) % numslabs and (i + d) % numslabs for d = 1 .. ceil(rad_outer / slab width), in any
    order.  Returns (id, mass, Menv) of the central halos - the content of the reference's env sidecar file (:748-756)."""
    cpos = np.asarray(central['x_L2com'])
    cmass = central['N'] * Mpart
    cid = np.asarray(central['id']).astype(np.int64)
    if len(np.unique(cid)) != len(cid):
        raise RuntimeError(f'Duplicate halo IDs found inside central slab {i}.')
    dx_slab = Lbox / numslabs
    x_center = -0.5 * Lbox + (i + 0.5) * dx_slab                       # unwrap_x_for_slab (:68-72)
    xu = x_center + _periodic_dx(cpos[:, 0], x_center, Lbox)
    edges = (xu.min(), xu.max())
    pos, mass, rvir, ids = [cpos], [cmass], [np.asarray(central['r98_L2com'])], [cid]
    for nb in neighbours:
        x = np.asarray(nb['x_L2com'])[:, 0]
        near = (np.abs(_periodic_dx(x, edges[0], Lbox)) <= rad_outer) | (np.abs(_periodic_dx(x, edges[1], Lbox)) <= rad_outer)
        if near.any():
            pos.append(np.asarray(nb['x_L2com'])[near])
            mass.append((nb['N'] * Mpart)[near])
            rvir.append(np.asarray(nb['r98_L2com'])[near])
            ids.append(np.asarray(nb['id']).astype(np.int64)[near])
    pos, mass, rvir, ids = np.concatenate(pos, axis=0), np.concatenate(mass), np.concatenate(rvir), np.concatenate(ids)
    _, first = np.unique(ids, return_index=True)                        # a halo reached through both edges counts once (:712-718)
    keep = np.sort(first)
    Menv = do_Menv_from_tree(pos[keep], mass[keep], r_inner=rvir[keep], r_outer=rad_outer, halo_lc=False, Lbox=Lbox, mcut=mcut)
    return cid, cmass, Menv[:len(cid)]


def _rows(a, idx):
    """a[idx] along the first axis; a 2-D column is gathered as one item of row size (NumPy's fancy indexing of (n, 3) arrays
    walks the elements: 2 - 3 times slower on the ~1e6-row tables of a slab)"""
    a = np.asarray(a)
    if a.ndim != 2 or not a.flags.c_contiguous or a.dtype.hasobject:
        return a[idx]
    v = a.view(np.dtype((np.void, a.dtype.itemsize * a.shape[1]))).ravel()
    return np.take(v, idx).view(a.dtype).reshape(-1, a.shape[1])


_POOL = None


def _rows_many(jobs):
    """{name: (array, index)} -> {name: array[index]}: the kept rows of the ~30 columns of a slab's two tables, gathered by a
    few threads (np.take releases the GIL; one thread walks a 1e6-row column in ~1 ms, the columns are independent)"""
    global _POOL
    jobs = list(jobs.items())
    if sum(len(idx) for _, (_, idx) in jobs) < 200_000:
        return {k: _rows(a, idx) for k, (a, idx) in jobs}
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)))
    return dict(zip([k for k, _ in jobs], _POOL.map(lambda job: _rows(*job[1]), jobs)))


LC_OFFSET = 10.0     # the light-cone catalogues stop this far inside the box faces (:481)


def _lightcone_cuboids(Lbox, offset, origins, chi_max):
    """(lo, hi) corners, relative to the observer, of the boxes a light-cone shell of outer radius chi_max can reach (:230-262):
    the box around 0 with `offset` shaved off every face an observer in its corner looks through - all six for a single
    observer in the centre - and, for the three-origin geometry once the shell reaches beyond the first box, its two copies."""
    half = Lbox / 2.0
    single = origins.shape[0] == 1
    lo = np.array([-half + offset, -half + offset, -half + offset])
    hi = np.array([half - offset, half - offset if single else half, half - offset if single else half])
    shifts = [np.array([0.0, 0.0, 0.0])]
    if not single:
        if origins.shape[0] != 3 or not (np.all(origins[1] + np.array([0.0, 0.0, Lbox]) == origins[0])
                                         and np.all(origins[2] + np.array([0.0, Lbox, 0.0]) == origins[0])):
            raise ValueError('light-cone origins: one observer, or three with origins[1] = origins[0] - (0, 0, L) and '
                             'origins[2] = origins[0] - (0, L, 0)')
        if chi_max >= (Lbox - offset):
            shifts += [np.array([0.0, 0.0, Lbox]), np.array([0.0, Lbox, 0.0])]
    return [((c - origins[0]) + lo, (c - origins[0]) + hi) for c in shifts]


def _lightcone_randoms(n, chi_min, chi_max, Lbox, offset, origins, rng):
    """n uniform points of the shell [chi_min, chi_max) around origins[0] - the octant of positive directions when there are
    three origins - cut to the boxes (gen_rand, :200-278: three `rng.random(n)` in this order: cos(theta), phi, radius).
    Returns (positions (m, 3) float64, distances (m,))."""
    if origins.shape[0] > 1:
        cost = rng.random(n)
        phi = rng.random(n) * np.pi / 2.0
    else:
        cost = rng.random(n) * 2.0 - 1.0
        phi = rng.random(n) * 2.0 * np.pi
    theta = np.arccos(cost)
    st = np.sin(theta)
    chi = rng.random(n) * (chi_max - chi_min) + chi_min
    xyz = [st * np.cos(phi) * chi, st * np.sin(phi) * chi, np.cos(theta) * chi]
    inside = np.zeros(n, dtype=bool)
    for lo, hi in _lightcone_cuboids(Lbox, offset, origins, chi_max):
        m = np.ones(n, dtype=bool)
        for d in range(3):
            m &= (xyz[d] > lo[d]) & (xyz[d] <= hi[d])
        inside |= m
    pos = np.vstack([c[inside] for c in xyz]).T
    pos += origins[0]
    return pos, chi[inside]


def _lightcone_interior(pos, dist, Lbox, offset, origins, pad, r_min, r_max):
    """points at least `pad` inside every boundary of the light-cone volume - the shaved box faces and the two shell radii
    (:483-511, :516-529).  Python-float bounds against float32 columns compare in float32, as in the reference."""
    half = Lbox / 2.0
    lo = -(half - offset - pad)
    hi_x = half - offset - pad
    hi_yz = half - offset - pad if origins.shape[0] == 1 else 3.0 / 2 * Lbox - pad
    return ((lo <= pos[:, 0]) & (hi_x >= pos[:, 0]) & (lo <= pos[:, 1]) & (hi_yz >= pos[:, 1]) & (lo <= pos[:, 2])
            & (hi_yz >= pos[:, 2]) & (r_min + pad <= dist) & (r_max - pad >= dist))


def lightcone_edge_norm(pos, r98, Lbox, origins, seed, rad_outer=10, rand_final=10):
    """The edge halos of a light-cone catalogue and the completeness of their environment annulus (:474-597).

    pos (n, 3), r98 (n,): halo positions and inner radii as the catalogue stores them (float32); origins: the header's
    `LightConeOrigins`; seed: the `halo_lc_randoms_seed` of the slab (`reference_seed`).  Returns (index_bounds, rand_norm):
    the halos closer than rad_outer to a boundary, and for each the number of randoms counted in [r98, rad_outer] over the
    number expected in an unbounded field (~1 inside, < 1 at the boundary).  Randoms: `default_rng(seed)`, len(pos) per
    round, until ten times as many as edge halos fell into the boundary layer of width 2 rad_outer; counted on the device."""
    pos = np.asarray(pos)
    r98 = np.asarray(r98)
    Lbox = float(Lbox)
    origins = np.asarray(origins, dtype=np.float64).reshape(-1, 3)
    dist = np.sqrt(np.sum((pos - origins[0]) ** 2.0, axis=1))
    r_min, r_max = dist.min(), dist.max()
    edge = np.flatnonzero(~_lightcone_interior(pos, dist, Lbox, LC_OFFSET, origins, rad_outer, r_min, r_max))
    norm = np.zeros(len(edge))
    if len(edge) == 0:
        return edge, norm
    n = pos.shape[0]
    shell = 4.0 / 3.0 * np.pi * (r_max**3 - r_min**3) if origins.shape[0] == 1 else 4.0 / 3.0 / 8.0 * np.pi * (r_max**3 - r_min**3)
    density = n / shell                          # randoms per volume and round
    rng = np.random.default_rng(seed)
    centres = np.ascontiguousarray(pos[edge], dtype=np.float64)
    inner = np.asarray(r98[edge], dtype=np.float64)
    count = rounds = 0
    while count < len(edge) * rand_final:
        rpos, rdist = _lightcone_randoms(n, r_min, r_max, Lbox, LC_OFFSET, origins, rng)
        rpos = rpos[~_lightcone_interior(rpos, rdist, Lbox, LC_OFFSET, origins, 2.0 * rad_outer, r_min, r_max)]
        if len(rpos):
            # randoms of mass 1 around centres of mass 0: every point is a centre (mcut < 0), only randoms weigh
            both = np.concatenate([centres, rpos])
            mass = np.concatenate([np.zeros(len(edge)), np.ones(len(rpos))])
            ri = np.concatenate([inner, np.zeros(len(rpos))])
            norm += do_Menv_from_tree(both, mass, r_inner=ri, r_outer=float(rad_outer), halo_lc=True, Lbox=Lbox, mcut=-1.0)[:len(edge)]
        rounds += 1
        count += len(rpos)
        if rounds >= 1000 and count == 0:      # the reference would loop for ever: no random ever falls into the boundary layer
            raise RuntimeError('lightcone_edge_norm: no randoms inside the light-cone volume near its boundary - do the '
                               'origins and Lbox describe the catalogue?')
    density *= rounds
    norm /= (rad_outer**3.0 - r98[edge] ** 3.0) * 4.0 / 3.0 * np.pi * density
    return edge, norm


def lightcone_environment(pos, masses, r98, Lbox, origins, seed, rad_outer=10, mcut=1e11):
    """Menv of a light-cone slab with the edge correction (:474-616): the open-geometry `do_Menv_from_tree`, divided for the
    edge halos by the completeness of their annulus (0 where no random fell into it)."""
    edge, norm = lightcone_edge_norm(pos, r98, Lbox, origins, seed, rad_outer=rad_outer)
    Menv = do_Menv_from_tree(pos, masses, r_inner=r98, r_outer=rad_outer, halo_lc=True, Lbox=Lbox, mcut=mcut)
    if len(edge):
        empty = norm == 0.0
        norm[empty] = 1.0
        corrected = Menv[edge] / norm
        corrected[empty] = 0.0
        Menv[edge] = corrected
    return Menv


def _targets_host(masses, pnum, MT):
    """submask_particles' target count (:152-174) in the reference's own floating-point expressions (the host loop of
    rng='numpy' must draw exactly what the reference draws)"""
    out = np.zeros(len(masses), dtype=np.int64)
    for j in np.nonzero(pnum > 0)[0]:
        m_in, n_in = masses[j], int(pnum[j])
        x = np.log10(m_in)
        if MT:
            if m_in < 1e11:
                continue
            out[j] = min(min(n_in, int(1 + 1.5 * 10 ** (x - 12.5))), 100)
        elif not 10 ** x < 1e12:
            out[j] = min(n_in, int(1 + 1.5 * 10 ** (x - 13)))
    return out


class _SlabArgs(C.Structure):      # struct abacus_prepare_slab_args (include/abacus_hip.h)
    _fields_ = ([('nh', C.c_int64), ('npart', C.c_int64)]
                + [(k, C.c_void_p) for k in ('N', 'x', 'v', 'r25', 'r90', 'r98', 'sigmav', 'npstartA', 'npoutA', 'id', 'pos', 'vel',
                                             'fenv_rank', 'shear_rank', 'mbins')]
                + [('n_edges', C.c_int32), ('MT', C.c_int32), ('want_ranks', C.c_int32), ('pad_', C.c_int32), ('Mpart', C.c_double),
                   ('h', C.c_double), ('seed', C.c_uint64), ('halo_index0', C.c_int64), ('part_index0', C.c_int64)])


# columns of the two tables in the order of the library's column arrays (ABACUS_PREP_H_* / ABACUS_PREP_P_*): name, dtype, row shape
_SLAB_HALO_IN = (('N', np.uint32, ()), ('x_L2com', np.float32, (3,)), ('v_L2com', np.float32, (3,)), ('r25_L2com', np.float32, ()),
                 ('r90_L2com', np.float32, ()), ('r98_L2com', np.float32, ()), ('npstartA', np.int64, ()), ('npoutA', np.int64, ()),
                 ('id', None, ()), ('sigmav3d_L2com', np.float32, ()))
_SLAB_HALO_OUT = (('N', np.uint32, ()), ('x_L2com', np.float32, (3,)), ('v_L2com', np.float32, (3,)), ('r25_L2com', np.float32, ()),
                  ('r90_L2com', np.float32, ()), ('r98_L2com', np.float32, ()), ('npstartA', np.float64, ()), ('npoutA', np.float64, ()),
                  ('id', None, ()), ('sigmav3d_L2com', np.float32, ()), ('mask_subsample', np.bool_, ()), ('multi_halos', np.float64, ()),
                  ('fenv_rank', np.float64, ()), ('deltac_rank', np.float64, ()), ('shear_rank', np.float64, ()), ('randoms', np.float64, ()),
                  ('randoms_exp', np.float64, (3,)), ('randoms_gaus_vrms', np.float64, (3,)))
_SLAB_PART_OUT = ((('pos', np.float32, (3,)), ('vel', np.float32, (3,))) + tuple((k, np.float64, ()) for k in RANK_COLUMNS)
                  + (('downsample_halo', np.float64, ()), ('halo_vel', np.float64, (3,)), ('halo_mass', np.float64, ()), ('Np', np.float64, ()),
                     ('halo_id', np.int64, ()), ('randoms', np.float64, ()), ('halo_deltac', np.float64, ()), ('halo_fenv', np.float64, ()),
                     ('halo_shear', np.float64, ())))


def _slab_col(a, dtype, keep):
    """pointer of one input column: a device array is used where it is, anything else as a contiguous NumPy array of `dtype`"""
    if isinstance(a, _lib.DeviceArray):
        if dtype is not None and a.dtype != np.dtype(dtype):
            raise TypeError(f'device column of dtype {a.dtype}, expected {np.dtype(dtype)}')
        keep.append(a)
        return C.c_void_p(a.ptr.value)
    a = np.ascontiguousarray(a) if dtype is None else np.ascontiguousarray(a, dtype=dtype)
    keep.append(a)
    return C.c_void_p(a.ctypes.data)


def _prepare_slab_device(halos, parts, Mpart, h, MT, want_ranks, want_AB, fenv_rank, shear_rank, mbins, seed, part_index0, halo_index0):
    """prepare_slab_arrays with rng = <seed> as ONE pass through HBM (abacus_prepare_slab / abacus_prepare_slab_fetch): every input
    column is uploaded once (or used in place when it is a `_lib.DeviceArray` - what the reader's unpack kernels produce), the kept
    rows of both tables are gathered on the device and every output column is copied out once, into page-locked memory"""
    keep = []
    id_col = halos['id']
    id_dtype = id_col.dtype if isinstance(id_col, _lib.DeviceArray) else np.asarray(id_col).dtype
    if np.dtype(id_dtype).itemsize != 8:
        raise TypeError(f'halo ids of dtype {id_dtype}: 8-byte integers expected')
    nh = int(halos['N'].shape[0])
    npart = int(parts['pos'].shape[0])
    a = _SlabArgs()
    a.nh, a.npart = nh, npart
    for field, (name, dt, _) in zip(('N', 'x', 'v', 'r25', 'r90', 'r98', 'npstartA', 'npoutA', 'id', 'sigmav'), _SLAB_HALO_IN):
        setattr(a, field, _slab_col(halos[name], dt, keep))
    a.pos, a.vel = _slab_col(parts['pos'], np.float32, keep), _slab_col(parts['vel'], np.float32, keep)
    a.fenv_rank = None if fenv_rank is None else _slab_col(fenv_rank, np.float64, keep)
    a.shear_rank = None if shear_rank is None else _slab_col(shear_rank, np.float64, keep)
    if want_AB:
        a.mbins, a.n_edges = _slab_col(mbins, np.float64, keep), len(mbins)
    else:
        a.mbins, a.n_edges = None, 0
    a.MT, a.want_ranks, a.Mpart, a.h = int(bool(MT)), int(bool(want_ranks)), float(Mpart), float(h)
    a.seed, a.halo_index0, a.part_index0 = int(seed), int(halo_index0), int(part_index0)
    nk, ns = C.c_int64(0), C.c_int64(0)
    mask8 = np.empty(nh, dtype=np.uint8)
    L = _lib.lib()
    _lib.check(L.abacus_prepare_slab(C.byref(a), C.byref(nk), C.byref(ns), _lib.ptr(mask8)))
    nk, ns = int(nk.value), int(ns.value)

    def table(spec, n, skip=()):
        cols, ptrs = {}, []
        for name, dt, tail in spec:
            if name in skip:
                ptrs.append(None)
                continue
            arr = _lib.pinned_empty((n,) + tail, id_dtype if dt is None else dt)
            cols[name] = arr
            ptrs.append(arr.ctypes.data if n else None)
        return cols, (C.c_void_p * len(ptrs))(*ptrs)

    Hk, hp = table(_SLAB_HALO_OUT, nk)
    P, pp = table(_SLAB_PART_OUT, ns, skip=() if want_ranks else RANK_COLUMNS)
    _lib.check(L.abacus_prepare_slab_fetch(hp, pp))
    mask = mask8.astype(bool)
    # the caller's key order (columns the library does not know are gathered here, like before)
    kept = None
    out = {}
    for k, v in halos.items():
        if k in Hk:
            out[k] = Hk[k]
        else:
            if kept is None:
                kept = np.flatnonzero(mask)
            out[k] = _rows(v.get() if isinstance(v, _lib.DeviceArray) else v, kept)
    for name, _, _ in _SLAB_HALO_OUT[10:]:
        out[name] = Hk[name]
    return out, P, mask


def prepare_slab_arrays(halos, parts, Mpart, h, MT, want_ranks=False, want_AB=True, Menv=None, shearmark=None, Lbox=None,
                        mcut=1e11, halo_lc=False, rng='numpy', part_index0=0, halo_index0=0, origins=None, lc_seed=None,
                        rad_outer=10):
    """halos: dict of columns N, x_L2com, v_L2com, r25_L2com, r90_L2com, r98_L2com, npstartA, npoutA, id, sigmav3d_L2com (what
    CompaSOHaloCatalog loads for prepare_slab, :404-425); parts: dict with pos, vel of the slab's subsample-A particles in halo
    order.  Returns (halo table of the kept halos, particle table of the kept particles, mask over the input halos): dicts with
    the field names and dtypes of the reference's 'halos' / 'particles' datasets (:1001-1045).
    rng: 'numpy' consumes NumPy's global legacy generator in the reference's order (seed it like the reference and the tables
    come out value for value; light cones: pass `origins` (header LightConeOrigins) and `lc_seed`, the second value of
    `reference_seed`); an integer seeds the device's counter-based generator - every draw is then a function of (seed,
    global halo / particle index = halo_index0 / part_index0 + row), so slabs prepared on different GPUs fit together."""
    nh = len(halos['N'])
    numpy_mode = isinstance(rng, str)
    if numpy_mode and rng != 'numpy':
        raise ValueError("rng must be 'numpy' (the reference's global generator) or an integer seed")
    seed = 0 if numpy_mode else int(rng) & (2**64 - 1)
    def _fits(col, dt):      # the device path returns the library's dtypes: only for inputs that already carry them
        d = col.dtype if isinstance(col, _lib.DeviceArray) else np.asarray(col).dtype
        return (d.kind in 'iu' and d.itemsize == 8) if dt is None else d == np.dtype(dt)

    if (not numpy_mode and not _lib.get_option('prep_columnwise') and all(k in halos and _fits(halos[k], dt) for k, dt, _ in _SLAB_HALO_IN)
            and _fits(parts['pos'], np.float32) and _fits(parts['vel'], np.float32)):
        # every draw is made on the device: the slab goes through HBM once (abacus_prepare_slab); the per-halo rank columns that need
        # host-side inputs (light-cone environment, shear) are computed as before and handed over
        mbins = np.logspace(np.log10(mcut), 15.5, NBINS + 1)
        fenv = shear = None
        def hcol(k):         # a host copy of a column the host-side rank inputs need
            return halos[k].get() if isinstance(halos[k], _lib.DeviceArray) else np.asarray(halos[k])
        hN = hcol('N')
        if want_AB and halo_lc:
            if Menv is None:
                if origins is None or lc_seed is None or Lbox is None:
                    raise ValueError('halo_lc with want_AB needs the environment masses (Menv), or Lbox, the light-cone origins '
                                     'and the seed of the randoms (lc_seed, see reference_seed) to work them out')
                Menv = lightcone_environment(hcol('x_L2com'), hN * Mpart, hcol('r98_L2com'), Lbox, origins, lc_seed,
                                             rad_outer=rad_outer, mcut=mcut)
            fenv = rank_in_mass_bins(Menv, hN * Mpart, mbins)
        if shearmark is not None:
            ndim = len(shearmark)
            g = (hcol('x_L2com') / (Lbox / ndim)).astype(int) % ndim
            shear = rank_in_mass_bins(shearmark[g[:, 0], g[:, 1], g[:, 2]], hN * Mpart, mbins)
        return _prepare_slab_device(halos, parts, Mpart, h, MT, want_ranks, want_AB, fenv, shear, mbins, seed, part_index0, halo_index0)
    N = np.ascontiguousarray(halos['N'], dtype=np.uint32)
    masses = halos['N'] * Mpart
    pstart = np.ascontiguousarray(halos['npstartA'], dtype=np.int64)
    pnum = np.ascontiguousarray(halos['npoutA'], dtype=np.int64)
    if numpy_mode:
        u = np.random.random(nh)                                                     # (:449)
    else:
        u = np.empty(nh)
        _lib.check(_lib.lib().abacus_prepare_randoms(C.c_int64(nh), None, C.c_int64(halo_index0), C.c_uint64(seed), 6, None,
                                                     _lib.ptr(u), None, None))
    p_halos, mask_halos, ntarget = _halo_factors(N, Mpart, MT, u=u, pnum=pnum)
    H = dict(halos)
    H['mask_subsample'] = mask_halos
    H['multi_halos'] = 1.0 / p_halos
    mbins = np.logspace(np.log10(mcut), 15.5, NBINS + 1)
    zeros = np.zeros(nh)
    if want_AB:
        if halo_lc:
            if Menv is None:
                if origins is None or lc_seed is None or Lbox is None:
                    raise ValueError('halo_lc with want_AB needs the environment masses (Menv), or Lbox, the light-cone origins '
                                     'and the seed of the randoms (lc_seed, see reference_seed) to work them out')
                Menv = lightcone_environment(halos['x_L2com'], masses, halos['r98_L2com'], Lbox, origins, lc_seed,
                                             rad_outer=rad_outer, mcut=mcut)                # (:474-616)
            H['fenv_rank'] = rank_in_mass_bins(Menv, masses, mbins)                  # calc_fenv_opt (:618)
        else:
            H['fenv_rank'] = zeros.copy()         # ranked over the whole box later, by AbacusHOD.staging() (:758-759)
        H['deltac_rank'] = rank_in_mass_bins(halos['r98_L2com'] / halos['r25_L2com'], masses, mbins)
    else:
        H['fenv_rank'], H['deltac_rank'] = zeros.copy(), zeros.copy()
    if shearmark is not None:                                                        # (:776-796)
        ndim = len(shearmark)
        g = (np.asarray(halos['x_L2com']) / (Lbox / ndim)).astype(int) % ndim
        H['shear_rank'] = rank_in_mass_bins(shearmark[g[:, 0], g[:, 1], g[:, 2]], masses, mbins)
    else:
        H['shear_rank'] = zeros.copy()

    pos = np.ascontiguousarray(parts['pos'], dtype=np.float32)
    vel = np.ascontiguousarray(parts['vel'], dtype=np.float32)
    npart = len(pos)
    submask = None
    if numpy_mode:      # the reference's serial loop: one `choice` per kept halo with particles above the mass floor
        tg = _targets_host(masses, np.where(mask_halos, pnum, 0), MT)
        submask = np.zeros(npart, dtype=np.uint8)
        for j in np.nonzero(tg > 0)[0]:
            submask[pstart[j] + np.random.choice(int(pnum[j]), int(tg[j]), replace=False)] = 1

    hmask8 = np.ascontiguousarray(mask_halos, dtype=np.uint8)
    hpos = np.ascontiguousarray(halos['x_L2com'], dtype=np.float32)
    hvel = np.ascontiguousarray(halos['v_L2com'], dtype=np.float32)
    r25 = np.ascontiguousarray(halos['r25_L2com'], dtype=np.float32)
    r98 = np.ascontiguousarray(halos['r98_L2com'], dtype=np.float32)
    pstart_new, pnum_new = np.empty(nh), np.empty(nh)
    nsel = C.c_int64(0)
    L = _lib.lib()

    def call(sub_in, cap, outs, sub_out, ranks=False):
        _lib.check(L.abacus_prepare_particles(
            C.c_int64(nh), _lib.ptr(hmask8), _lib.ptr(pstart), _lib.ptr(pnum), _lib.ptr(N), _lib.ptr(hpos), _lib.ptr(hvel),
            _lib.ptr(r25), _lib.ptr(r98), C.c_int64(npart), _lib.ptr(pos), _lib.ptr(vel), _lib.ptr(sub_in),
            None if sub_in is not None else _lib.ptr(ntarget), C.c_uint64(seed), C.c_int64(part_index0), C.c_double(Mpart),
            C.c_double(h), int(bool(ranks)), _lib.ptr(pstart_new), _lib.ptr(pnum_new), C.byref(nsel), C.c_int64(cap),
            *[_lib.ptr(o) for o in outs], _lib.ptr(sub_out)))

    # The number of kept particles is known before the call - the sum of the kept halos' targets (device draw) or of the host
    # draw's mask - so the outputs are allocated once and a single call selects and emits (a size query plus a second call
    # remains as the fallback should the bound ever be exceeded).
    cap = int(submask.sum()) if submask is not None else int(ntarget[mask_halos].sum(dtype=np.int64))

    def outputs(m):
        return (np.empty(m, dtype=np.int64), np.empty(m, dtype=np.int64), np.empty(m),
                [np.empty(m) for _ in RANK_COLUMNS] if want_ranks else [None] * 5)

    sel_idx, sel_host, sel_np, rk = outputs(cap)
    if cap:
        call(submask, cap, [sel_idx, sel_host, sel_np] + rk, None, ranks=want_ranks)
    else:
        call(submask, 0, [None] * 8, None)
    n = int(nsel.value)
    if n > cap:                                  # not expected: the bound above is exact for both kinds of draw
        if submask is None:
            submask = np.zeros(max(npart, 1), dtype=np.uint8)[:npart]
            call(None, 0, [None] * 8, submask)
        sel_idx, sel_host, sel_np, rk = outputs(n)
        call(submask, n, [sel_idx, sel_host, sel_np] + rk, None, ranks=want_ranks)
    elif n < cap:
        sel_idx, sel_host, sel_np = sel_idx[:n], sel_host[:n], sel_np[:n]
        rk = [r[:n] if r is not None else None for r in rk]
    H['npstartA'], H['npoutA'] = pstart_new, pnum_new
    kept = np.flatnonzero(mask_halos)
    if numpy_mode:                                                                   # (:984-996): drawn for every halo
        sig = np.repeat(halos['sigmav3d_L2com'], 3).reshape((-1, 3)) / np.sqrt(3)
        H['randoms'] = np.random.random(nh)
        H['randoms_exp'] = (np.random.randint(0, 2, size=(nh, 3)) * 2 - 1) * np.random.exponential(scale=sig, size=(nh, 3))
        H['randoms_gaus_vrms'] = np.random.normal(loc=0, scale=sig, size=(nh, 3))
    Hk = _rows_many({k: (v, kept) for k, v in H.items()})
    if not numpy_mode:                                                               # drawn on the device, kept halos only
        nk = len(kept)
        scale = np.asarray(halos['sigmav3d_L2com'])[kept] / np.sqrt(3)
        scale = np.ascontiguousarray(scale, dtype=np.float64)
        Hk['randoms'], Hk['randoms_exp'], Hk['randoms_gaus_vrms'] = np.empty(nk), np.empty((nk, 3)), np.empty((nk, 3))
        _lib.check(L.abacus_prepare_randoms(C.c_int64(nk), _lib.ptr(kept), C.c_int64(halo_index0), C.c_uint64(seed), 4,
                                            _lib.ptr(scale), _lib.ptr(Hk['randoms']), _lib.ptr(Hk['randoms_exp']),
                                            _lib.ptr(Hk['randoms_gaus_vrms'])))

    G = _rows_many({'pos': (pos, sel_idx), 'vel': (vel, sel_idx), 'downsample_halo': (p_halos, sel_host), 'halo_vel': (hvel, sel_host),
                    'halo_mass': (masses, sel_host), 'halo_id': (np.asarray(halos['id']), sel_host),
                    'halo_deltac': (H['deltac_rank'], sel_host), 'halo_fenv': (H['fenv_rank'], sel_host),
                    'halo_shear': (H['shear_rank'], sel_host)})
    P = {'pos': G['pos'], 'vel': G['vel']}
    if want_ranks:
        for name, col in zip(RANK_COLUMNS, rk):
            P[name] = col
    P['downsample_halo'] = G['downsample_halo']
    P['halo_vel'] = G['halo_vel'].astype(np.float64)
    P['halo_mass'] = G['halo_mass'].astype(np.float64)
    P['Np'] = sel_np
    P['halo_id'] = G['halo_id'].astype(np.int64)
    if numpy_mode:
        P['randoms'] = np.random.random(n)                                           # (:1029)
    else:
        P['randoms'] = np.empty(n)
        _lib.check(L.abacus_prepare_randoms(C.c_int64(n), _lib.ptr(sel_idx), C.c_int64(part_index0), C.c_uint64(seed), 5, None,
                                            _lib.ptr(P['randoms']), None, None))
    P['halo_deltac'], P['halo_fenv'], P['halo_shear'] = G['halo_deltac'], G['halo_fenv'], G['halo_shear']
    return Hk, P, mask_halos


def save_subsample(halo_table, particle_table, halo_fn, particle_fn):
    """the two HDF5 files of a slab as the reference writes them (:1001-1045: one compound dataset each); needs h5py"""
    try:
        import h5py
    except ImportError as e:
        raise ImportError('writing the prepare_sim HDF5 files needs h5py; the tables themselves can be handed to '
                          'AbacusHOD.from_prepared without touching the disk') from e

    def compound(tab):
        n = len(next(iter(tab.values())))
        dt = np.dtype([(k, v.dtype, v.shape[1:]) for k, v in tab.items()])
        out = np.empty(n, dtype=dt)
        for k, v in tab.items():
            out[k] = v
        return out

    for fn, name, tab in ((halo_fn, 'halos', halo_table), (particle_fn, 'particles', particle_table)):
        with h5py.File(fn, 'w') as f:
            f.create_dataset(name, data=compound(tab))


# ---- the reference's drivers: one slab from the CompaSO files, and the loop over a simulation's slabs ------------------------
def load_env_halos(slabname, cleaning, filter_func=None):
    """the four columns the padded environment needs from a neighbouring slab (:53-66)"""
    from ..data.compaso_halo_catalog import CompaSOHaloCatalog
    cat = CompaSOHaloCatalog(slabname, fields=['N', 'x_L2com', 'r98_L2com', 'id'], cleaned=cleaning, filter_func=filter_func)
    halos = cat.halos
    if cleaning:
        halos = halos[halos['N'] > 0]
    return halos


def _slab_paths(i, savedir, newseed, MT, want_ranks):
    """output names of slab i (:316-343)"""
    halos = f'{savedir}/halos_xcom_{i}_seed{newseed}_abacushod_oldfenv'
    parts = f'{savedir}/particles_xcom_{i}_seed{newseed}_abacushod_oldfenv'
    if MT:
        halos += '_MT'
        parts += '_MT'
    if want_ranks:
        parts += '_withranks'
    return halos + '_new.h5', parts + '_new.h5', f'{savedir}/env_xcom_{i}_abacushod_localenv_new.h5'


def prepare_slab(i, savedir, simdir, simname, z_mock, z_type, tracer_flags, MT, want_ranks, want_AB, want_shear, shearmark,
                 cleaning, newseed, halo_lc=False, nthread=1, overwrite=1, mcut=1e11, rad_outer=10, numslabs=None,
                 return_tables=False):
    """Subsample slab i of a simulation for the HOD (hod/prepare_sim.py:295-1052, the reference's signature): load the slab's
    CompaSO halos and subsample-A particles (abacusutils_amd.data.compaso_halo_catalog), run the subsampling on the device
    (`prepare_slab_arrays(rng='numpy')`: NumPy's global generator, seeded and consumed like the reference, so the tables come
    out value for value), write the reference's three HDF5 files.  Returns 0 like the reference; `return_tables=True`
    (extension) returns (halo table, particle table, env sidecar or None) and writes files only if h5py is importable."""
    import os

    from ..data.compaso_halo_catalog import CompaSOHaloCatalog
    fn_halos, fn_parts, fn_env = _slab_paths(i, savedir, newseed, MT, want_ranks)
    print('processing slab ', i)
    lc_seed = reference_seed(newseed, i)                                             # (:345-347)
    need_env_file = want_AB and (not halo_lc)
    if (not int(overwrite)) and os.path.exists(fn_halos) and os.path.exists(fn_parts) and ((not need_env_file) or os.path.exists(fn_env)):
        print('files exists, skipping ', i)
        return 0
    zdir = 'z' + str(z_mock).ljust(5, '0')
    if halo_lc:
        slabname = f'{simdir}/{simname}/{zdir}/lc_halo_info.asdf'
        id_key, pos_key, vel_key, N_key = 'index_halo', 'pos_interp', 'vel_interp', 'N_interp'
    else:
        slabname = f'{simdir}/{simname}/halos/{zdir}/halo_info/halo_info_{str(i).zfill(3)}.asdf'
        id_key, pos_key, vel_key, N_key = 'id', 'x_L2com', 'v_L2com', 'N'
    fields = [N_key, pos_key, vel_key, 'r90_L2com', 'r25_L2com', 'r98_L2com', 'npstartA', 'npoutA', id_key, 'sigmav3d_L2com']
    with_parts = z_type in ('primary', 'lightcone')
    if not with_parts:
        raise NotImplementedError('prepare_slab at a secondary redshift (no particle subsamples on disk): the reference itself '
                                  'goes on to use the particles it did not load (:804-806)')
    cat = CompaSOHaloCatalog(slabname, subsamples=dict(A=True, rv=True), fields=fields, cleaned=cleaning)
    assert halo_lc == cat.halo_lc
    halos = cat.halos
    if halo_lc:
        halos['id'], halos['x_L2com'], halos['v_L2com'], halos['N'] = halos[id_key], halos[pos_key], halos[vel_key], halos[N_key]
    parts = cat.subsamples
    if cleaning:
        # the subsample indexing of the halos that stay is untouched by the selection: npstartA still addresses `parts`
        halos = halos[halos['N'] > 0]
    header = cat.header
    Lbox, Mpart, h = header['BoxSizeHMpc'], header['ParticleMassHMsun'], header['H0'] / 100.0
    env = None
    if want_AB and not halo_lc:
        if numslabs is None:
            raise ValueError('prepare_slab needs numslabs for the padded env calculation.')
        dx_slab = Lbox / numslabs
        x_center = -0.5 * Lbox + (i + 0.5) * dx_slab
        xu = x_center + _periodic_dx(np.asarray(halos['x_L2com'])[:, 0], x_center, Lbox)
        neighbours = []
        for d in range(1, max(1, int(np.ceil(rad_outer / dx_slab))) + 1):            # (:651-704)
            for j, edge in (((i - d) % numslabs, xu.min()), ((i + d) % numslabs, xu.max())):
                name = f'{simdir}/{simname}/halos/{zdir}/halo_info/halo_info_{str(j).zfill(3)}.asdf'
                nb = load_env_halos(name, cleaning, filter_func=lambda t, e=edge: np.abs(_periodic_dx(t['x_L2com'][:, 0], e, Lbox)) <= rad_outer)
                if len(nb):
                    neighbours.append(nb)
        env = slab_environment(i, halos, neighbours, numslabs, Lbox, Mpart, rad_outer=rad_outer, mcut=mcut)
    origins = np.asarray(header['LightConeOrigins']).reshape(-1, 3) if halo_lc and 'LightConeOrigins' in header else None
    H, P, _ = prepare_slab_arrays(dict(halos), {'pos': parts['pos'], 'vel': parts['vel']}, Mpart, h, MT, want_ranks=want_ranks,
                                  want_AB=want_AB, shearmark=shearmark if want_shear else None, Lbox=Lbox, mcut=mcut,
                                  halo_lc=halo_lc, rng='numpy', origins=origins, lc_seed=lc_seed, rad_outer=rad_outer)
    try:
        import h5py  # noqa: F401
        have_h5 = True
    except ImportError:
        have_h5 = False
        if not return_tables:
            raise ImportError('prepare_slab writes HDF5 files like the reference and needs h5py (or pass return_tables=True and hand '
                              'the tables to AbacusHOD.from_prepared)') from None
    if have_h5:
        import h5py
        os.makedirs(savedir, exist_ok=True)
        if env is not None:                                                          # (:748-756)
            if os.path.exists(fn_env):
                os.remove(fn_env)
            with h5py.File(fn_env, 'w') as f:
                for k, v in zip(('id', 'mass', 'Menv'), env):
                    f.create_dataset(k, data=v)
        for fn in (fn_halos, fn_parts):
            if os.path.exists(fn):
                os.remove(fn)
        save_subsample(H, P, fn_halos, fn_parts)
    if return_tables:
        return H, P, env
    return 0


_PRIMARY_Z = [3.0, 2.5, 2.0, 1.7, 1.4, 1.1, 0.8, 0.5, 0.4, 0.3, 0.2, 0.1, 0.0]
_SECONDARY_Z = [0.15, 0.25, 0.35, 0.45, 0.575, 0.65, 0.725, 0.875, 0.95, 1.025, 1.175, 1.25, 1.325, 1.475, 1.55, 1.625, 1.85, 2.25,
                2.75, 3.0, 5.0, 8.0]


def main(path2config, params=None, alt_simname=None, alt_z=None, newseed=600, halo_lc=False, overwrite=1):
    """prepare_sim for every slab of the simulation a config names (hod/prepare_sim.py:1130-1291, the reference's signature).
    The slabs are prepared one after the other in this process (the reference spreads them over a process pool; a GPU process
    is not forked)."""
    import os
    from pathlib import Path

    import yaml
    print('compiling compaso halo catalogs into subsampled catalogs')
    config = yaml.safe_load(open(path2config))
    if params:
        config.update(params)
    if alt_simname:
        config['sim_params']['sim_name'] = alt_simname
    if alt_z:
        config['sim_params']['z_mock'] = alt_z
    simname, simdir = config['sim_params']['sim_name'], config['sim_params']['sim_dir']
    z_mock = float(config['sim_params']['z_mock'])
    savedir = config['sim_params']['subsample_dir'] + simname + '/z' + str(z_mock).ljust(5, '0')
    cleaning = config['sim_params']['cleaned_halos']
    if 'halo_lc' in config['sim_params']:
        halo_lc = config['sim_params']['halo_lc']
    if halo_lc:
        ztype = 'lightcone'
    elif z_mock in _PRIMARY_Z:
        ztype = 'primary'
    elif z_mock in _SECONDARY_Z:
        ztype = 'secondary'
    else:
        raise Exception('illegal redshift')
    if halo_lc:
        halo_info_fns = [str(Path(simdir) / Path(simname) / ('z%4.3f' % z_mock) / 'lc_halo_info.asdf')]
    else:
        search_path = Path(simdir) / Path(simname) / 'halos' / ('z%4.3f' % z_mock) / 'halo_info'
        halo_info_fns = list(sorted(search_path.glob('*.asdf')))
        if not halo_info_fns:
            raise ValueError(f'no halo info files found in {search_path}')
    numslabs = len(halo_info_fns)
    os.makedirs(savedir, exist_ok=True)
    tracer_flags = config['HOD_params']['tracer_flags']
    MT = bool(tracer_flags['ELG'] or tracer_flags['QSO'])
    want_ranks = config['HOD_params'].get('want_ranks', False)
    want_AB = config['HOD_params'].get('want_AB', False)
    want_shear = config['HOD_params'].get('want_shear', False)
    shearmark = None
    if want_shear:
        if (not ztype == 'primary') and (not halo_lc):
            raise Exception('redshift does not have particle data, cant compute shear')
        Ndim, Rsm = config['HOD_params'].get('shear_N', 1000), config['HOD_params'].get('shear_R', 2)
        partdown = config['HOD_params'].get('partdown', 100)
        shear_fn = savedir + '/shear_N' + str(Ndim) + '_R' + str(Rsm) + '_down' + str(partdown)
        if os.path.exists(shear_fn + '.npy'):
            shearmark = np.load(shear_fn + '.npy')
        else:
            raise NotImplementedError('the shear field (calc_shearmark, :1055-1127: field particles, smoothing, tidal tensor) is not '
                                      f'computed by the MI355X build: supply {shear_fn}.npy')
    for i in range(numslabs):
        prepare_slab(i, savedir=savedir, simdir=simdir, simname=simname, z_mock=z_mock, z_type=ztype, tracer_flags=tracer_flags,
                     MT=MT, want_ranks=want_ranks, want_AB=want_AB, want_shear=want_shear, shearmark=shearmark, cleaning=cleaning,
                     newseed=newseed, halo_lc=halo_lc, nthread=1, overwrite=overwrite, numslabs=numslabs)
