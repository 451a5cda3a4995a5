"""CPU ORACLE - TEST INFRASTRUCTURE.  NumPy restatement of the data-parallel core of the reference's
`prepare_sim.prepare_slab` (abacusnbody/hod/prepare_sim.py:296-1052): halo down-sampling, concentration / environment /
shear ranks, per-halo particle selection, the five satellite rank columns, the per-particle host columns and the random
columns - everything between "the CompaSO loader handed over `halos` and `parts`" and "the two HDF5 datasets are written".
The light-cone edge correction of the environment (:474-616, `gen_rand` :200-278) is restated by `lightcone_menv`.
File I/O and the CompaSO / ASDF readers are not restated.

Pinned by golden vectors of the REFERENCE's own prepare_slab run under the shims of oracle/make_golden.py on seeded synthetic
slabs (tests/golden/prepare_sim.npz, tests/test_oracle_prepare.py).  Random numbers: the reference consumes NumPy's global
legacy generator in a fixed order (:349-350 seeding, :449 halo mask, :163/:172 one `choice` per kept halo, :984-996 halo
randoms, :1029 particle randoms); `rng='numpy'` below consumes it in exactly that order, so that a run seeded like the
reference is comparable value for value.
"""
import numpy as np
from scipy.spatial import cKDTree

NBINS = 100     # mass bins of the rank columns (:454)


def subsample_halos(m, MT):
    """fraction of halos kept as a function of mass (:83-108)"""
    x = np.log10(m)
    if not MT:
        f = 1.0 / (1.0 + 0.1 * np.exp(-(x - 11.8) * 10))
        f[x > 13.0] = 1
        return f
    f = np.zeros(len(x))
    lo, mid = x < 11.4, x < 11.6
    f[lo] = 0.2 / (1.0 + 10 * np.exp(-(x[lo] - 11.2) * 25))
    sel = mid & ~lo
    f[sel] = 0.4 / (1.0 + 10 * np.exp(-(x[sel] - 11.3) * 25))
    f[~mid] = 1.0 / (1.0 + 0.1 * np.exp(-(x[~mid] - 11.7) * 10))
    return f


def particle_target(m_in, n_in, MT):
    """how many of a halo's n_in subsample particles are kept (submask_particles, :152-174); 0 below the mass floor"""
    x = np.log10(m_in)
    if MT:
        if m_in < 1e11:
            return 0
        return int(min(min(n_in, int(1 + 1.5 * 10 ** (x - 12.5))), 100))
    if 10 ** x < 1e12:
        return 0
    return int(min(n_in, int(1 + 1.5 * 10 ** (x - 13))))


def rank_in_mass_bins(values, masses, mbins, denom='max'):
    """per mass bin (strictly inside both edges, :763 / :287), the rank of `values` rescaled to [-0.5, 0.5]; bins with one
    halo and halos on an edge stay 0.  denom: 'max' divides by the largest rank (:771,790), 'n-1' by N - 1 (:291)"""
    out = np.zeros(len(values))
    for b in range(len(mbins) - 1):
        sel = (masses > mbins[b]) & (masses < mbins[b + 1])
        n = int(sel.sum())
        if n > 1:
            r = values[sel].argsort().argsort()
            out[sel] = r / (np.max(r) if denom == 'max' else (n - 1)) - 0.5
    return out


def satellite_ranks(ppos, pvel, hpos, hvel, allpos, N, Mpart, h, r25, r98):
    """the five rank columns of one halo's selected particles (:899-977): nearest-neighbour distance among ALL subsample
    particles of the halo, distance and speed relative to the halo, radial velocity, NFW perihelion - each as
    (rank - mean rank) / mean rank"""
    def norm(key):
        r = key.argsort().argsort()
        return (r - np.mean(r)) / np.mean(r)

    tree = cKDTree(allpos)
    ranksc = norm(tree.query(ppos, k=2)[0][:, 1])
    r_rel = ppos - hpos
    ranks = norm(np.sum(r_rel ** 2, axis=1))
    v_rel = pvel - hvel
    v_rel2 = np.sum(v_rel ** 2, axis=1)
    ranksv = norm(v_rel2)
    r0 = np.sqrt(np.sum(r_rel ** 2, axis=1))
    vel_rad = np.sum(v_rel * (r_rel / r0[:, None]), axis=1)
    ranksr = norm(vel_rad)
    v_rad2 = vel_rad ** 2
    v_tan2 = v_rel2 - v_rad2
    m = N * Mpart / h
    rs = r25
    c = r98 / rs
    r0_kpc = r0 * 1000
    alpha = 1.0 / (np.log(1 + c) - c / (1 + c)) * 2 * 6.67e-11 * m * 2e30 / r0_kpc / 3.086e19 / 1e6
    x2 = v_tan2 / (v_tan2 + v_rad2)
    A = v_tan2 + v_rad2
    B = np.log(1 + r0_kpc / rs)
    with np.errstate(all='ignore'):
        for _ in range(20):
            oldx = np.sqrt(x2)
            x2 = v_tan2 / (A + alpha * (np.log(1 + oldx * r0_kpc / rs) / oldx - B))
    x2[np.isnan(x2)] = 1
    ranksp = norm(r0_kpc ** 2 * x2)
    return ranks, ranksv, ranksp, ranksr, ranksc


def prepare_slab_core(halos, parts, Mpart, h, MT, want_ranks=False, want_AB=True, Menv=None, shearmark=None, Lbox=None,
                      mcut=1e11, halo_lc=False):
    """halos / parts: dicts of columns as CompaSOHaloCatalog hands them over (abacusutils_amd.synth.synth_compaso_slabs).
    Menv: the raw environment masses of the slab's halos (do_Menv_from_tree; the reference writes them to the env file for the
    global ranking later, :748-756) - unused here except for light cones, where fenv_rank is ranked per slab (:618).
    Returns (halo table of the KEPT halos, particle table of the kept particles) as dicts with the field names of the
    reference's HDF5 datasets (:1001-1045), plus `mask_halos` over the input halos."""
    nh = len(halos['N'])
    masses = halos['N'] * Mpart
    p_halos = subsample_halos(masses, MT)
    mask_halos = np.random.random(nh) < p_halos                                              # (:449)
    H = {k: v for k, v in halos.items()}
    H['mask_subsample'] = mask_halos
    H['multi_halos'] = 1.0 / p_halos
    mbins = np.logspace(np.log10(mcut), 15.5, NBINS + 1)
    if want_AB:
        if halo_lc:
            H['fenv_rank'] = rank_in_mass_bins(np.asarray(Menv), masses, mbins, denom='n-1')  # calc_fenv_opt (:283-293,618)
        else:
            H['fenv_rank'] = np.zeros(nh)                     # ranked globally by AbacusHOD.staging() later (:758-759)
        conc = halos['r98_L2com'] / halos['r25_L2com']
        # the median the reference subtracts per bin (:768) does not change the order
        H['deltac_rank'] = rank_in_mass_bins(conc, masses, mbins)
    else:
        H['fenv_rank'] = np.zeros(nh)
        H['deltac_rank'] = np.zeros(nh)
    if shearmark is not None:                                                                # (:776-796)
        ndim = len(shearmark)
        cell = Lbox / ndim
        g = (halos['x_L2com'] / cell).astype(int) % ndim
        H['shear_rank'] = rank_in_mass_bins(shearmark[g[:, 0], g[:, 1], g[:, 2]], masses, mbins)
    else:
        H['shear_rank'] = np.zeros(nh)

    pstart, pnum = halos['npstartA'], halos['npoutA']
    npart = len(parts['pos'])
    mask_parts = np.zeros(npart, dtype=bool)
    host = np.full(npart, -1, dtype=np.int64)
    Np = np.full(npart, -1.0)
    rk = {k: np.full(npart, -1.0) for k in ('ranks', 'ranksv', 'ranksp', 'ranksr', 'ranksc')}
    pstart_new = np.zeros(nh)
    pnum_new = np.zeros(nh)
    tracker = 0
    for j in range(nh):
        if not (mask_halos[j] and pnum[j] > 0):
            pstart_new[j] = pnum_new[j] = -1
            continue
        a, n_in = int(pstart[j]), int(pnum[j])
        ntarget = particle_target(masses[j], n_in, MT)
        sub = np.zeros(n_in, dtype=bool)
        if ntarget > 0 or (MT and masses[j] >= 1e11) or (not MT and 10 ** np.log10(masses[j]) >= 1e12):
            sub[np.random.choice(n_in, ntarget, replace=False)] = True                        # (:163,172)
        k = int(sub.sum())
        mask_parts[a:a + n_in] = sub
        host[a:a + n_in] = j
        Np[a:a + n_in] = k
        pstart_new[j], pnum_new[j] = tracker, k
        tracker += k
        if want_ranks and k > 0:
            idx = a + np.nonzero(sub)[0]
            if k == 1:
                for arr in rk.values():
                    arr[idx] = 0
                continue
            r = satellite_ranks(parts['pos'][idx], parts['vel'][idx], halos['x_L2com'][j], halos['v_L2com'][j],
                                parts['pos'][a:a + n_in], halos['N'][j], Mpart, h, halos['r25_L2com'][j], halos['r98_L2com'][j])
            for name, val in zip(('ranks', 'ranksv', 'ranksp', 'ranksr', 'ranksc'), r):
                rk[name][idx] = val
    H['npstartA'], H['npoutA'] = pstart_new, pnum_new
    sig = np.repeat(halos['sigmav3d_L2com'], 3).reshape((-1, 3)) / np.sqrt(3)
    H['randoms'] = np.random.random(nh)                                                       # (:984)
    H['randoms_exp'] = (np.random.randint(0, 2, size=(nh, 3)) * 2 - 1) * np.random.exponential(scale=sig, size=(nh, 3))
    H['randoms_gaus_vrms'] = np.random.normal(loc=0, scale=sig, size=(nh, 3))
    Hk = {k: np.asarray(v)[mask_halos] for k, v in H.items()}

    hp = host[mask_parts]
    P = {'pos': parts['pos'][mask_parts], 'vel': parts['vel'][mask_parts]}
    if want_ranks:
        for k, v in rk.items():
            P[k] = v[mask_parts]
    P['downsample_halo'] = p_halos[hp]
    P['halo_vel'] = halos['v_L2com'][hp].astype(np.float64)
    P['halo_mass'] = masses[hp].astype(np.float64)
    P['Np'] = Np[mask_parts]
    P['halo_id'] = halos['id'][hp].astype(np.int64)
    P['randoms'] = np.random.random(int(mask_parts.sum()))                                     # (:1029)
    P['halo_deltac'] = H['deltac_rank'][hp]
    P['halo_fenv'] = H['fenv_rank'][hp]
    P['halo_shear'] = H['shear_rank'][hp]
    return Hk, P, mask_halos


# ---- light cones: environment with the edge correction (:474-616) ------------------------------------------------------------
def lightcone_shell_randoms(N, chi_min, chi_max, Lbox, offset, origins, rng):
    """gen_rand (:200-278) with fac = 1: N points uniform in direction (the positive octant for the three-origin layout) and
    in radius, kept where they fall into the light-cone boxes; returns positions in box coordinates and the radii"""
    origin = origins[0]
    octant = origins.shape[0] > 1
    costheta = rng.random(N) if octant else rng.random(N) * 2.0 - 1.0                        # (:214-218)
    phi = rng.random(N) * np.pi / 2.0 if octant else rng.random(N) * 2.0 * np.pi
    theta = np.arccos(costheta)
    chis = rng.random(N) * (chi_max - chi_min) + chi_min                                     # (:223)
    xyz = np.empty((N, 3))
    xyz[:, 0] = np.sin(theta) * np.cos(phi)
    xyz[:, 1] = np.sin(theta) * np.sin(phi)
    xyz[:, 2] = np.cos(theta)
    xyz *= chis[:, None]
    # the cube of half-width Lbox / 2 with `offset` removed from the faces the catalogue lacks (:242-254), seen from the observer
    vert = (2 * ((np.arange(8)[:, None] & (1 << np.arange(3))) > 0) - 1) * (Lbox / 2.0)
    neg, posi = vert < 0, vert > 0
    vert[neg] += offset
    vert[posi[:, 0], 0] -= offset
    if not octant:
        vert[posi[:, 1], 1] -= offset
        vert[posi[:, 2], 2] -= offset
    centres = [np.zeros(3) - origin]
    if octant and chi_max >= (Lbox - offset):                                                # (:258-262)
        centres += [np.array([0.0, 0.0, Lbox]) - origin, np.array([0.0, Lbox, 0.0]) - origin]
    mask = np.zeros(N, dtype=bool)
    for c in centres:
        v = c + vert
        mask |= np.all((xyz > v.min(axis=0)) & (xyz <= v.max(axis=0)), axis=1)              # is_in_cube (:181-197)
    return xyz[mask] + origin, chis[mask]


def lightcone_menv(pos, masses, r98, Lbox, origins, seed, rad_outer=10, mcut=1e11, offset=10.0, rand_final=10):
    """Menv of a halo light-cone slab (:474-616): halos within rad_outer of a boundary (box faces less `offset`, the two
    shell radii) have their annulus mass divided by the annulus' completeness, measured by counting randoms of
    `default_rng(seed)` in [r98, rad_outer] around them (KD-tree ball queries).  Returns (Menv, index_bounds, rand_norm)."""
    from . import oracle as O
    origins = np.asarray(origins, dtype=np.float64).reshape(-1, 3)
    single = origins.shape[0] == 1
    alldist = np.sqrt(np.sum((pos - origins[0]) ** 2.0, axis=1))
    r_min, r_max = alldist.min(), alldist.max()

    def away_from_edges(p, d, pad):
        lo = -(Lbox / 2.0 - offset - pad)
        xhi = Lbox / 2.0 - offset - pad
        yzhi = xhi if single else 3.0 / 2 * Lbox - pad
        m = (lo <= p[:, 0]) & (xhi >= p[:, 0])
        for ax in (1, 2):
            m = m & (lo <= p[:, ax]) & (yzhi >= p[:, ax])
        return m & (r_min + pad <= d) & (r_max - pad >= d)

    index_bounds = np.arange(pos.shape[0], dtype=int)[~away_from_edges(pos, alldist, rad_outer)]
    rand_norm = np.zeros(len(index_bounds))
    if len(index_bounds) > 0:
        rand_N = int(pos.shape[0])
        vol = 4.0 / 3.0 * np.pi * (r_max**3 - r_min**3) if single else 4.0 / 3.0 / 8.0 * np.pi * (r_max**3 - r_min**3)
        rand_n = rand_N / vol
        rng = np.random.default_rng(seed)
        count = repeats = 0
        while count < len(index_bounds) * rand_final:
            randpos, randdist = lightcone_shell_randoms(pos.shape[0], r_min, r_max, Lbox, offset, origins, rng)
            randpos = randpos[~away_from_edges(randpos, randdist, 2.0 * rad_outer)]
            if randpos.shape[0] > 0:
                tree = cKDTree(randpos)
                inner = tree.query_ball_point(pos[index_bounds], r=r98[index_bounds], return_length=True)
                outer = tree.query_ball_point(pos[index_bounds], r=rad_outer, return_length=True)
                rand_norm += outer - inner
            repeats += 1
            count += randpos.shape[0]
        rand_n *= repeats
        rand_norm /= (rad_outer**3.0 - r98[index_bounds] ** 3.0) * 4.0 / 3.0 * np.pi * rand_n
    Menv = O.menv_brute(pos, masses, r98, rad_outer, True, Lbox, mcut=mcut)
    if len(index_bounds) > 0:
        zero = rand_norm == 0.0
        norm = np.where(zero, 1.0, rand_norm)
        fixed = Menv[index_bounds] / norm
        fixed[zero] = 0.0
        Menv[index_bounds] = fixed
    return Menv, index_bounds, rand_norm


# ---- the device's own random columns (`rng=<seed>` of abacusutils_amd.hod.prepare_sim; csrc/prepare.hip prep_halo_randoms /
# prep_part_randoms): Philox4x32-10 blocks (counter = global index, stream, block; key = seed) through the oracle's C restatement
# of the generator - held to the published known-answer vectors by tests/test_oracle_reseed.py - and the fixed float64
# evaluations of log / sin / cos.  Scalar loops: for small samples only.
def _u53(a, b):
    return float(((int(a) >> 5) << 26) | (int(b) >> 6)) * 1.1102230246251565e-16


def device_uniform(seed, index, stream):
    """one U[0,1) float64 per global object index on stream 5 (particle randoms) or 6 (halo mask draws)"""
    from oracle import oracle
    key = (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    out = np.empty(len(index))
    for r, g in enumerate(index):
        w = oracle.philox4x32_10((int(g) & 0xFFFFFFFF, (int(g) >> 32) & 0xFFFFFFFF, stream, 0), key)
        out[r] = _u53(w[0], w[1])
    return out


def device_halo_randoms(seed, index, scale):
    """(randoms, randoms_exp, randoms_gaus_vrms) of the halos with global indices `index` and scales sigmav3d / sqrt(3)"""
    from oracle import oracle
    key = (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    n = len(index)
    rnd, rexp, rg = np.empty(n), np.empty((n, 3)), np.empty((n, 3))
    for r, g in enumerate(index):
        c = (int(g) & 0xFFFFFFFF, (int(g) >> 32) & 0xFFFFFFFF)
        b = [oracle.philox4x32_10(c + (4, k), key) for k in range(5)]
        sc = float(scale[r])
        rnd[r] = _u53(b[0][0], b[0][1])
        e = [-oracle.rs_log(1.0 - _u53(b[1][0], b[1][1])), -oracle.rs_log(1.0 - _u53(b[1][2], b[1][3])),
             -oracle.rs_log(1.0 - _u53(b[2][0], b[2][1]))]
        for k in range(3):
            rexp[r, k] = (1.0 if (int(b[0][2]) >> k) & 1 else -1.0) * (e[k] * sc)
        m0 = np.sqrt(-2.0 * oracle.rs_log(1.0 - _u53(b[2][2], b[2][3])))
        m1 = np.sqrt(-2.0 * oracle.rs_log(1.0 - _u53(b[3][2], b[3][3])))
        s0, c0 = oracle.rs_sincos2pi(_u53(b[3][0], b[3][1]))
        s1, c1 = oracle.rs_sincos2pi(_u53(b[4][0], b[4][1]))
        rg[r] = ((m0 * c0) * sc, (m0 * s0) * sc, (m1 * c1) * sc)
    return rnd, rexp, rg
