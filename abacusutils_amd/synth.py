"""Seeded synthetic inputs for the hot path.

Shapes, dtypes and key names are those `AbacusHOD.staging()` produces in the
reference (abacusnbody/hod/abacus_hod.py:355-383,659-702): float64 arrays,
(N,3) C-order positions/velocities, int64 ids.  Used by bench.py, the parity
tests and oracle/make_golden.py so all three see identical inputs.
"""

import numpy as np

# AbacusSummit base box header values (tests/halo_light_cones/.../lc_halo_info.asdf:253,346,383)
LBOX_BASE = 2000.0
MPART_BASE = 2109081520.453063
VELZSPACE_TO_KMS_BASE = 208774.9025637363

# tests/abacus_hod.yaml:31-47,50-70,73-90 of the reference (the HOD the reference tests run)
LRG_PARAMS = dict(
    logM_cut=13.3, logM1=14.3, sigma=0.3, alpha=1.0, kappa=0.4,
    alpha_c=0, alpha_s=1, s=0, s_v=0, s_p=0, s_r=0,
    Acent=0, Asat=0, Bcent=0, Bsat=0, ic=0.97,
)
ELG_PARAMS = dict(
    p_max=0.33, Q=100.0, logM_cut=11.75, kappa=1.0, sigma=0.58, logM1=13.53,
    alpha=1.0, gamma=4.12, A_s=1.0, alpha_c=0, alpha_s=1, s=0, s_v=0, s_p=0,
    s_r=0, Acent=0, Asat=0, Bcent=0, Bsat=0, ic=1.0,
)
QSO_PARAMS = dict(
    p_max=0.33, logM_cut=12.21, kappa=1.0, sigma=0.56, logM1=13.94, alpha=0.4,
    A_s=1.0, alpha_c=0, alpha_s=1, s=0, s_v=0, s_p=0, s_r=0,
    Acent=0, Asat=0, Bcent=0, Bsat=0, ic=1.0,
)

# "production" multi-tracer mix of BASELINE config 5 (LRG + ELG + QSO on one catalogue): the yaml blocks above with
# assembly bias, satellite rank modulation, velocity bias and ELG conformity switched on (the keys an MCMC samples,
# docs/hod.rst; magnitudes typical of the AbacusSummit fits).  QSO incompleteness 0.05: the yaml's ic = 1 on the synthetic
# mass function of synth_hod_inputs would put a quasar in 70 % of the halos.
PRODUCTION_TRACERS = {
    'LRG': dict(LRG_PARAMS, alpha_c=0.3, alpha_s=0.9, s=0.2, s_v=-0.1, s_p=0.1, s_r=-0.15,
                Acent=-0.15, Asat=0.1, Bcent=0.08, Bsat=-0.12),
    'ELG': dict(ELG_PARAMS, alpha_c=0.2, alpha_s=1.1, s=-0.2, s_v=0.1, s_p=-0.1, s_r=0.1,
                Acent=0.1, Asat=-0.1, Bcent=-0.1, Bsat=0.15, Ccent=0.05, Csat=-0.05,
                logM1_EE=13.2, alpha_EE=0.9, logM1_EL=13.8, alpha_EL=1.1),
    'QSO': dict(QSO_PARAMS, alpha_c=0.4, alpha_s=1.0, s=0.1, s_v=0.1, s_p=-0.1, s_r=0.05,
                Acent=0.1, Asat=0.05, Bcent=-0.05, Bsat=0.1, ic=0.05),
}


def _downfactor_lrg(m):
    """halo subsampling fraction, LRG-only branch (hod/prepare_sim.py:103-107)"""
    x = np.log10(m)
    d = 1.0 / (1.0 + 0.1 * np.exp(-(x - 11.8) * 10))
    d[x > 13.0] = 1
    return d


def synth_hod_inputs(n_halo, n_part, seed=600, lbox=LBOX_BASE, z=0.5,
                     with_ranks=False, with_shear=True, origin=None):
    """Synthetic halo + particle subsample in the `staging()` layout.

    Returns (halo_data, particle_data, params).
    """
    rng = np.random.default_rng(seed)
    logm = 11.0 + rng.exponential(0.45, n_halo)
    logm = np.minimum(logm, 15.5)
    hmass = 10.0**logm
    hpos = (rng.random((n_halo, 3)) - 0.5) * lbox
    hvel = rng.standard_normal((n_halo, 3)) * 300.0
    hsigma3d = 300.0 * (hmass / 1e13) ** (1.0 / 3.0)
    hveldev = rng.standard_normal((n_halo, 3)) * (hsigma3d / np.sqrt(3.0))[:, None]
    hmultis = 1.0 / _downfactor_lrg(hmass)
    hrandoms = rng.random(n_halo)
    hdeltac = rng.random(n_halo) - 0.5
    hfenv = rng.random(n_halo) - 0.5
    hshear = rng.random(n_halo) - 0.5
    hid = 1000 * np.arange(n_halo, dtype=np.int64)

    # particles: host drawn proportional to mass, stored in host order (as the
    # slab files are), so `pinds` is non-decreasing
    cdf = np.cumsum(hmass)
    # (the draws are sorted BEFORE the look-up: searchsorted is monotone, so this is the sorted host list the look-up of the
    # unsorted draws followed by a sort gives, value for value - without 4e7 cache-missing binary searches at the C4 shard size)
    draws = rng.random(n_part) * cdf[-1]
    draws.sort()
    host = np.searchsorted(cdf, draws)
    del draws
    host = np.minimum(host, n_halo - 1)
    np_host = np.bincount(host, minlength=n_halo).astype(np.float64)
    ppos = hpos[host] + rng.standard_normal((n_part, 3)) * 0.3
    pvel = hvel[host] + rng.standard_normal((n_part, 3)) * hsigma3d[host][:, None] / np.sqrt(3.0)
    halo_data = dict(
        hpos=hpos, hvel=hvel, hmass=hmass, hid=hid, hmultis=hmultis,
        hrandoms=hrandoms, hveldev=hveldev, hsigma3d=hsigma3d,
        hc=np.full(n_halo, 5.0), hrvir=np.full(n_halo, 0.5),
        hdeltac=hdeltac, hfenv=hfenv,
    )
    particle_data = dict(
        ppos=ppos, pvel=pvel, phvel=np.ascontiguousarray(hvel[host]),
        phmass=hmass[host], phid=hid[host],
        pweights=1.0 / np_host[host], prandoms=rng.random(n_part),
        pinds=host.astype(np.int64), pdeltac=hdeltac[host], pfenv=hfenv[host],
    )
    if with_shear:
        halo_data['hshear'] = hshear
        particle_data['pshear'] = hshear[host]
    if with_ranks:
        for k in ('pranks', 'pranksv', 'pranksp', 'pranksr', 'pranksc'):
            particle_data[k] = rng.random(n_part) * 2.0 - 1.0
    else:
        # hod/abacus_hod.py:698-702
        for k in ('pranks', 'pranksv', 'pranksp', 'pranksr', 'pranksc'):
            particle_data[k] = np.ones(n_part)
    params = dict(
        z=z, h=0.6736, Lbox=lbox, Mpart=MPART_BASE,
        velz2kms=VELZSPACE_TO_KMS_BASE / lbox,
        origin=None if origin is None else np.asarray(origin, dtype=np.float64),
        chunk=-1, numslabs=1,
    )
    return halo_data, particle_data, params


def synth_positions(n, lbox, seed=300, dtype=np.float32, clustered=False):
    """Particle positions for the P(k) path (scripts/power/bench.py:28,39 protocol:
    `rng.random((N,3), dtype='f4')`, seed 300), scaled to `lbox`."""
    rng = np.random.default_rng(seed)
    pos = rng.random((n, 3), dtype=np.float32)
    if clustered:
        # cheap non-Poisson signal: displace along a few long-wavelength modes
        for ax in range(3):
            pos[:, ax] += np.float32(0.02) * np.sin(
                np.float32(2 * np.pi * (ax + 2)) * pos[:, (ax + 1) % 3]
            ).astype(np.float32)
        pos -= np.floor(pos)
    pos = (pos * np.float32(lbox)).astype(dtype)
    return pos


def synth_compaso_slabs(numslabs=3, n_halo=3000, seed=900, lbox=300.0, mpart=MPART_BASE, subsample_frac=0.03):
    """`numslabs` CompaSO-like slabs of one periodic box, as `prepare_sim.prepare_slab` sees them after
    `CompaSOHaloCatalog(..., subsamples=dict(A=True, rv=True))` (hod/prepare_sim.py:404-441): per slab a dict
    `halos` - N (u4), x_L2com / v_L2com (n,3) f4, r25 / r90 / r98_L2com f4, npstartA / npoutA (i8: the halo's slice of
    `parts`), id (u8, unique across slabs), sigmav3d_L2com f4 - and `parts` - pos / vel (m,3) f4, the subsample-A particles
    of the slab's halos in halo order.  Halo x lies in the slab's x-range of [-L/2, L/2); masses follow a steep mass function
    from 35 particles up, so every branch of subsample_halos / submask_particles (:83-174) is populated."""
    out = []
    dx = lbox / numslabs
    for s in range(numslabs):
        rng = np.random.default_rng(seed + s)
        logm = np.minimum(10.9 + rng.exponential(0.55, n_halo), 15.3)
        N = np.maximum((10 ** logm / mpart).astype(np.int64), 35).astype(np.uint32)
        x = np.empty((n_halo, 3), dtype=np.float32)
        x[:, 0] = (-0.5 * lbox + s * dx + rng.random(n_halo) * dx).astype(np.float32)
        x[:, 1:] = ((rng.random((n_halo, 2)) - 0.5) * lbox).astype(np.float32)
        np.clip(x, -0.5 * lbox, np.nextafter(np.float32(0.5 * lbox), np.float32(0)), out=x)
        v = (rng.standard_normal((n_halo, 3)) * 300).astype(np.float32)
        m = N.astype(np.float64) * mpart
        r98 = (0.25 * (m / 1e13) ** (1.0 / 3.0) * (1 + 0.1 * rng.standard_normal(n_halo))).astype(np.float32)
        r98 = np.maximum(r98, np.float32(0.02))
        conc = (4.0 + 6.0 * rng.random(n_halo)).astype(np.float32)
        r25 = (r98 / conc).astype(np.float32)
        r90 = (r98 * np.float32(0.8)).astype(np.float32)
        sig = (300.0 * (m / 1e13) ** (1.0 / 3.0)).astype(np.float32)
        npout = rng.binomial(N.astype(np.int64), subsample_frac).astype(np.int64)
        npout[rng.random(n_halo) < 0.02] = 0                      # halos without subsample particles (:868)
        npstart = np.concatenate(([0], np.cumsum(npout)[:-1])).astype(np.int64)
        host = np.repeat(np.arange(n_halo), npout)
        nprt = len(host)
        rr = (r98[host] * (0.02 + 0.98 * rng.random(nprt) ** 1.5))[:, None] * _unit_vectors(rng, nprt)   # never on the centre
        pos = (x[host] + rr.astype(np.float32)).astype(np.float32)
        vel = (v[host] + (rng.standard_normal((nprt, 3)) * sig[host][:, None] / np.sqrt(3.0)).astype(np.float32)).astype(np.float32)
        halos = dict(N=N, x_L2com=x, v_L2com=v, r25_L2com=r25, r90_L2com=r90, r98_L2com=r98, npstartA=npstart, npoutA=npout,
                     id=(np.uint64(s) * np.uint64(10**12) + np.arange(n_halo, dtype=np.uint64)), sigmav3d_L2com=sig)
        out.append(dict(halos=halos, parts=dict(pos=pos, vel=vel)))
    header = dict(BoxSizeHMpc=float(lbox), BoxSize=float(lbox), ParticleMassHMsun=float(mpart), H0=67.36,
                  VelZSpace_to_kms=VELZSPACE_TO_KMS_BASE * lbox / LBOX_BASE)
    return out, header


def synth_lightcone_slab(n_halo=2500, seed=950, lbox=300.0, geometry='octant', chi=None, **kw):
    """One slab of a halo light-cone catalogue as `prepare_slab(halo_lc=True)` sees it (hod/prepare_sim.py:362-433): the tables
    of `synth_compaso_slabs` with the halos moved into a shell around the observer - `geometry='octant'`: the three-origin
    layout of the base boxes (observer 10 Mpc/h inside the box corner, header `LightConeOrigins` of three rows), positive
    directions only; `'centre'`: one observer in the middle of the box, the shell cut by the box faces.  The particles move
    with their hosts.  Returns (slab dict, header)."""
    slabs, header = synth_compaso_slabs(numslabs=1, n_halo=n_halo, seed=seed, lbox=lbox, **kw)
    halos, parts = slabs[0]['halos'], slabs[0]['parts']
    rng = np.random.default_rng(seed + 7919)
    half, off = 0.5 * lbox, 10.0
    if geometry == 'octant':
        o0 = np.array([-half + off] * 3)
        origins = np.array([o0, o0 - [0.0, 0.0, lbox], o0 - [0.0, lbox, 0.0]])
        chi = chi or (0.3 * lbox, 0.85 * lbox)
    elif geometry == 'centre':
        origins = np.zeros((1, 3))
        chi = chi or (0.2 * lbox, 0.5 * lbox)
    else:
        raise ValueError(geometry)
    x = np.empty((0, 3), dtype=np.float32)
    while len(x) < n_halo:
        m = 4 * n_halo
        u = _unit_vectors(rng, m)
        if geometry == 'octant':
            u = np.abs(u)
        r = (chi[0] ** 3 + rng.random(m) * (chi[1] ** 3 - chi[0] ** 3)) ** (1.0 / 3.0)
        cand = (origins[0] + u * r[:, None]).astype(np.float32)
        hi = half - off if geometry == 'centre' else half
        ok = np.all(cand > np.float32(-half + off), axis=1) & (cand[:, 0] <= np.float32(half - off)) & np.all(cand[:, 1:] <= np.float32(hi), axis=1)
        x = np.concatenate([x, cand[ok]])[:n_halo]
    shift = x - halos['x_L2com']
    host = np.repeat(np.arange(n_halo), halos['npoutA'])
    parts['pos'] = (parts['pos'] + shift[host]).astype(np.float32)
    halos['x_L2com'] = x
    header = dict(header, LightConeOrigins=[float(v) for v in origins.ravel()])
    return dict(halos=halos, parts=parts), header


def _unit_vectors(rng, n):
    u = rng.standard_normal((n, 3))
    return u / np.sqrt((u * u).sum(axis=1))[:, None]
