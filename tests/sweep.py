"""Seeded random HOD cases shared by the oracle-vs-reference (CPU) and HIP-vs-oracle (GPU) sweeps."""
import numpy as np

from abacusutils_amd import synth

_SUBSETS = [('LRG',), ('ELG',), ('QSO',), ('LRG', 'ELG'), ('LRG', 'QSO'), ('ELG', 'QSO'), ('LRG', 'ELG', 'QSO')]


def sweep_case(seed, nh=60000, npart=90000):
    """(halo_data, particle_data, params, tracers, enable_ranks, rsd) for one seed: parameters drawn uniformly over the
    ranges an MCMC explores, every fourth case on the light-cone RSD branch"""
    rng = np.random.default_rng(1000 + seed)
    u = lambda a, b: float(rng.uniform(a, b))   # noqa: E731
    hd, pd, params = synth.synth_hod_inputs(nh, npart, seed=200 + seed, with_ranks=True)
    if seed % 4 == 3:
        params = dict(params, origin=np.array([-990.0, -990.0, -990.0]))
    ab = lambda: u(-0.3, 0.3) if rng.random() < 0.7 else 0.0   # noqa: E731
    lrg = dict(logM_cut=u(12.3, 13.6), logM1=u(13.2, 14.8), sigma=u(0.05, 1.0), alpha=u(0.6, 1.5), kappa=u(0.0, 1.5),
               alpha_c=u(0, 0.6), alpha_s=u(0.6, 1.4), s=u(-0.5, 0.5), s_v=u(-0.5, 0.5), s_p=u(-0.5, 0.5), s_r=u(-0.5, 0.5),
               Acent=ab(), Asat=ab(), Bcent=ab(), Bsat=ab(), ic=u(0.5, 1.0))
    elg = dict(p_max=u(0.1, 0.9), Q=u(20, 200), logM_cut=u(11.3, 12.3), kappa=u(0.2, 2.0), sigma=u(0.2, 1.2),
               logM1=u(12.8, 14.2), alpha=u(0.6, 1.4), gamma=u(1.0, 8.0), A_s=u(0.5, 1.5), alpha_c=u(0, 0.6),
               alpha_s=u(0.6, 1.4), s=u(-0.5, 0.5), s_v=u(-0.5, 0.5), s_p=u(-0.5, 0.5), s_r=u(-0.5, 0.5), Acent=ab(),
               Asat=ab(), Bcent=ab(), Bsat=ab(), Ccent=ab(), Csat=ab(), ic=u(0.5, 1.0), logM1_EE=u(12.8, 14.2),
               alpha_EE=u(0.6, 1.4), logM1_EL=u(12.8, 14.2), alpha_EL=u(0.6, 1.4))
    qso = dict(p_max=u(0.1, 0.9), logM_cut=u(11.8, 12.8), kappa=u(0.2, 2.0), sigma=u(0.2, 1.2), logM1=u(13.0, 14.5),
               alpha=u(0.3, 1.2), A_s=u(0.5, 1.5), alpha_c=u(0, 0.6), alpha_s=u(0.6, 1.4), s=u(-0.5, 0.5), s_v=u(-0.5, 0.5),
               s_p=u(-0.5, 0.5), s_r=u(-0.5, 0.5), Acent=ab(), Asat=ab(), Bcent=ab(), Bsat=ab(), ic=u(0.5, 1.0))
    pick = _SUBSETS[seed % len(_SUBSETS)]
    tracers = {k: v for k, v in (('LRG', lrg), ('ELG', elg), ('QSO', qso)) if k in pick}
    return hd, pd, params, tracers, bool(seed % 2), bool((seed // 2) % 3)
