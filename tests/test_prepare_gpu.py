"""prepare_sim on the device (abacusutils_amd/hod/prepare_sim.py, csrc/prepare.hip) - VERDICT r02 item 6:
  * with NumPy's stream consumed in the reference's order, against the golden vectors of the REFERENCE's own prepare_slab
    (hod/prepare_sim.py:296-1052, tests/golden/prepare_sim.npz) and, at a larger size, against the oracle's restatement;
  * with the device's Philox selection: the rules the selection obeys, determinism, uniformity;
  * end to end: synthetic slabs -> prepare on the device -> AbacusHOD.from_prepared -> run_hod == the oracle's catalogue."""
import json

import numpy as np
import pytest
from conftest import assert_mock_equal, load_golden

from abacusutils_amd import synth
from oracle import oracle
from oracle import prepare_oracle as po

pytestmark = pytest.mark.gpu

CASES = {'mt_ab': (1, True, False, True, False), 'lrg_ranks_ab': (0, False, True, True, False),
         'mt_ranks_ab_shear': (2, True, True, True, True)}
EXACT = ('N', 'x_L2com', 'v_L2com', 'N_interp', 'pos_interp', 'vel_interp', 'index_halo', 'r25_L2com', 'r90_L2com', 'r98_L2com', 'id', 'sigmav3d_L2com', 'mask_subsample', 'npstartA',
         'npoutA', 'randoms', 'randoms_exp', 'randoms_gaus_vrms', 'fenv_rank', 'deltac_rank', 'shear_rank',
         'pos', 'vel', 'halo_vel', 'halo_mass', 'Np', 'halo_id', 'halo_deltac', 'halo_fenv', 'halo_shear',
         'ranks', 'ranksv', 'ranksr')
ULP = ('multi_halos', 'downsample_halo')         # exp / log10 of ocml against NumPy's: last-place differences


def shearmark(ndim=16, seed=5):
    return np.random.default_rng(seed).random((ndim, ndim, ndim))


def compare_tables(got, want, label):
    assert sorted(got) == sorted(want), (label, sorted(got), sorted(want))
    for k, w in want.items():
        g = got[k]
        assert g.shape == w.shape and g.dtype == w.dtype, (label, k, g.shape, w.shape, g.dtype, w.dtype)
        if k in EXACT:
            np.testing.assert_array_equal(g, w, err_msg=f'{label}.{k}')
        elif k in ULP:
            np.testing.assert_allclose(g, w, rtol=5e-16, atol=0, err_msg=f'{label}.{k}')
        else:
            # ranksp: the perihelion iteration starts from float32 logarithms (np.log of float32 arrays, :945,:956-961), which
            # NumPy's SIMD kernels do not round like a float64 log rounded once: two particles of a halo whose r_p^2 agree
            # to ~1e-7 can swap places.  ranksc: two kept particles that are each other's nearest neighbour have EQUAL keys;
            # the reference's order among ties is whatever NumPy's unstable argsort makes of them (the device breaks ties by
            # particle index).  A swap exchanges two neighbouring ranks of one halo: everything else about the columns holds.
            assert k in ('ranksp', 'ranksc'), k
            same = g == w
            assert same.mean() > 0.99, (label, k, float(same.mean()))
            np.testing.assert_allclose(np.sort(g), np.sort(w), rtol=0, atol=1e-12)
            host = got['halo_id']
            for hid in np.unique(host[~same]):
                sel = host == hid
                np.testing.assert_allclose(np.sort(g[sel]), np.sort(w[sel]), rtol=0, atol=1e-12, err_msg=f'{label}.{k} halo {hid}')


@pytest.mark.parametrize('case', list(CASES))
def test_numpy_stream_reproduces_the_reference(case):
    from abacusutils_amd.hod import prepare_sim as ps
    g = load_golden('prepare_sim')
    slabs, header = synth.synth_compaso_slabs(**json.loads(str(g['meta.synth_json'])))
    i, MT, want_ranks, want_AB, want_shear = CASES[case]
    ps.reference_seed(600, i)
    H, P, mask = ps.prepare_slab_arrays(slabs[i]['halos'], slabs[i]['parts'], header['ParticleMassHMsun'], header['H0'] / 100.0,
                                        MT, want_ranks=want_ranks, want_AB=want_AB, shearmark=shearmark() if want_shear else None,
                                        Lbox=header['BoxSizeHMpc'], rng='numpy')
    for kind, got in (('halos', H), ('particles', P)):
        want = {k.split('.', 2)[2]: g[k] for k in g if k.startswith(f'{case}.{kind}.')}
        compare_tables(got, want, f'{case}.{kind}')
    # the padded environment of the slab (the reference's env sidecar, :622-756)
    n = len(slabs)
    cid, cmass, Menv = ps.slab_environment(i, slabs[i]['halos'], [slabs[(i - 1) % n]['halos'], slabs[(i + 1) % n]['halos']], n,
                                           header['BoxSizeHMpc'], header['ParticleMassHMsun'], rad_outer=10, mcut=1e11)
    np.testing.assert_array_equal(cid, g[f'{case}.env.id'])
    np.testing.assert_array_equal(cmass, g[f'{case}.env.mass'])
    np.testing.assert_allclose(Menv, g[f'{case}.env.Menv'], rtol=1e-12, atol=1e-12 * g[f'{case}.env.mass'].max())


LC_CASES = {'lc_octant': ('octant', 0, True, False), 'lc_centre': ('centre', 2, False, True)}


@pytest.mark.parametrize('case', list(LC_CASES))
def test_lightcone_environment_reproduces_the_reference(case):
    """halo light cones (:474-616): edge halos, completeness of their annulus from randoms counted on the device, corrected
    environment masses - against the reference's own intermediate (the argument of its calc_fenv_opt) and the oracle - and
    the two tables of prepare_slab(halo_lc=True)"""
    from abacusutils_amd.hod import prepare_sim as ps
    g = load_golden('prepare_sim')
    geometry, i, MT, want_ranks = LC_CASES[case]
    slab, header = synth.synth_lightcone_slab(geometry=geometry, **json.loads(str(g['meta.lc_synth_json'])))
    halos, parts = slab['halos'], slab['parts']
    Mpart, Lbox, origins = header['ParticleMassHMsun'], header['BoxSizeHMpc'], header['LightConeOrigins']
    lc_seed = ps.reference_seed(600, i)
    masses = halos['N'] * Mpart
    # the counts are integers and the normalisation is the reference's expression: equal to the KD-tree's, bit for bit
    Menv_o, edge_o, norm_o = po.lightcone_menv(halos['x_L2com'], masses, halos['r98_L2com'], Lbox, origins, lc_seed)
    edge, norm = ps.lightcone_edge_norm(halos['x_L2com'], halos['r98_L2com'], Lbox, origins, lc_seed)
    np.testing.assert_array_equal(edge, edge_o)
    np.testing.assert_array_equal(norm, norm_o)
    want = g[f'{case}.Menv_corrected']
    Menv = ps.lightcone_environment(halos['x_L2com'], masses, halos['r98_L2com'], Lbox, origins, lc_seed)
    np.testing.assert_allclose(Menv, want, rtol=1e-12, atol=0)
    np.testing.assert_array_equal(Menv == 0, want == 0)
    rename = {'id': 'index_halo', 'x_L2com': 'pos_interp', 'v_L2com': 'vel_interp', 'N': 'N_interp'}   # the loader's keys (:370-373)

    def tables(**kw):
        ps.reference_seed(600, i)
        H, P, mask = ps.prepare_slab_arrays(halos, parts, Mpart, header['H0'] / 100.0, MT, want_ranks=want_ranks, want_AB=True,
                                            Lbox=Lbox, halo_lc=True, rng='numpy', **kw)
        for k, alias in rename.items():
            H[alias] = H[k]
        return H, P, mask

    gold = {kind: {k.split('.', 2)[2]: g[k] for k in g if k.startswith(f'{case}.{kind}.')} for kind in ('halos', 'particles')}

    def check(H, P, mask, exact_env):
        """every column as the reference wrote it; the environment rank (:618, calc_fenv_opt :283-293) where it is defined:
        halos that share their environment mass with another halo of their mass bin - the many with none at all - are
        ranked among themselves in whatever order an unstable argsort leaves them (NumPy's under the shim, Numba's in
        production; the device orders ties by index), so a tie group is compared as a set"""
        fr, hf = H.pop('fenv_rank'), P.pop('halo_fenv')
        wfr = gold['halos']['fenv_rank']
        compare_tables(H, {k: v for k, v in gold['halos'].items() if k != 'fenv_rank'}, f'{case}.halos')
        compare_tables(P, {k: v for k, v in gold['particles'].items() if k != 'halo_fenv'}, f'{case}.particles')
        mbins = np.logspace(np.log10(1e11), 15.5, 101)
        used = want if exact_env else Menv
        full = ps.rank_in_mass_bins(used, masses, mbins)                       # the device's ranks of ALL halos of the slab
        full_o = po.rank_in_mass_bins(want, masses, mbins, denom='n-1')        # the oracle's (pinned to the reference's)
        if exact_env:
            np.testing.assert_array_equal(fr, full[mask])
        else:       # the cell lists are filled through atomics: the order of a sum, and with it its last bit, varies from run to run
            assert (fr == full[mask]).mean() > 0.98
        np.testing.assert_array_equal(wfr, full_o[mask])
        bins = np.searchsorted(mbins, masses)
        _, group, size = np.unique(np.stack([bins.astype(np.float64), want]), axis=1, return_inverse=True, return_counts=True)
        group = group.ravel()
        single = size[group] == 1
        assert single.sum() > 0.2 * len(full)
        if exact_env:
            np.testing.assert_array_equal(full[single], full_o[single])
            np.testing.assert_array_equal(full[np.lexsort((full, group))], full_o[np.lexsort((full_o, group))])
        else:
            assert (full[single] == full_o[single]).mean() > 0.98
            np.testing.assert_allclose(full[single], full_o[single], rtol=0, atol=0.1)
            np.testing.assert_array_equal(full[np.lexsort((full, bins))], full_o[np.lexsort((full_o, bins))])
        order = np.argsort(H['id'])
        np.testing.assert_array_equal(hf, fr[order[np.searchsorted(H['id'][order], P['halo_id'])]])

    # from the recorded masses
    check(*tables(Menv=want), exact_env=True)
    # the whole path (masses worked out here): sums in another order than the tree's differ in the last bits, which may
    # swap the ranks of halos with nearly equal masses around them
    check(*tables(origins=origins, lc_seed=lc_seed), exact_env=False)


def test_lightcone_needs_its_inputs():
    from abacusutils_amd.hod import prepare_sim as ps
    slab, header = synth.synth_lightcone_slab(n_halo=300)
    with pytest.raises(ValueError, match='origins'):
        ps.prepare_slab_arrays(slab['halos'], slab['parts'], header['ParticleMassHMsun'], 0.67, True, want_AB=True, halo_lc=True,
                               Lbox=header['BoxSizeHMpc'])
    with pytest.raises(ValueError, match='origins'):
        ps.lightcone_edge_norm(slab['halos']['x_L2com'], slab['halos']['r98_L2com'], 300.0, np.zeros((2, 3)), 1)


@pytest.mark.parametrize('MT,want_ranks', [(True, True), (False, True), (True, False)])
def test_numpy_stream_against_the_oracle_at_size(MT, want_ranks):
    from abacusutils_amd.hod import prepare_sim as ps
    slabs, header = synth.synth_compaso_slabs(numslabs=1, n_halo=12000, seed=41, lbox=500.0)
    halos, parts = slabs[0]['halos'], slabs[0]['parts']
    Mpart, h = header['ParticleMassHMsun'], header['H0'] / 100.0
    ps.reference_seed(600, 7)
    with np.errstate(all='ignore'):
        Ho, Po, mo = po.prepare_slab_core(halos, parts, Mpart, h, MT, want_ranks=want_ranks, want_AB=True)
    ps.reference_seed(600, 7)
    H, P, m = ps.prepare_slab_arrays(halos, parts, Mpart, h, MT, want_ranks=want_ranks, want_AB=True, rng='numpy')
    np.testing.assert_array_equal(m, mo)
    compare_tables(H, Ho, 'halos')
    compare_tables(P, Po, 'particles')
    assert len(P['pos']) > 1000


def test_device_selection_obeys_the_rules():
    """rng = seed: per halo the kept count is submask_particles' target (:152-174), only kept halos keep particles, the kept
    particles lie in their halo's slice, new offsets are the running sum (:895-897); same seed same draw; uniform over the slice"""
    from abacusutils_amd.hod import prepare_sim as ps
    slabs, header = synth.synth_compaso_slabs(numslabs=1, n_halo=20000, seed=43, lbox=500.0)
    halos, parts = slabs[0]['halos'], slabs[0]['parts']
    Mpart, h = header['ParticleMassHMsun'], header['H0'] / 100.0
    for MT in (True, False):
        H, P, mask = ps.prepare_slab_arrays(halos, parts, Mpart, h, MT, want_ranks=True, want_AB=False, rng=1234)
        H2, P2, mask2 = ps.prepare_slab_arrays(halos, parts, Mpart, h, MT, want_ranks=True, want_AB=False, rng=1234)
        for k in P:
            np.testing.assert_array_equal(P[k], P2[k])
        H3, P3, _ = ps.prepare_slab_arrays(halos, parts, Mpart, h, MT, want_ranks=False, want_AB=False, rng=99)
        assert len(P3['pos']) != len(P['pos']) or not np.array_equal(P3['pos'], P['pos'])
        masses = halos['N'] * Mpart
        want = np.array([po.particle_target(masses[j], int(halos['npoutA'][j]), MT) if halos['npoutA'][j] > 0 else 0
                         for j in range(len(masses))])
        kept = np.where(halos['npoutA'][mask] > 0, want[mask], -1)
        np.testing.assert_array_equal(H['npoutA'], kept)
        live = H['npoutA'] >= 0
        np.testing.assert_array_equal(H['npstartA'][live], np.concatenate(([0], np.cumsum(H['npoutA'][live])[:-1])))
        assert int(H['npoutA'][live].sum()) == len(P['pos'])
        # every kept particle is one of its halo's subsample particles, none twice
        host = np.searchsorted(halos['id'], P['halo_id'].astype(np.uint64))
        assert mask[host].all()
        keyed = {tuple(r) for r in np.column_stack((host, P['pos'].view(np.uint32).reshape(len(host), 3))).tolist()}
        for j in np.unique(host)[:200]:
            a, nn = int(halos['npstartA'][j]), int(halos['npoutA'][j])
            sl = parts['pos'][a:a + nn]
            mine = P['pos'][host == j]
            assert len(mine) == want[j]
            assert all(any(np.array_equal(r, s) for s in sl) for r in mine[:5])
        assert len(keyed) >= len(host) - 5                                     # (bitwise duplicates of positions aside)
        np.testing.assert_array_equal(P['Np'], want[host].astype(np.float64))
        # rank columns: per halo a permutation of (r - mean) / mean, r = 0 .. k-1
        big = np.nonzero(want[mask] >= 5)[0][:50]
        for col in ('ranks', 'ranksv', 'ranksp', 'ranksr', 'ranksc'):
            for jj in big:
                s0, k = int(H['npstartA'][jj]), int(H['npoutA'][jj])
                mean = 0.5 * (k - 1)
                np.testing.assert_allclose(np.sort(P[col][s0:s0 + k]), (np.arange(k) - mean) / mean, rtol=0, atol=1e-12)
    # uniformity of the draw inside a slice: position of the kept particles within their halo's slice, over many halos
    H, P, mask = ps.prepare_slab_arrays(halos, parts, Mpart, h, True, want_ranks=False, want_AB=False, rng=7)
    sel = np.zeros(len(parts['pos']), dtype=bool)
    hostfull = np.repeat(np.arange(len(halos['N'])), halos['npoutA'])
    first = np.repeat(halos['npstartA'], halos['npoutA'])
    code = {tuple(r) for r in P['pos'].view(np.uint32).reshape(-1, 3).tolist()}
    sel = np.array([tuple(r) in code for r in parts['pos'].view(np.uint32).reshape(-1, 3).tolist()])
    frac = ((np.arange(len(sel)) - first) + 0.5) / np.repeat(halos['npoutA'], halos['npoutA'])
    partial = np.repeat((halos['npoutA'] >= 20) & mask & (want_mt(halos, Mpart) < halos['npoutA']), halos['npoutA'])
    f = frac[sel & partial]
    assert len(f) > 3000 and abs(f.mean() - 0.5) < 4 / np.sqrt(12 * len(f))
    hist = np.histogram(f, bins=10, range=(0, 1))[0]
    assert np.all(np.abs(hist - len(f) / 10) < 5 * np.sqrt(len(f) / 10))


def want_mt(halos, Mpart):
    m = halos['N'] * Mpart
    return np.array([po.particle_target(m[j], int(halos['npoutA'][j]), True) if halos['npoutA'][j] > 0 else 0 for j in range(len(m))])


@pytest.mark.parametrize('MT,want_ranks,want_AB,shear', [(True, True, True, True), (False, True, True, False), (True, False, False, False)])
def test_one_pass_slab_equals_the_column_by_column_path(MT, want_ranks, want_AB, shear, options):
    """rng = <seed>: the slab through HBM once (abacus_prepare_slab: inputs uploaded once or used in place as device arrays, kept
    rows gathered on the device, one copy per output column) against the column-by-column path (option prep_columnwise: the host
    gathers of hod/prepare_sim.py:984-1045's tables) - same Philox streams, so every column of both tables and the mask are EQUAL,
    key order and dtypes included; also with the CompaSO columns already in HBM (what the reader's unpack kernels leave there)"""
    from abacusutils_amd import _lib
    from abacusutils_amd.hod import prepare_sim as ps
    slabs, header = synth.synth_compaso_slabs(numslabs=1, n_halo=60000, seed=77, lbox=700.0)
    halos, parts = slabs[0]['halos'], slabs[0]['parts']
    Mpart, h = header['ParticleMassHMsun'], header['H0'] / 100.0
    kw = dict(want_ranks=want_ranks, want_AB=want_AB, shearmark=shearmark() if shear else None, Lbox=header['BoxSizeHMpc'], rng=4242,
              part_index0=2**33 + 11, halo_index0=2**32 + 3)
    options.set('prep_columnwise', 1)
    H0, P0, m0 = ps.prepare_slab_arrays(halos, parts, Mpart, h, MT, **kw)
    options.set('prep_columnwise', 0)
    _lib.profile_reset()
    _lib.profile_enable(True)
    H1, P1, m1 = ps.prepare_slab_arrays(halos, parts, Mpart, h, MT, **kw)
    _lib.profile_enable(False)
    assert 'prep_compact' in _lib.profile_get(), sorted(_lib.profile_get())      # the one-pass path ran
    dev_h = {k: _lib.DeviceArray(v) for k, v in halos.items()}
    dev_p = {k: _lib.DeviceArray(v) for k, v in parts.items()}
    H2, P2, m2 = ps.prepare_slab_arrays(dev_h, dev_p, Mpart, h, MT, **kw)
    for a in list(dev_h.values()) + list(dev_p.values()):
        a.free()
    assert len(P0['pos']) > 1000 and m0.sum() > 1000
    for H, P, m, label in ((H1, P1, m1, 'host columns'), (H2, P2, m2, 'device columns')):
        np.testing.assert_array_equal(m, m0, err_msg=label)
        for got, want in ((H, H0), (P, P0)):
            assert list(got) == list(want), (label, list(got), list(want))
            for k in want:
                assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, (label, k, got[k].dtype, want[k].dtype, got[k].shape, want[k].shape)
                np.testing.assert_array_equal(got[k], want[k], err_msg=f'{label}.{k}')


def test_prepare_to_run_hod_end_to_end():
    """three synthetic slabs -> prepare on the device (Philox selection, ranks, padded environments) -> AbacusHOD.from_prepared
    -> run_hod: the catalogue is the oracle's on the same staged arrays; the staged arrays carry what staging() derives
    (pweights, pinds, global fenv ranks)"""
    from abacusutils_amd.hod import prepare_sim as ps
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    slabs, header = synth.synth_compaso_slabs(numslabs=3, n_halo=20000, seed=51, lbox=600.0)
    Mpart, h, L = header['ParticleMassHMsun'], header['H0'] / 100.0, header['BoxSizeHMpc']
    HT, PT, env = [], [], []
    for i, s in enumerate(slabs):
        H, P, _ = ps.prepare_slab_arrays(s['halos'], s['parts'], Mpart, h, True, want_ranks=True, want_AB=True, rng=600 + i)
        HT.append(H)
        PT.append(P)
        env.append(ps.slab_environment(i, s['halos'], [slabs[(i - 1) % 3]['halos'], slabs[(i + 1) % 3]['halos']], 3, L, Mpart))
    env = tuple(np.concatenate([e[q] for e in env]) for q in range(3))
    hod = dict(tracer_flags={'LRG': True, 'ELG': True, 'QSO': False}, want_ranks=True, want_AB=True, want_shear=False,
               want_rsd=True, LRG_params=dict(synth.LRG_PARAMS, logM_cut=12.6, logM1=13.6, s=0.2, Acent=0.2, Bsat=-0.1),
               ELG_params=dict(synth.ELG_PARAMS, s_v=0.1, Bcent=0.1), QSO_params=synth.QSO_PARAMS)
    ball = AbacusHOD.from_prepared(HT, PT, header, 0.5, hod, env=env)
    hd, pd = ball.halo_data, ball.particle_data
    assert np.all(hd['hid'][:-1] <= hd['hid'][1:]) and np.array_equal(hd['hid'][pd['pinds']], pd['phid'])
    assert -0.5 <= hd['hfenv'].min() and hd['hfenv'].max() <= 0.5 and np.ptp(hd['hfenv']) > 0.9
    np.testing.assert_array_equal(pd['pfenv'], hd['hfenv'][pd['pinds']])
    mock = ball.run_hod()
    want = oracle.gen_gal_cat(hd, pd, ball.tracers, ball.params, Nthread=4, enable_ranks=True, rsd=True)
    assert_mock_equal(mock, want, exact=True)
    assert len(mock['LRG']['x']) > 500 and len(mock['ELG']['x']) > 500


def test_device_random_columns_match_the_oracle_bit_for_bit():
    """abacus_prepare_randoms (the `rng=<seed>` columns) against the oracle's restatement of the stream: Philox blocks pinned by
    the published vectors + the fixed float64 transforms; indices beyond 2^32, index lists and offsets (shard invariance)"""
    import ctypes as C

    from abacusutils_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    n, seed = 1500, 0x9abcdef012345678
    index = np.sort(rng.integers(0, 2**40, n)).astype(np.int64)
    scale = rng.uniform(10, 900, n)
    r, e, g = np.empty(n), np.empty((n, 3)), np.empty((n, 3))
    _lib.check(L.abacus_prepare_randoms(C.c_int64(n), _lib.ptr(index), C.c_int64(17), C.c_uint64(seed), 4, _lib.ptr(scale), _lib.ptr(r),
                                        _lib.ptr(e), _lib.ptr(g)))
    ro, eo, go = po.device_halo_randoms(seed, index + 17, scale)
    np.testing.assert_array_equal(r, ro)
    np.testing.assert_array_equal(e, eo)
    np.testing.assert_array_equal(g, go)
    for stream in (5, 6):
        u = np.empty(n)
        _lib.check(L.abacus_prepare_randoms(C.c_int64(n), None, C.c_int64(2**33 + 5), C.c_uint64(seed), stream, None, _lib.ptr(u), None, None))
        np.testing.assert_array_equal(u, po.device_uniform(seed, 2**33 + 5 + np.arange(n), stream))
    with pytest.raises(_lib.AbacusHipError):
        _lib.check(L.abacus_prepare_randoms(C.c_int64(n), None, C.c_int64(0), C.c_uint64(seed), 4, None, _lib.ptr(u), None, None))


def test_device_random_columns_distributions_and_shards():
    """10^6 halos: uniform, two-sided exponential and normal moments with the per-halo scale; a slab prepared with
    halo_index0 / part_index0 draws what the same objects draw inside a larger catalogue"""
    import ctypes as C

    from abacusutils_amd import _lib
    L = _lib.lib()
    n = 1_000_000
    scale = np.full(n, 250.0)
    r, e, g = np.empty(n), np.empty((n, 3)), np.empty((n, 3))
    _lib.check(L.abacus_prepare_randoms(C.c_int64(n), None, C.c_int64(0), C.c_uint64(77), 4, _lib.ptr(scale), _lib.ptr(r), _lib.ptr(e),
                                        _lib.ptr(g)))
    assert 0.0 <= r.min() and r.max() < 1.0 and abs(r.mean() - 0.5) < 2e-3 and abs(r.var() - 1 / 12) < 1e-3
    assert abs(np.abs(e).mean() / 250.0 - 1) < 5e-3 and abs((e > 0).mean() - 0.5) < 2e-3 and abs(e.mean()) < 1.0
    assert abs(g.std() / 250.0 - 1) < 3e-3 and abs(g.mean()) < 0.8
    cc = np.corrcoef(np.column_stack((r, e, g)).T)
    assert np.abs(cc - np.eye(7)).max() < 6e-3
    from scipy import stats
    assert stats.kstest(g.ravel()[:200000] / 250.0, 'norm').pvalue > 1e-3
    assert stats.kstest(np.abs(e).ravel()[:200000] / 250.0, 'expon').pvalue > 1e-3
    # shards: rows [a, b) drawn with index0 = a equal the slice of the whole
    a, b = 123456, 223456
    r2, e2, g2 = np.empty(b - a), np.empty((b - a, 3)), np.empty((b - a, 3))
    sc2 = scale[a:b].copy()       # (held: _lib.ptr does not keep its array alive)
    _lib.check(L.abacus_prepare_randoms(C.c_int64(b - a), None, C.c_int64(a), C.c_uint64(77), 4, _lib.ptr(sc2), _lib.ptr(r2),
                                        _lib.ptr(e2), _lib.ptr(g2)))
    np.testing.assert_array_equal(r2, r[a:b])
    np.testing.assert_array_equal(e2, e[a:b])
    np.testing.assert_array_equal(g2, g[a:b])
    from abacusutils_amd.hod import prepare_sim as ps
    slabs, header = synth.synth_compaso_slabs(numslabs=1, n_halo=5000, seed=47, lbox=300.0)
    halos, parts = slabs[0]['halos'], slabs[0]['parts']
    Mpart, h = header['ParticleMassHMsun'], header['H0'] / 100.0
    H, P, m = ps.prepare_slab_arrays(halos, parts, Mpart, h, True, rng=5, halo_index0=1000, part_index0=50000)
    kept = np.flatnonzero(m)
    ro, eo, go = po.device_halo_randoms(5, kept[:300] + 1000, np.asarray(halos['sigmav3d_L2com'])[kept[:300]] / np.sqrt(3))
    np.testing.assert_array_equal(H['randoms'][:300], ro)
    np.testing.assert_array_equal(H['randoms_exp'][:300], eo)
    np.testing.assert_array_equal(H['randoms_gaus_vrms'][:300], go)
    np.testing.assert_array_equal(m, po.device_uniform(5, 1000 + np.arange(len(m)), 6) < po.subsample_halos(halos['N'] * Mpart, True))


@pytest.mark.parametrize('MT', [False, True])
def test_subsample_halos_on_masses_like_the_reference(MT):
    """subsample_halos(m, MT) with the reference's two arguments evaluates the formula on the float64 MASSES
    (hod/prepare_sim.py:83-108) - masses up to 1e15 used to be cast to uint32 particle counts"""
    from abacusutils_amd.hod.prepare_sim import subsample_halos
    from oracle import prepare_oracle as po
    m = np.concatenate([10 ** np.random.default_rng(4).uniform(10.5, 15.2, 20000), [1e11, 1e12, 2.5e11, 4e11, 1e13, 9.99e12, 1e15]])
    np.testing.assert_allclose(subsample_halos(m, MT), po.subsample_halos(m, MT), rtol=1e-13, atol=1e-300)
    Mpart = 2.109081520453063e9
    np.testing.assert_allclose(subsample_halos(m, MT, Mpart), po.subsample_halos(np.rint(m / Mpart) * Mpart, MT), rtol=1e-13, atol=1e-300)


def _assert_close_like_the_reference(got, want, name):
    """tests/common.py assert_close: integers equal, floats numpy.testing.assert_allclose (rtol 1e-7)"""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if want.dtype.kind in 'iub':
        np.testing.assert_array_equal(got, want, err_msg=name)
    else:
        np.testing.assert_allclose(got, want, rtol=1e-7, err_msg=name)


MINI = None


def _mini():
    from conftest import GOLD
    return GOLD / 'Mini_N64_L32'


@pytest.mark.parametrize('slab', [2, 0, 1])
def test_prepare_slab_reproduces_the_reference_held_files(slab, tmp_path):
    """THE reference-held pin of prepare_sim (tests/test_hod.py:89-100): prepare_slab with the reference's signature on the
    Mini_N64_L32 CompaSO files (read by abacusutils_amd.data.compaso_halo_catalog, particles unpacked on the device) equals
    tests/ref_hod/Mini_N64_L32/z0.000/{halos,particles}_xcom_{slab}_seed600_abacushod_oldfenv_MT_new.h5 field by field
    (tests/golden/prepare_mini.npz) - slab 2 is the one the reference's test checks, 0 and 1 come from the same run"""
    from abacusutils_amd.hod import prepare_sim as PS
    g = load_golden('prepare_mini')
    H, P, env = PS.prepare_slab(slab, str(tmp_path), str(_mini()), 'Mini_N64_L32', 0.0, 'primary', {'LRG': True, 'ELG': True, 'QSO': False},
                                True, False, True, False, None, True, 600, numslabs=3, return_tables=True)
    for kind, tab in (('halos', H), ('particles', P)):
        names = [k.split('.', 2)[2] for k in g if k.startswith(f's{slab}.{kind}.')]
        assert set(names) == set(tab), (kind, set(names) ^ set(tab))
        for name in names:
            _assert_close_like_the_reference(tab[name], g[f's{slab}.{kind}.{name}'], f'{kind}.{name}')
    cid, cmass, menv = env
    assert len(cid) == len(cmass) == len(menv) >= len(H['id']) and np.all(np.isin(H['id'].astype(np.int64), cid))
    assert np.all(menv >= 0) and menv.max() > 0


def test_prepare_main_and_the_compaso_reader_on_the_device(tmp_path):
    """prepare_sim.main with the reference's signature on its own example configuration (tests/abacus_hod.yaml, paths
    redirected like tests/test_hod.py:52-85 does): the three slabs are prepared, and - where h5py is installed - the files of
    slab 2 equal the reference's.  And the whole-box catalogue, subsamples unpacked by the device kernels, equals the
    reference reader's (tests/golden/compaso_mini.npz, see tests/test_compaso_reader.py)"""
    import yaml
    from test_compaso_reader import check_catalogue

    from abacusutils_amd.hod import prepare_sim as PS
    check_catalogue('clean')
    check_catalogue('unclean')
    config = yaml.safe_load(open(_mini() / 'abacus_hod.yaml'))
    config['sim_params']['sim_dir'] = str(_mini()) + '/'
    config['sim_params']['subsample_dir'] = str(tmp_path / 'data_subs') + '/'
    try:
        import h5py
    except ImportError:
        with pytest.raises(ImportError, match='h5py'):
            PS.main(str(_mini() / 'abacus_hod.yaml'), params=config)
        return
    PS.main(str(_mini() / 'abacus_hod.yaml'), params=config)
    g = load_golden('prepare_mini')
    d = tmp_path / 'data_subs' / 'Mini_N64_L32' / 'z0.000'
    for kind, dset in (('halos', 'halos'), ('particles', 'particles')):
        a = h5py.File(d / f'{kind}_xcom_2_seed600_abacushod_oldfenv_MT_new.h5', 'r')[dset][:]
        for name in a.dtype.names:
            _assert_close_like_the_reference(a[name], g[f's2.{kind}.{name}'], f'{kind}.{name}')
    assert (d / 'env_xcom_1_abacushod_localenv_new.h5').exists()
