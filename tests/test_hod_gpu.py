"""HIP HOD path (libabacus_hip.so through the C ABI) vs the CPU oracle, the reference's fixtures and the
golden vectors.  Needs an MI355X: run with `-m gpu`."""
import numpy as np
import pytest
from conftest import (SYNTH_CASES, assert_mock_equal, load_golden, synth_case, unpack_inputs, unpack_mock)

from abacusutils_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def G():
    from abacusutils_amd.hod import GRAND_HOD
    return GRAND_HOD


@pytest.mark.parametrize('name', ['hod_mini', 'hod_lc'])
def test_reference_fixture(G, name):
    """tests/ref_hod/**/galaxies_rsd/*.dat of the reference (tests/test_hod.py:109-134, test_lc_hod.py)"""
    g = load_golden(name)
    hd, pd, params = unpack_inputs(g)
    tracers = {'LRG': synth.LRG_PARAMS, 'ELG': synth.ELG_PARAMS}
    mock = G.gen_gal_cat(hd, pd, tracers, params, Nthread=4, enable_ranks=False, rsd=True)
    assert_mock_equal(mock, unpack_mock(g, 'expect'), exact=False, rtol=1e-14)
    assert_mock_equal(mock, unpack_mock(g, 'shim'), exact=True)


@pytest.mark.parametrize('name', SYNTH_CASES)
def test_synthetic_golden(G, name):
    """bit-exact against what the reference returned on the same seeded inputs"""
    g = load_golden('hod_synth_' + name)
    hd, pd, params, tracers, ranks, rsd = synth_case(g)
    mock = G.gen_gal_cat(hd, pd, tracers, params, enable_ranks=ranks, rsd=rsd)
    assert_mock_equal(mock, unpack_mock(g, 'expect'), exact=True)


@pytest.mark.parametrize('nh,npart', [(1, 1), (2047, 2049), (2048, 4096), (100003, 250007), (0, 0), (5000, 0)])
def test_vs_oracle_ragged_sizes(G, nh, npart):
    """tile edges (2048 objects / workgroup), odd tails, empty inputs: keep masks and catalogs bit-equal"""
    from oracle import oracle
    hd, pd, params = synth.synth_hod_inputs(max(nh, 1), max(npart, 1), seed=11, with_ranks=True)
    hd = {k: v[:nh] for k, v in hd.items()}
    pd = {k: v[:npart] for k, v in pd.items()}
    if npart:
        pd['pinds'] = np.minimum(pd['pinds'], max(nh - 1, 0))
    if nh == 0:
        pd = {k: v[:0] for k, v in pd.items()}
    tracers = {'LRG': dict(synth.LRG_PARAMS, logM_cut=12.5, logM1=13.5, alpha_c=0.2, alpha_s=0.9, s=0.1),
               'ELG': dict(synth.ELG_PARAMS, Ccent=0.05, logM1_EE=13.0, alpha_EL=1.2),
               'QSO': synth.QSO_PARAMS}
    st = G.StagedCatalog(hd, pd)
    p = G.marshal_params(tracers, params, True, True)
    st.populate(p)
    kc, ks = st.fetch_keep()
    mock = {tr: st.fetch(tr) for tr in tracers}
    want, wkc, wks = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=4, enable_ranks=True, rsd=True,
                                        return_keep=True)
    np.testing.assert_array_equal(kc, wkc)
    np.testing.assert_array_equal(ks, wks)
    assert_mock_equal(mock, want, exact=True)
    st.free()


def test_full_size_c2_bit_exact(G):
    """BASELINE config 2: 10^7 halos + 10^7 particles, LRG HOD of tests/abacus_hod.yaml:31-47, fixed seed:
    galaxy counts, keep masks and every output array bit-equal to the CPU oracle; stable order."""
    from oracle import oracle
    n = 10_000_000
    hd, pd, params = synth.synth_hod_inputs(n, n, seed=600)
    tracers = {'LRG': synth.LRG_PARAMS}
    st = G.StagedCatalog(hd, pd)
    p = G.marshal_params(tracers, params, False, True)
    ncent, nsat = st.populate(p)
    kc, ks = st.fetch_keep()
    mock = {'LRG': st.fetch('LRG')}
    # repeat populate on the resident catalog (MCMC pattern): identical result
    st.populate(p)
    again = {'LRG': st.fetch('LRG')}
    assert_mock_equal(again, mock, exact=True)
    st.free()
    want, wkc, wks = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=oracle.max_threads(), enable_ranks=False,
                                        rsd=True, return_keep=True)
    assert ncent[0] == want['LRG']['Ncent'] and ncent[0] + nsat[0] == len(want['LRG']['x'])
    np.testing.assert_array_equal(kc, wkc)
    np.testing.assert_array_equal(ks, wks)
    assert_mock_equal(mock, want, exact=True)
    # size-independent properties: ids of centrals strictly increasing (stable compaction of hid = 1000*arange),
    # satellites' host ids non-decreasing, RSD keeps z inside the box
    idc = mock['LRG']['id'][: ncent[0]]
    assert np.all(np.diff(idc) > 0)
    assert np.all(np.diff(mock['LRG']['id'][ncent[0]:]) >= 0)
    L = params['Lbox']
    assert np.all((mock['LRG']['z'] >= -L / 2 - 1.0) & (mock['LRG']['z'] < L / 2 + 1.0))


def test_full_size_c5_multi_tracer_bit_exact(G):
    """BASELINE config 5's HOD at C2 size (the `hod_multi` leg of bench.py): 10^7 halos + 10^7 particles, LRG + ELG + QSO
    with assembly bias, rank modulation, velocity bias and ELG conformity (synth.PRODUCTION_TRACERS) - the envelope
    table of the two-stage filter, the conformity variants bounded ahead of the exact central pass - against the CPU
    oracle: keep masks, counts and all eight columns of the three catalogues bit-equal"""
    from oracle import oracle
    n = 10_000_000
    hd, pd, params = synth.synth_hod_inputs(n, n, seed=600, with_ranks=True)
    tracers = synth.PRODUCTION_TRACERS
    st = G.StagedCatalog(hd, pd)
    p = G.marshal_params(tracers, params, True, True)
    ncent, nsat = st.populate(p)
    kc, ks = st.fetch_keep()
    mock = {tr: st.fetch(tr) for tr in tracers}
    st.free()
    want, wkc, wks = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=oracle.max_threads(), enable_ranks=True,
                                        rsd=True, return_keep=True)
    np.testing.assert_array_equal(kc, wkc)
    np.testing.assert_array_equal(ks, wks)
    for t, tr in enumerate(('LRG', 'ELG', 'QSO')):
        assert ncent[t] == want[tr]['Ncent'] and ncent[t] + nsat[t] == len(want[tr]['x'])
        assert ncent[t] > 1000 and nsat[t] > 100     # every branch populated
    assert_mock_equal(mock, want, exact=True)


def test_c4_per_gpu_shard_bit_exact(G):
    """BASELINE config 4 hands every GPU an eighth of 3e8 halos: 3.75e7 halos + 3.75e7 particles on ONE device (the keys,
    queues, superblock counters and 64-bit offsets at that size; a filter stream of 0.3 GB), LRG + ELG with conformity,
    against the CPU oracle: keep masks, counts and all columns bit-equal"""
    from oracle import oracle
    n = 37_500_000
    hd, pd, params = synth.synth_hod_inputs(n, n, seed=604)
    tracers = {'LRG': synth.LRG_PARAMS, 'ELG': dict(synth.ELG_PARAMS, logM1_EE=13.2, alpha_EE=0.9, logM1_EL=13.8, alpha_EL=1.1)}
    st = G.StagedCatalog(hd, pd)
    p = G.marshal_params(tracers, params, False, True)
    ncent, nsat = st.populate(p)
    kc, ks = st.fetch_keep()
    mock = {tr: st.fetch(tr) for tr in tracers}
    st.free()
    want, wkc, wks = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=oracle.max_threads(), enable_ranks=False, rsd=True,
                                        return_keep=True)
    np.testing.assert_array_equal(kc, wkc)
    np.testing.assert_array_equal(ks, wks)
    assert ncent[0] > 100000 and ncent[1] > 1000000 and nsat[1] > 10000
    assert_mock_equal(mock, want, exact=True)


def test_capacity_growth_and_param_change(G):
    """first populate emits few galaxies, second many more than the catalog buffers hold: buffers grow, order kept"""
    from oracle import oracle
    hd, pd, params = synth.synth_hod_inputs(300000, 300000, seed=3)
    st = G.StagedCatalog(hd, pd)
    for logM_cut in (14.5, 11.2):
        tracers = {'LRG': dict(synth.LRG_PARAMS, logM_cut=logM_cut, logM1=logM_cut + 0.8)}
        p = G.marshal_params(tracers, params, False, True)
        st.populate(p)
        got = {'LRG': st.fetch('LRG')}
        want = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=4)
        assert_mock_equal(got, want, exact=True)
    st.free()


def test_keep_masks_across_a_sequence_of_populates(G, options):
    """one staged catalogue, many populates: the key filter of a sparse (LRG-only) mix leaves the keep masks alone and
    hod_exact un-keeps what the populate before it kept; a change of the mix (other superblock size), a reseed, the NFW
    path or the comparator filter in between must never leave a stale byte.  Masks and catalogues against the oracle after
    every step"""
    from oracle import oracle
    hd, pd, params = synth.synth_hod_inputs(250_000, 350_000, seed=21, with_ranks=True)
    st = G.StagedCatalog(hd, pd)
    lrg = lambda lc: {'LRG': dict(synth.LRG_PARAMS, logM_cut=lc, logM1=lc + 0.9)}   # noqa: E731
    steps = [('lrg', lrg(12.6), False), ('lrg', lrg(12.9), False), ('lrg', lrg(12.3), False), ('mix', synth.PRODUCTION_TRACERS, True),
             ('lrg', lrg(12.7), False), ('lrg', lrg(12.7), False), ('nokeys', lrg(12.5), False), ('lrg', lrg(13.0), False),
             ('reseed', lrg(12.8), False), ('lrg', lrg(12.4), False), ('nolazy', lrg(12.6), False), ('lrg', lrg(12.6), False)]
    for what, tracers, ranks in steps:
        options.set('hod_nokeys', 1 if what == 'nokeys' else 0)
        options.set('hod_nolazy', 1 if what == 'nolazy' else 0)
        if what == 'reseed':
            st.reseed(1234, hsigma3d=hd['hsigma3d'])
            hd = dict(hd, hrandoms=st.fetch_field('hrandoms'), hveldev=st.fetch_field('hveldev').reshape(-1, 3))
            pd = dict(pd, prandoms=st.fetch_field('prandoms'))
        p = G.marshal_params(tracers, params, ranks, True)
        st.populate(p)
        kc, ks = st.fetch_keep()
        mock = {tr: st.fetch(tr) for tr in tracers}
        want, wkc, wks = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=4, enable_ranks=ranks, rsd=True, return_keep=True)
        np.testing.assert_array_equal(kc, wkc, err_msg=what)
        np.testing.assert_array_equal(ks, wks, err_msg=what)
        assert_mock_equal(mock, want, exact=True)
    st.free()


def test_errors(G):
    g = load_golden('hod_mini')
    hd, pd, params = unpack_inputs(g)
    with pytest.raises(ValueError):
        G.gen_gal_cat(hd, pd, {'LRG': synth.LRG_PARAMS}, params, rsd=1)
    bad = dict(synth.LRG_PARAMS)
    del bad['alpha_c']
    with pytest.raises(KeyError):
        G.gen_gal_cat(hd, pd, {'LRG': bad}, params)
    with pytest.raises(ValueError):
        G.gen_gal_cat(hd, pd, {'LRG': synth.LRG_PARAMS}, params, nfw=True)


def test_write_to_disk(G, tmp_path):
    g = load_golden('hod_mini')
    hd, pd, params = unpack_inputs(g)
    tracers = {'LRG': synth.LRG_PARAMS, 'ELG': synth.ELG_PARAMS}
    G.gen_gal_cat(hd, pd, tracers, params, write_to_disk=True, savedir=tmp_path)
    txt = (tmp_path / 'galaxies_rsd' / 'LRGs.dat').read_text().splitlines()
    assert txt[0] == '# %ECSV 1.0' and any('Ncent: 7' in line for line in txt)
    rows = [line.split() for line in txt if not line.startswith('#')][1:]
    np.testing.assert_array_equal(np.array([int(r[7]) for r in rows]), g['expect.LRG.id'])
    np.testing.assert_array_equal(np.array([float(r[2]) for r in rows]), g['expect.LRG.z'])


# ---- device Philox reseed (abacus_hod_reseed) ---------------------------------------------------------------
def _stage_small(nh=200_000, npart=300_000, seed=5):
    from abacusutils_amd.hod import GRAND_HOD as G
    hd, pd, params = synth.synth_hod_inputs(nh, npart, seed=seed)
    return G, hd, pd, params, G.StagedCatalog(hd, pd)


def test_reseed_distributions_and_determinism():
    from scipy import stats
    G, hd, pd, params, st = _stage_small()
    st.reseed(1234, hsigma3d=hd['hsigma3d'])
    r1, v1, p1 = st.fetch_field('hrandoms'), st.fetch_field('hveldev'), st.fetch_field('prandoms')
    st.reseed(1234)
    np.testing.assert_array_equal(st.fetch_field('hrandoms'), r1)      # deterministic
    np.testing.assert_array_equal(st.fetch_field('hveldev'), v1)
    st.reseed(1235)
    assert np.mean(st.fetch_field('hrandoms') == r1) < 1e-3            # another seed, another stream
    for u in (r1, p1):
        assert u.min() >= 0.0 and u.max() < 1.0
        assert np.all(u == u.astype(np.float32))                       # float32 draws, as the reference's dtype
        assert stats.kstest(u, 'uniform').pvalue > 1e-4
        assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.01
    z = v1 / (hd['hsigma3d'][:, None] / np.sqrt(3))                    # hveldev = N(0,1) * hsigma3d / sqrt(3)
    assert stats.kstest(z.ravel(), 'norm').pvalue > 1e-4
    assert np.abs(np.corrcoef(z.T) - np.eye(3)).max() < 0.01
    assert abs(np.corrcoef(r1, z[:, 0])[0, 1]) < 0.01
    st.reseed(7, want_expvel=True)                                     # two-sided exponential (:799-801)
    ze = st.fetch_field('hveldev') / (hd['hsigma3d'][:, None] / np.sqrt(3))
    assert stats.kstest(ze.ravel(), 'laplace').pvalue > 1e-4
    st.free()


def test_reseed_is_sharding_invariant_and_feeds_populate():
    """a shard reseeded with its global index offsets draws what the whole catalogue draws; populate after a device
    reseed equals the oracle run on the fetched arrays (bit-exact)"""
    from abacusutils_amd.hod import shard
    G, hd, pd, params, st = _stage_small(60_000, 90_000)
    st.reseed(99, hsigma3d=hd['hsigma3d'])
    full = {k: st.fetch_field(k) for k in ('hrandoms', 'hveldev', 'prandoms')}
    h0 = p0 = 0
    for rank in range(3):
        h, p = shard.shard_catalog(hd, pd, rank, 3)
        if rank:
            h0 += nh_prev
            p0 += np_prev
        nh_prev, np_prev = len(h['hmass']), len(p['phmass'])
        s2 = G.StagedCatalog(h, p)
        s2.reseed(99, hsigma3d=h['hsigma3d'], halo_index0=h0, part_index0=p0)
        np.testing.assert_array_equal(s2.fetch_field('hrandoms'), full['hrandoms'][h0:h0 + len(h['hmass'])])
        np.testing.assert_array_equal(s2.fetch_field('hveldev'), full['hveldev'][h0:h0 + len(h['hmass'])])
        np.testing.assert_array_equal(s2.fetch_field('prandoms'), full['prandoms'][p0:p0 + len(p['phmass'])])
        s2.free()
    tracers = {'LRG': dict(synth.LRG_PARAMS), 'ELG': dict(synth.ELG_PARAMS)}
    got = G.gen_gal_cat(hd, pd, tracers, params, rsd=True, staged=st)
    hd2, pd2 = dict(hd), dict(pd)
    hd2['hrandoms'], hd2['hveldev'], pd2['prandoms'] = full['hrandoms'], full['hveldev'], full['prandoms']
    from oracle import oracle
    ref = oracle.gen_gal_cat(hd2, pd2, tracers, params, Nthread=4, rsd=True)
    for tr in tracers:
        assert got[tr]['Ncent'] == ref[tr]['Ncent'] and len(ref[tr]['x']) > 100
        for k in ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass', 'id'):
            np.testing.assert_array_equal(got[tr][k], ref[tr][k])
    st.free()


@pytest.mark.parametrize('seed', range(28))     # more seeds: scripts/gpu_hod_fuzz.sh
def test_random_parameter_sweep_bit_exact(G, seed):
    """seeded random HOD parameters over the ranges an MCMC explores (incl. assembly bias, conformity, velocity bias,
    rank parameters, incompleteness, tracer subsets, light-cone RSD; tests/sweep.py): catalogues and keep masks
    bit-equal to the oracle (which tests/test_oracle_hod.py holds to the shimmed reference on the same cases)"""
    from oracle import oracle
    from sweep import sweep_case
    hd, pd, params, tracers, ranks, rsd = sweep_case(seed)
    st = G.StagedCatalog(hd, pd)
    st.populate(G.marshal_params(tracers, params, ranks, rsd))
    kc, ks = st.fetch_keep()
    mock = {tr: st.fetch(tr) for tr in tracers}
    want, wkc, wks = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=4, enable_ranks=ranks, rsd=rsd, return_keep=True)
    np.testing.assert_array_equal(kc, wkc)
    np.testing.assert_array_equal(ks, wks)
    assert sum(len(m['x']) for m in want.values()) > 0
    assert_mock_equal(mock, want, exact=True)
    st.free()


def test_stale_pinds_rejected_at_staging():
    """keep_cent[pinds] is gathered on the device: an index outside [0, n_halo) (pinds kept after sub-selecting the
    halos) is refused when the catalogue is staged, with nothing leaked (a second staging of the same size works)"""
    from abacusutils_amd import _lib, synth
    from abacusutils_amd.hod.GRAND_HOD import StagedCatalog
    hd, pd, _ = synth.synth_hod_inputs(5000, 8000, seed=9)
    bad = dict(pd, pinds=pd['pinds'].copy())
    bad['pinds'][1234] = 5000
    with pytest.raises(_lib.AbacusHipError, match='pinds'):
        StagedCatalog(hd, bad)
    bad['pinds'][1234] = -1
    with pytest.raises(_lib.AbacusHipError, match='pinds'):
        StagedCatalog(hd, bad)
    st = StagedCatalog(hd, pd)
    st.free()
