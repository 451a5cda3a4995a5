"""AbacusHOD class on the MI355X path: the reference's own mini-box fixture through run_hod (C1 plumbing config),
the MCMC call pattern of scripts/hod/run_hod.py:40-73, compute_power / compute_xirppi / compute_wp / compute_ngal.
Needs an MI355X: run with `-m gpu`."""
import numpy as np
import pytest
from conftest import assert_mock_equal, load_golden, unpack_inputs, unpack_mock

from abacusutils_amd import synth

pytestmark = pytest.mark.gpu

HOD_PARAMS = dict(tracer_flags={'LRG': True, 'ELG': True, 'QSO': False}, want_ranks=False, want_AB=True,
                  want_shear=False, want_rsd=True, LRG_params=synth.LRG_PARAMS, ELG_params=synth.ELG_PARAMS,
                  QSO_params=synth.QSO_PARAMS)
CLUSTERING = dict(clustering_type='xirppi', pimax=30, pi_bin_size=5,
                  bin_params=dict(logmin=-0.7728787904780005, logmax=1.4771212597864314, nbins=9))


def _ball(name='hod_mini', tmp_path='./'):
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    g = load_golden(name)
    hd, pd, params = unpack_inputs(g)
    return AbacusHOD.from_arrays(hd, pd, params, HOD_PARAMS, CLUSTERING, mock_dir=tmp_path), g


def test_run_hod_reference_fixture(tmp_path):
    """tests/test_hod.py:102-134 of the reference: write_to_disk run, catalogs equal to galaxies_rsd/*.dat"""
    ball, g = _ball('hod_mini', tmp_path)
    assert ball.want_rsd and set(ball.tracers) == {'LRG', 'ELG'} and ball.rpbins.shape == (10,)
    mock = ball.run_hod(ball.tracers, ball.want_rsd, write_to_disk=False, Nthread=4)
    assert_mock_equal(mock, unpack_mock(g, 'shim'), exact=True)
    ball.run_hod(ball.tracers, ball.want_rsd, write_to_disk=True, Nthread=4)
    back = ball.gal_reader()
    for tr in ('LRG', 'ELG'):
        np.testing.assert_array_equal(back[tr]['id'], g[f'expect.{tr}.id'])
        np.testing.assert_array_equal(back[tr]['x'], g[f'expect.{tr}.x'])
        assert back[tr]['Ncent'] == int(g[f'expect.{tr}.Ncent'])


def test_lightcone_fixture():
    ball, g = _ball('hod_lc')
    mock = ball.run_hod()
    assert_mock_equal(mock, unpack_mock(g, 'shim'), exact=True)


def test_mcmc_pattern_and_reseed():
    """resident catalog, parameters mutated between calls (scripts/hod/run_hod.py:62-73); reseed smoke
    (tests/test_hod.py:136-143) - the reseeded run is deterministic for a given seed and changes the catalog"""
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    from oracle import oracle
    hd, pd, params = synth.synth_hod_inputs(200000, 200000, seed=8)
    ball = AbacusHOD.from_arrays(hd, pd, params, dict(HOD_PARAMS, tracer_flags={'LRG': True, 'ELG': False}))
    for logM_cut in (12.8, 13.0, 13.3):
        ball.tracers['LRG'] = dict(ball.tracers['LRG'], logM_cut=logM_cut)
        mock = ball.run_hod(ball.tracers, want_rsd=True, Nthread=16)
        want = oracle.gen_gal_cat(ball.halo_data, ball.particle_data, ball.tracers, ball.params, Nthread=4)
        assert_mock_equal(mock, want, exact=True)
    a = ball.run_hod(reseed=0xABCDEF)
    assert ball.halo_data['hrandoms'].dtype == np.float32
    want = oracle.gen_gal_cat(ball.halo_data, ball.particle_data, ball.tracers, ball.params, Nthread=4)
    assert_mock_equal(a, want, exact=True)    # device copy of the randoms was refreshed
    b = ball.run_hod(reseed=0xABCDEF)
    assert_mock_equal(a, b, exact=True)
    assert len(a['LRG']['x']) != len(mock['LRG']['x']) or not np.array_equal(a['LRG']['id'], mock['LRG']['id'])
    with pytest.raises(ValueError):
        ball.run_hod(want_rsd=1)


def test_secondary_redshift_needs_nfw():
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    hd, pd, params = synth.synth_hod_inputs(1000, 1000, seed=8)
    ball = AbacusHOD.from_arrays(hd, pd, params, HOD_PARAMS, z_type='secondary')
    with pytest.raises(RuntimeError):
        ball.run_hod()


def test_compute_power_and_clustering():
    from abacusutils_amd.analysis.power_spectrum import calc_power
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    hd, pd, params = synth.synth_hod_inputs(300000, 300000, seed=9, lbox=1000.0)
    hod = dict(HOD_PARAMS, LRG_params=dict(synth.LRG_PARAMS, logM_cut=12.3, logM1=13.3),
               ELG_params=dict(synth.ELG_PARAMS))
    ball = AbacusHOD.from_arrays(hd, pd, params, hod, CLUSTERING)
    mock = ball.run_hod()
    # positions live in [-L/2, L/2): both estimators must cope (periodic)
    spec = ball.compute_power(mock, nbins_k=8, nbins_mu=2, k_hMpc_max=0.2, logk=False, poles=[0, 2], num_cells=64)
    assert set(spec) >= {'LRG_LRG', 'LRG_ELG', 'ELG_LRG', 'ELG_ELG', 'LRG_LRG_ell', 'LRG_ELG_modes', 'k_binc', 'mu_binc'}
    assert spec['LRG_LRG'].shape == (8, 2) and spec['LRG_ELG_ell'].shape == (8, 2)
    pos = np.stack((mock['LRG']['x'], mock['LRG']['y'], mock['LRG']['z']), axis=1)
    tab = calc_power(pos, 1000.0, 8, 2, 0.2, False, 'TSC', 64, False, False, poles=[0, 2])
    np.testing.assert_allclose(spec['LRG_LRG'], tab['power'], rtol=1e-6)
    with pytest.raises(KeyError):   # the reference reads power['poles'] unconditionally (abacus_hod.py:1431)
        ball.compute_power(mock, 8, 2, 0.2, False, poles=[], num_cells=32)
    xi = ball.compute_clustering(mock, ball.rpbins, ball.pimax, ball.pi_bin_size)
    assert xi['LRG_LRG'].shape == (9, 6) and np.array_equal(xi['LRG_ELG'], xi['ELG_LRG'])
    wp = ball.compute_wp(mock, ball.rpbins, ball.pimax, ball.pi_bin_size)
    assert wp['ELG_ELG'].shape == (9,)
    ngal, fsat = ball.compute_ngal()
    n_lrg = len(mock['LRG']['x'])
    assert abs(ngal['LRG'] / n_lrg - 1) < 0.15 and 0 <= fsat['LRG'] <= 1


def test_clustering_from_the_catalogue_in_hbm_matches_the_host_path():
    """run_hod's mock_dict remembers its columns in HBM (GRAND_HOD.MockDict): compute_power (every field deposited and
    transformed once, all pairs binned on the device), compute_xirppi / compute_wp / compute_multipole (abacus_paircount_dev
    on the float64 columns) give what the same calls give on plain host dicts; the device path is dropped as soon as the
    catalogue is stale (a later run_hod) or the host columns were touched"""
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    hd, pd, params = synth.synth_hod_inputs(300000, 300000, seed=19, lbox=1000.0)
    hod = dict(HOD_PARAMS, LRG_params=dict(synth.LRG_PARAMS, logM_cut=12.3, logM1=13.3), ELG_params=dict(synth.ELG_PARAMS))
    ball = AbacusHOD.from_arrays(hd, pd, params, hod, CLUSTERING)
    mock = ball.run_hod()
    assert all(mock.device_xyz(tr) is not None for tr in mock)
    plain = {tr: dict(mock[tr]) for tr in mock}                      # a plain dict: the host path
    sbins = np.linspace(0.5, 30.0, 8)
    got = dict(power=ball.compute_power(mock, 8, 2, 0.2, False, poles=[0, 2], num_cells=64, compensated=True, interlaced=True),
               xirppi=ball.compute_xirppi(mock, ball.rpbins, ball.pimax, ball.pi_bin_size),
               wp=ball.compute_wp(mock, ball.rpbins, ball.pimax, ball.pi_bin_size),
               multi=ball.compute_multipole(mock, ball.rpbins, ball.pimax, sbins, 10))
    want = dict(power=ball.compute_power(plain, 8, 2, 0.2, False, poles=[0, 2], num_cells=64, compensated=True, interlaced=True),
                xirppi=ball.compute_xirppi(plain, ball.rpbins, ball.pimax, ball.pi_bin_size),
                wp=ball.compute_wp(plain, ball.rpbins, ball.pimax, ball.pi_bin_size),
                multi=ball.compute_multipole(plain, ball.rpbins, ball.pimax, sbins, 10))
    for stat in ('xirppi', 'wp', 'multi'):                           # integer pair counts: identical
        assert set(got[stat]) == set(want[stat])
        for k in want[stat]:
            np.testing.assert_array_equal(got[stat][k], want[stat][k], err_msg=f'{stat} {k}')
    assert set(got['power']) == set(want['power'])
    for k in want['power']:
        if k.endswith('modes'):
            np.testing.assert_array_equal(got['power'][k], want['power'][k])
        else:
            np.testing.assert_allclose(got['power'][k], want['power'][k], rtol=2e-6, atol=1e-6 * np.abs(want['power'][k]).max(), err_msg=k)
    # staleness: ANY in-place edit of a host column is seen (whole-column checksum against the device column), also one
    # that touches a single row or swaps two rows
    z = mock['LRG']['z']
    z[len(z) // 3] += 1e-9
    assert mock.device_xyz('LRG') is None and mock.device_xyz('ELG') is not None
    y = mock['ELG']['y']
    y[[5, 7]] = y[[7, 5]]
    assert mock.device_xyz('ELG') is None
    y[[5, 7]] = y[[7, 5]]
    assert mock.device_xyz('ELG') is not None
    mock2 = ball.run_hod()
    assert mock.device_xyz('ELG') is None and mock2.device_xyz('ELG') is not None
    # a mock travels like the reference's plain dict: pickle (multiprocessing / emcee pools, np.save)
    import pickle
    back = pickle.loads(pickle.dumps(mock2))
    assert type(back) is dict and set(back) == set(mock2)
    for tr in mock2:
        assert back[tr]['Ncent'] == mock2[tr]['Ncent']
        for c in ('x', 'vz', 'mass', 'id'):
            np.testing.assert_array_equal(back[tr][c], mock2[tr][c])


def test_lazy_columns():
    """AbacusHOD.lazy_columns: run_hod returns with the columns still in HBM; clustering runs from there; the first read
    copies all eight columns and gives exactly what the eager call returns; a never-read mock refuses to be read after a
    later run_hod replaced it"""
    import pickle

    from abacusutils_amd.hod.GRAND_HOD import LazyTracer
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    hd, pd, params = synth.synth_hod_inputs(300000, 300000, seed=19, lbox=1000.0)
    hod = dict(HOD_PARAMS, LRG_params=dict(synth.LRG_PARAMS, logM_cut=12.3, logM1=13.3), ELG_params=dict(synth.ELG_PARAMS))
    ball = AbacusHOD.from_arrays(hd, pd, params, hod, CLUSTERING)
    eager = ball.run_hod()
    wp_eager = ball.compute_wp(eager, ball.rpbins, ball.pimax, ball.pi_bin_size)
    ball.lazy_columns = True
    lazy = ball.run_hod()
    assert all(isinstance(lazy[tr], LazyTracer) and '_staged' in lazy[tr].__dict__ for tr in lazy)   # nothing copied yet
    assert lazy['LRG']['Ncent'] == eager['LRG']['Ncent'] and 'x' in lazy['LRG'] and set(lazy['LRG']) == set(eager['LRG'])
    wp_lazy = ball.compute_wp(lazy, ball.rpbins, ball.pimax, ball.pi_bin_size)                        # from HBM
    assert all('_staged' in lazy[tr].__dict__ for tr in lazy)
    for k in wp_eager:
        np.testing.assert_array_equal(wp_lazy[k], wp_eager[k])
    for tr in eager:                                                                                  # first read
        for c in ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass', 'id'):
            np.testing.assert_array_equal(lazy[tr][c], eager[tr][c], err_msg=f'{tr} {c}')
        assert dict(lazy[tr]).keys() == eager[tr].keys()
    assert lazy.device_xyz('LRG') is not None
    lazy2 = ball.run_hod()
    back = pickle.loads(pickle.dumps(lazy2))                   # pickling reads the columns
    np.testing.assert_array_equal(back['ELG']['x'], eager['ELG']['x'])
    lazy3 = ball.run_hod()
    ball.run_hod()
    with pytest.raises(RuntimeError, match='never read'):
        lazy3['LRG']['x']
    np.testing.assert_array_equal(lazy2['LRG']['x'], eager['LRG']['x'])   # read in time: still whole


def test_compute_ngal_device_vs_numpy():
    """AbacusHOD.compute_ngal on the device (sum over halos) against the oracle's NumPy restatement of the reference's
    sums over the 100^3 / 100^4 histograms (hod/abacus_hod.py:861-1179)"""
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    from oracle import oracle
    hd, pd, params = synth.synth_hod_inputs(300000, 1000, seed=12)
    hp = dict(HOD_PARAMS, tracer_flags={'LRG': True, 'ELG': True, 'QSO': True})
    ball = AbacusHOD.from_arrays(hd, pd, params, hp)
    cases = [{'LRG': dict(ball.tracers['LRG'], Acent=0.3, Asat=-0.2, Bcent=0.1, Bsat=0.4, logM_cut_pr=0.5, z_pivot=0.8),
              'ELG': dict(ball.tracers['ELG'], Acent=0.2, Bsat=-0.3, Ccent=0.5, Csat=0.25, logM1_EE=13.0, alpha_EE=0.8),
              'QSO': dict(ball.tracers['QSO'], Bcent=-0.4, Asat=0.3, ic=0.7)}]
    for tracers in cases:
        ngal, fsat = ball.compute_ngal(tracers)
        ngal_ref, fsat_ref = oracle.compute_ngal_numpy(ball, tracers)
        for tr in tracers:
            assert ngal_ref[tr] > 10
            np.testing.assert_allclose(ngal[tr], ngal_ref[tr], rtol=1e-11)
            np.testing.assert_allclose(fsat[tr], fsat_ref[tr], rtol=1e-11)
    # the expectation describes the mocks: central LRG count of a run_hod within 5 sigma (+2 % for the histogram's
    # cell-centre approximation); the satellites of this catalogue come from a 1000-particle subsample, not compared
    mock = ball.run_hod({'LRG': ball.tracers['LRG']})
    ngal, fsat = ball.compute_ngal({'LRG': ball.tracers['LRG']})
    ncent = ngal['LRG'] * (1 - fsat['LRG'])
    assert abs(mock['LRG']['Ncent'] - ncent) < 5 * np.sqrt(ncent) + 0.02 * ncent


def test_compute_ngal_vs_reference_golden():
    """AbacusHOD.compute_ngal on the device against the REFERENCE's compute_ngal (abacus_hod.py:861-1179, run under the
    shim by oracle/make_golden.py ngal on a 12-cell-per-dimension histogram): three parameter cases, including an
    evolving ELG whose conformity parameters take compute_ngal's own (raw) defaults"""
    import json

    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    g = load_golden('ngal')
    nh = len(g['hmass'])
    hd, pd, params = synth.synth_hod_inputs(nh, 10, seed=77)
    for k in ('hmass', 'hdeltac', 'hfenv', 'hshear', 'hmultis'):
        np.testing.assert_array_equal(hd[k], g[k])          # the generator of the golden inputs has not drifted
    hp = dict(HOD_PARAMS, tracer_flags={'LRG': True, 'ELG': True, 'QSO': True})
    ball = AbacusHOD.from_arrays(hd, pd, params, hp)
    nb = int(g['nbin'])
    ball.logMbins = np.linspace(np.log10(np.min(hd['hmass'])), np.log10(np.max(hd['hmass'])), nb + 1)
    ball.deltacbins = ball.fenvbins = ball.shearbins = np.linspace(-0.5, 0.5, nb + 1)
    for name, tracers in json.loads(str(g['cases_json'])).items():
        ngal, fsat = ball.compute_ngal(tracers)
        for t in tracers:
            np.testing.assert_allclose(ngal[t], float(g[f'{name}.{t}.ngal']), rtol=1e-11, err_msg=f'{name} {t}')
            np.testing.assert_allclose(fsat[t], float(g[f'{name}.{t}.fsat']), rtol=1e-11, err_msg=f'{name} {t}')
