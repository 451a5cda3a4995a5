"""The oracle's restatement of prepare_sim.prepare_slab's data-parallel core (oracle/prepare_oracle.py) against golden
vectors of the REFERENCE's own prepare_slab (hod/prepare_sim.py:296-1052, run under the stand-ins of oracle/make_golden.py
on the seeded synthetic slabs of abacusutils_amd.synth.synth_compaso_slabs): every column of the halo and particle
datasets, bit for bit - the restatement consumes NumPy's global generator in the reference's order."""
import json

import numpy as np
import pytest
from conftest import load_golden

from abacusutils_amd import synth
from oracle import prepare_oracle as po

CASES = {'mt_ab': (1, True, False, True, False), 'lrg_ranks_ab': (0, False, True, True, False),
         'mt_ranks_ab_shear': (2, True, True, True, True)}


LC_CASES = {'lc_octant': ('octant', 0, True, False), 'lc_centre': ('centre', 2, False, True)}


def reference_seed(newseed, i):
    """(:349-351) the slab's NumPy seed; returns the seed of the light-cone randoms"""
    seeder = np.random.default_rng(newseed + i)
    np.random.seed(seeder.integers(0, 2**32 - 1))
    return seeder.integers(0, 2**32 - 1)


def shearmark(ndim=16, seed=5):
    return np.random.default_rng(seed).random((ndim, ndim, ndim))


@pytest.mark.parametrize('case', list(CASES))
def test_prepare_slab_core_reproduces_the_reference(case):
    g = load_golden('prepare_sim')
    slabs, header = synth.synth_compaso_slabs(**json.loads(str(g['meta.synth_json'])))
    assert header == json.loads(str(g['meta.header_json']))
    i, MT, want_ranks, want_AB, want_shear = CASES[case]
    reference_seed(600, i)
    with np.errstate(all='ignore'):
        H, P, mask = po.prepare_slab_core(slabs[i]['halos'], slabs[i]['parts'], header['ParticleMassHMsun'], header['H0'] / 100.0,
                                          MT, want_ranks=want_ranks, want_AB=want_AB, shearmark=shearmark() if want_shear else None,
                                          Lbox=header['BoxSizeHMpc'])
    for kind, got in (('halos', H), ('particles', P)):
        keys = sorted(k.split('.', 2)[2] for k in g if k.startswith(f'{case}.{kind}.'))
        assert sorted(got) == keys, (kind, sorted(got), keys)
        for k in keys:
            want = g[f'{case}.{kind}.{k}']
            assert got[k].shape == want.shape, (kind, k, got[k].shape, want.shape)
            np.testing.assert_array_equal(got[k], want, err_msg=f'{kind}.{k}')
    # consistency of the selection with the rules (:152-174, :871-884): kept counts, new offsets
    assert int(mask.sum()) == len(H['id'])
    live = H['npoutA'] >= 0
    assert np.array_equal(H['npstartA'][live], np.concatenate(([0], np.cumsum(H['npoutA'][live])[:-1])))
    assert int(H['npoutA'][live].sum()) == len(P['pos'])


@pytest.mark.parametrize('case', list(LC_CASES))
def test_lightcone_environment_reproduces_the_reference(case):
    """halo light cones (:474-616): the environment masses after the edge correction by randoms - the argument of the
    reference's calc_fenv_opt, recorded by oracle/make_golden.py - and then every column of the two tables"""
    g = load_golden('prepare_sim')
    geometry, i, MT, want_ranks = LC_CASES[case]
    slab, header = synth.synth_lightcone_slab(geometry=geometry, **json.loads(str(g['meta.lc_synth_json'])))
    halos, parts = slab['halos'], slab['parts']
    lc_seed = reference_seed(600, i)
    Mpart = header['ParticleMassHMsun']
    Menv, edge, norm = po.lightcone_menv(halos['x_L2com'], halos['N'] * Mpart, halos['r98_L2com'], header['BoxSizeHMpc'],
                                         header['LightConeOrigins'], lc_seed)
    want = g[f'{case}.Menv_corrected']
    assert 0.1 * len(want) < len(edge) < len(want) and (norm >= 0).all() and 0.2 < np.median(norm) < 1.2
    np.testing.assert_allclose(Menv, want, rtol=1e-13, atol=0)
    # the tables from the recorded masses: the mass sums of the restatement differ from the tree's in the last bits (order of
    # summation), which swaps the ranks of halos with (nearly) equal environments
    with np.errstate(all='ignore'):
        H, P, mask = po.prepare_slab_core(halos, parts, Mpart, header['H0'] / 100.0, MT, want_ranks=want_ranks, want_AB=True,
                                          Menv=want, Lbox=header['BoxSizeHMpc'], halo_lc=True)
    rename = {'id': 'index_halo', 'x_L2com': 'pos_interp', 'v_L2com': 'vel_interp', 'N': 'N_interp'}   # the loader's keys (:370-373)
    for k, alias in rename.items():
        H[alias] = H[k]
    for kind, got in (('halos', H), ('particles', P)):
        keys = sorted(k.split('.', 2)[2] for k in g if k.startswith(f'{case}.{kind}.'))
        assert sorted(got) == keys, (kind, sorted(got), keys)
        for k in keys:
            np.testing.assert_array_equal(got[k], g[f'{case}.{kind}.{k}'], err_msg=f'{kind}.{k}')


@pytest.mark.parametrize('geometry', ['octant', 'centre'])
def test_lightcone_host_logic_matches_the_oracle(geometry):
    """the host side of the product's light-cone environment (abacusutils_amd/hod/prepare_sim.py: the randoms of a round, the
    boxes they are cut to, the edge set - plain NumPy, no GPU involved) against the oracle's restatement, which the test
    above holds to the reference: same generator state in, same points and the same edge halos out"""
    from abacusutils_amd.hod import prepare_sim as ps
    slab, header = synth.synth_lightcone_slab(n_halo=1200, seed=77, geometry=geometry)
    pos = slab['halos']['x_L2com']
    Lbox = header['BoxSizeHMpc']
    origins = np.asarray(header['LightConeOrigins']).reshape(-1, 3)
    dist = np.sqrt(np.sum((pos - origins[0]) ** 2.0, axis=1))
    r_min, r_max = dist.min(), dist.max()
    for chi_max in (r_max, Lbox):                         # the second reaches beyond the first box: three boxes for the octant
        a, ad = ps._lightcone_randoms(5000, r_min, chi_max, Lbox, 10.0, origins, np.random.default_rng(5))
        b, bd = po.lightcone_shell_randoms(5000, r_min, chi_max, Lbox, 10.0, origins, np.random.default_rng(5))
        assert 100 < len(a) <= 5000
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(ad, bd)
    _, edge_o, _ = po.lightcone_menv(pos, slab['halos']['N'] * header['ParticleMassHMsun'], slab['halos']['r98_L2com'], Lbox,
                                     origins, 3)
    inside = ps._lightcone_interior(pos, dist, float(Lbox), ps.LC_OFFSET, origins, 10, r_min, r_max)
    np.testing.assert_array_equal(np.flatnonzero(~inside), edge_o)
    assert 0 < len(edge_o) < len(pos)
    with pytest.raises(ValueError, match='origins'):
        ps._lightcone_cuboids(Lbox, 10.0, np.zeros((2, 3)), r_max)


def test_rows_gathers_like_fancy_indexing():
    from abacusutils_amd.hod.prepare_sim import _rows
    r = np.random.default_rng(0)
    for a in (r.random((50, 3), dtype=np.float32), r.random(50), r.random((50, 3)), np.zeros((0, 3), dtype=np.float32),
              r.random((50, 3))[:, ::-1], r.integers(0, 9, (50, 2)), r.random((50, 2, 2))):
        for idx in (np.array([3, 1, 1, 49]), np.array([], dtype=np.int64)):
            if len(a) == 0 and len(idx):
                continue
            got, want = _rows(a, idx), a[idx]
            assert got.dtype == want.dtype and got.shape == want.shape
            np.testing.assert_array_equal(got, want)


def test_device_stream_restatement_is_sane():
    """the oracle's restatement of the device's random columns (bit-compared with the kernels in tests/test_prepare_gpu.py):
    ranges, rough moments, independence of index order"""
    from oracle import prepare_oracle as po
    idx = np.arange(4000) + 2**35
    r, e, g = po.device_halo_randoms(9, idx, np.full(4000, 100.0))
    assert 0 <= r.min() and r.max() < 1 and abs(r.mean() - 0.5) < 0.03
    assert abs(np.abs(e).mean() / 100 - 1) < 0.05 and abs((e > 0).mean() - 0.5) < 0.03
    assert abs(g.std() / 100 - 1) < 0.04 and abs(g.mean()) < 4
    r2, e2, g2 = po.device_halo_randoms(9, idx[::-1], np.full(4000, 100.0))
    np.testing.assert_array_equal(r2[::-1], r)
    np.testing.assert_array_equal(g2[::-1], g)
    u = po.device_uniform(9, idx, 5)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.03 and not np.array_equal(u, po.device_uniform(9, idx, 6))
