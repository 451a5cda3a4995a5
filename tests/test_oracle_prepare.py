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


def reference_seed(newseed, i):
    """(:349-350) the slab's NumPy seed"""
    seeder = np.random.default_rng(newseed + i)
    np.random.seed(seeder.integers(0, 2**32 - 1))


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
