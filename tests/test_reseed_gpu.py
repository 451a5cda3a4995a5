"""`abacus_hod_reseed` on the device == the oracle's restatement of the same stream, bit for bit:
1e6 halos + particles, shard offsets, both velocity-deviate laws; and `AbacusHOD.run_hod(reseed=...)` end to end."""
import numpy as np
import pytest

from abacusutils_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('expvel', [False, True])
@pytest.mark.parametrize('h0,p0', [(0, 0), (123_457, 6), (2**33 + 5, 2**34 + 3)])
def test_device_stream_equals_oracle_bitwise(expvel, h0, p0):
    from abacusutils_amd.hod.GRAND_HOD import StagedCatalog
    nh, npart = 1_000_000, 1_000_003
    hd, pd, params = synth.synth_hod_inputs(nh, npart, seed=5)
    st = StagedCatalog(hd, pd)
    sig = 250.0 + 100.0 * np.random.default_rng(2).random(nh)
    st.reseed(0x1234ABCD5678, hsigma3d=sig, want_expvel=expvel, halo_index0=h0, part_index0=p0)
    hr, hv, pr = oracle.reseed(0x1234ABCD5678, nh, npart, hsigma3d=sig, want_expvel=expvel, halo_index0=h0, part_index0=p0)
    np.testing.assert_array_equal(st.fetch_field('hrandoms'), hr)
    np.testing.assert_array_equal(st.fetch_field('hveldev'), hv)
    np.testing.assert_array_equal(st.fetch_field('prandoms'), pr)
    st.free()


def test_run_hod_reseed_matches_the_oracle_end_to_end():
    """run_hod(reseed=s): the three columns the reference rewrites (hod/abacus_hod.py:824-835) come out of the device
    generator exactly as the oracle draws them, with the reference's dtypes (float32 randoms, float64 hveldev), and the
    catalogue populated from them is the oracle's catalogue"""
    from abacusutils_amd.hod.abacus_hod import AbacusHOD
    from conftest import assert_mock_equal
    hd, pd, params = synth.synth_hod_inputs(200_000, 300_000, seed=7)
    hod = dict(tracer_flags={'LRG': True, 'ELG': True, 'QSO': False}, want_ranks=False, want_AB=True, want_shear=False,
               want_rsd=True, LRG_params=synth.LRG_PARAMS, ELG_params=synth.ELG_PARAMS, QSO_params=synth.QSO_PARAMS)
    ball = AbacusHOD.from_arrays(hd, pd, params, hod)
    mock = ball.run_hod(reseed=424242)
    hr, hv, pr = oracle.reseed(424242, 200_000, 300_000, hsigma3d=hd['hsigma3d'])
    assert ball.halo_data['hrandoms'].dtype == np.float32 and ball.particle_data['prandoms'].dtype == np.float32
    np.testing.assert_array_equal(ball.halo_data['hrandoms'], hr.astype(np.float32))
    np.testing.assert_array_equal(ball.halo_data['hveldev'], hv)
    np.testing.assert_array_equal(ball.particle_data['prandoms'], pr.astype(np.float32))
    want = oracle.gen_gal_cat(dict(hd, hrandoms=hr, hveldev=hv), dict(pd, prandoms=pr), ball.tracers, params, Nthread=4)
    assert_mock_equal(mock, want, exact=True)
