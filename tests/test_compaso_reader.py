"""The CompaSO / ASDF reader (abacusutils_amd/data/asdf.py, compaso_halo_catalog.py) on the reference's Mini_N64_L32
simulation (its test DATA, copied to tests/golden/Mini_N64_L32 by oracle/make_golden.py mini) against the catalogues the
REAL reference reader produced from the same files and keeps for its own tests (tests/ref_data/test_halos_clean.asdf,
test_halos_unclean.asdf, test_subsamples_clean.asdf, test_subsamples_unclean.asdf; tests/test_data.py:29-158) ->
tests/golden/compaso_mini.npz.  Host logic only: the bit unpacking is the oracle's here, the device kernels' in
tests/test_prepare_gpu.py."""
import numpy as np
import pytest
from conftest import GOLD, load_golden

from oracle import oracle

SIM = GOLD / 'Mini_N64_L32' / 'Mini_N64_L32' / 'halos' / 'z0.000'


@pytest.fixture
def host_unpackers(monkeypatch):
    from abacusutils_amd.data import bitpacked

    def rv(intdata, boxsize, float_dtype=np.float32, posout=None, velout=None):
        return oracle.unpack_rvint(intdata, boxsize, float_dtype)

    def pids(packed, box=None, ppd=None, float_dtype=np.float32, **which):
        full = oracle.unpack_pids(packed, box, ppd, float_dtype)
        return {k: full[k] for k, on in which.items() if on}

    monkeypatch.setattr(bitpacked, 'unpack_rvint', rv)
    monkeypatch.setattr(bitpacked, 'unpack_pids', pids)


def check_catalogue(tag, subsamples=True):
    from abacusutils_amd.data.compaso_halo_catalog import CompaSOHaloCatalog
    g = load_golden('compaso_mini')
    cols = [k.split('.', 1)[1] for k in g if k.startswith(f'halos_{tag}.')]
    fields = [c for c in cols if c not in ('N_merge', 'is_merged_to', 'haloindex') or tag == 'clean']
    cat = CompaSOHaloCatalog(SIM, cleaned=(tag == 'clean'), subsamples=dict(A=True, B=True, rv=True, pid=True) if subsamples else False,
                             fields=fields)
    assert len(cat.halos) == len(g[f'halos_{tag}.id']) == 381
    for c in fields:
        want = g[f'halos_{tag}.{c}']
        got = cat.halos[c]
        assert got.shape == want.shape, c
        if c.startswith(('npstart', 'npout')) and not subsamples:
            continue                                  # re-indexed only when the particles are loaded
        if np.issubdtype(want.dtype, np.integer):
            np.testing.assert_array_equal(got, want, err_msg=c)
        else:
            assert got.dtype == want.dtype, c
            np.testing.assert_allclose(got, want, rtol=1e-7, err_msg=c)      # the reference's assert_close uses rtol 1e-7
    if subsamples:
        for c in ('pos', 'vel', 'pid'):
            want = g[f'subsamples_{tag}.{c}']
            got = cat.subsamples[c]
            assert got.shape == want.shape and (c == 'pid' or got.dtype == want.dtype), c
            if c == 'pid':
                np.testing.assert_array_equal(got, want)
            else:
                np.testing.assert_allclose(got, want, rtol=1e-7)
        assert len(cat.subsamples['pos']) == int(np.sum(cat.halos['npoutA'], dtype=np.int64) + np.sum(cat.halos['npoutB'], dtype=np.int64))
    return cat


@pytest.mark.parametrize('tag', ['clean', 'unclean'])
def test_catalogue_equals_the_reference_readers(tag, host_unpackers):
    cat = check_catalogue(tag)
    assert cat.header['SimName'] == 'Mini_N64_L32' and cat.header['cleaned_halos'] == (tag == 'clean')
    if tag == 'clean':   # tests/test_data.py:69-75
        assert np.all(cat.halos['is_merged_to'][cat.halos['N'] == 0] != -1)
        np.testing.assert_array_equal(cat.halos['N_merge'][cat.halos['N'] == 0], 0)


def test_one_slab_file_filter_and_subset(host_unpackers):
    """a single halo_info file with its subsample-A particles (what prepare_slab loads, hod/prepare_sim.py:396-418), a
    filter_func (tests/test_data.py:242-255) and the argument checks"""
    from abacusutils_amd.data.compaso_halo_catalog import CompaSOHaloCatalog
    g = load_golden('compaso_mini')
    fn = SIM / 'halo_info' / 'halo_info_002.asdf'
    cat = CompaSOHaloCatalog(fn, subsamples=dict(A=True, rv=True), cleaned=True,
                             fields=['N', 'x_L2com', 'v_L2com', 'r90_L2com', 'r25_L2com', 'r98_L2com', 'npstartA', 'npoutA', 'id', 'sigmav3d_L2com'])
    n2 = len(cat.halos)
    assert n2 == 100 and list(cat.subsamples) == ['pos', 'vel']
    np.testing.assert_array_equal(cat.halos['id'], g['halos_clean.id'][-n2:])          # slab 2 closes the whole-box table
    np.testing.assert_array_equal(cat.halos['N'], g['halos_clean.N'][-n2:])
    np.testing.assert_allclose(cat.halos['r98_L2com'], g['halos_clean.r98_L2com'][-n2:], rtol=1e-7)
    # the slab's particles, halo by halo, are the whole-box catalogue's
    full_start, full_out = g['halos_clean.npstartA'][-n2:], g['halos_clean.npoutA'][-n2:]
    np.testing.assert_array_equal(cat.halos['npoutA'], full_out)
    for j in np.flatnonzero(full_out)[:20]:
        a = cat.subsamples['pos'][int(cat.halos['npstartA'][j]):int(cat.halos['npstartA'][j]) + int(full_out[j])]
        b = g['subsamples_clean.pos'][int(full_start[j]):int(full_start[j]) + int(full_out[j])]
        np.testing.assert_allclose(a, b, rtol=1e-7)
    kept = cat.halos[cat.halos['N'] > 0]
    assert 0 < len(kept) <= n2 and set(kept.colnames) == set(cat.halos.colnames)
    big = CompaSOHaloCatalog(SIM, cleaned=True, fields=['N', 'x_L2com'], filter_func=lambda h: h['N'] > 100)
    assert len(big.halos) == int((g['halos_clean.N'] > 100).sum()) and big.halos['N'].min() > 100
    with pytest.raises(ValueError):
        CompaSOHaloCatalog(fn, subsamples=dict(A=True, rv=True, pos=True))
    with pytest.raises(KeyError):
        CompaSOHaloCatalog(fn, fields=['sigmar_eigenvecsMaj_L2com'])
    with pytest.raises(FileNotFoundError):
        CompaSOHaloCatalog(fn, cleaned=True, cleandir=GOLD / 'no_such_dir')
