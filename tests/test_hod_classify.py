"""The float32 interval classifier of the fused HOD kernel (abacusutils_amd/csrc/hod_classify.hpp), compiled for the host
(tests/native/classify_host.cpp, g++), against the CPU oracle's exact keep masks: whenever the classifier decides, its
decision must be the reference's; the undecided fraction (objects whose random lies inside a marker's band, which the
kernel sends through the float64 chain) must stay small.  CPU only - no GPU, no libabacus_hip.so compute."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest
from sweep import sweep_case

from abacusutils_amd import synth
from abacusutils_amd.hod.GRAND_HOD import marshal_params

HERE = Path(__file__).resolve().parent / 'native'


@pytest.fixture(scope='module')
def cls():
    so, src = HERE / 'libclassify_host.so', HERE / 'classify_host.cpp'
    hdr = HERE.parents[1] / 'abacusutils_amd' / 'csrc' / 'hod_classify.hpp'
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.check_call(['g++', '-O2', '-fopenmp', '-ffp-contract=off', '-shared', '-fPIC', '-o', str(so), str(src)])
    return C.CDLL(str(so))


def _p(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.c_void_p)


def _classify(cls, hd, pd, p, keep_c):
    nh, npart = len(hd['hmass']), len(pd['phmass'])
    oc, os_ = np.empty(nh, np.int8), np.empty(npart, np.int8)
    arrs = [np.ascontiguousarray(hd[k], dtype=np.float64) for k in ('hmass', 'hmultis', 'hrandoms')]
    opt = [np.ascontiguousarray(hd[k], dtype=np.float64) if k in hd else None for k in ('hdeltac', 'hfenv', 'hshear')]
    cls.cls_cent(C.byref(p), C.c_int64(nh), *[_p(a) for a in arrs], *[_p(a) for a in opt], oc.ctypes.data_as(C.c_void_p))
    parr = [np.ascontiguousarray(pd[k], dtype=np.float64) for k in ('phmass', 'pweights', 'prandoms')]
    popt = [np.ascontiguousarray(pd[k], dtype=np.float64) if k in pd else None
            for k in ('pdeltac', 'pfenv', 'pshear', 'pranks', 'pranksv', 'pranksp', 'pranksr')]
    kc = np.ascontiguousarray(keep_c[pd['pinds']], dtype=np.int8)
    cls.cls_sat(C.byref(p), C.c_int64(npart), *[_p(a) for a in parr], *[_p(a) for a in popt],
                kc.ctypes.data_as(C.c_void_p), os_.ctypes.data_as(C.c_void_p))
    return oc, os_


def _check(cls, hd, pd, params, tracers, enable_ranks, max_undecided=2e-3):
    from oracle import oracle
    _, kc, ks = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=4, enable_ranks=enable_ranks, rsd=True, return_keep=True)
    p = marshal_params(tracers, params, enable_ranks, True)
    oc, os_ = _classify(cls, hd, pd, p, kc)
    for got, want, what in ((oc, kc, 'centrals'), (os_, ks, 'satellites')):
        dec = got >= 0
        bad = np.flatnonzero(dec & (got != want))
        assert bad.size == 0, (what, bad[:5], got[bad[:5]], want[bad[:5]])
        assert (~dec).mean() <= max_undecided, (what, (~dec).mean())
    return (oc < 0).mean(), (os_ < 0).mean()


@pytest.mark.parametrize('seed', range(40))
def test_classifier_never_contradicts_the_oracle_sweep(cls, seed):
    """the 40 seeded parameter sets of the HOD sweep (assembly bias, conformity, ranks, all tracer subsets)"""
    hd, pd, params, tracers, enable_ranks, _ = sweep_case(seed)
    _check(cls, hd, pd, params, tracers, enable_ranks)


def test_classifier_production_mix_and_lrg(cls):
    hd, pd, params = synth.synth_hod_inputs(400000, 400000, seed=600, with_ranks=True)
    u1 = _check(cls, hd, pd, params, synth.PRODUCTION_TRACERS, True)
    u2 = _check(cls, hd, pd, params, {'LRG': synth.LRG_PARAMS}, False)
    assert max(u1 + u2) < 5e-4


@pytest.mark.parametrize('case', ['narrow_sigma', 'negative_elg', 'neg_weights', 'float32_randoms', 'tiny_markers', 'alpha_zero'])
def test_classifier_edge_cases(cls, case):
    """parameter corners: very narrow erfc transitions, a negative ELG amplitude (p_max < 1/Q: the chain is not monotone),
    negative / zero multiplicities and weights, float32-quantised randoms including exact zeros (what `reseed` draws),
    occupations deep in the erfc tail, alpha = 0"""
    rng = np.random.default_rng(5)
    hd, pd, params = synth.synth_hod_inputs(200000, 200000, seed=77, with_ranks=True)
    tracers = {k: dict(v) for k, v in synth.PRODUCTION_TRACERS.items()}
    lim = 2e-3
    if case == 'narrow_sigma':
        tracers['LRG']['sigma'] = 0.004
        tracers['QSO']['sigma'] = 0.01
        tracers['ELG']['sigma'] = 0.02
        lim = 2e-2
    elif case == 'negative_elg':
        tracers['ELG']['p_max'] = 0.001
        tracers['ELG']['Q'] = 20.0
        tracers['LRG']['s'] = -3.0          # 1 + s * rank changes sign
    elif case == 'neg_weights':
        hd['hmultis'][::7] = 0.0
        hd['hmultis'][3::11] = -1.0
        pd['pweights'][::5] = 0.0
        pd['pweights'][1::13] = -0.5
        hd['hrandoms'][::14] = 0.0
        lim = 0.1      # a zero random sits inside every band around a zero marker
    elif case == 'float32_randoms':
        hd['hrandoms'] = rng.random(len(hd['hrandoms']), dtype=np.float32).astype(np.float64)
        pd['prandoms'] = rng.random(len(pd['prandoms']), dtype=np.float32).astype(np.float64)
        hd['hrandoms'][::1000] = 0.0
        pd['prandoms'][::1000] = 0.0
    elif case == 'tiny_markers':
        tracers = {'LRG': dict(synth.LRG_PARAMS, logM_cut=15.2, sigma=0.05), 'QSO': dict(synth.QSO_PARAMS, logM_cut=15.4, sigma=0.06)}
        hd['hrandoms'] = hd['hrandoms'] * 10.0 ** rng.uniform(-40, 0, len(hd['hrandoms']))
        pd['prandoms'] = pd['prandoms'] * 10.0 ** rng.uniform(-40, 0, len(pd['prandoms']))
        lim = 0.9      # randoms spread over 40 decades sit inside the absolute floor of the bands by construction
    elif case == 'alpha_zero':
        tracers['QSO']['alpha'] = 0.0
        tracers['ELG']['alpha_EE'] = 0.0
        tracers['ELG']['kappa'] = 0.0
    _check(cls, hd, pd, params, tracers, True, max_undecided=lim)
