"""Fork safety of the drop-in (the reference is used under emcee / multiprocessing pools with
NUMBA_THREADING_LAYER=forksafe, docs/hod.rst:226-240): the library initialises HIP lazily, on the first compute call of a
PROCESS, so a parent that has imported the package and dlopen'ed libabacus_hip.so - but not computed - can fork workers that
each open the GPU themselves.  Run in a fresh interpreter (the test process itself has long initialised HIP)."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent

SCRIPT = r'''
import multiprocessing as mp, sys
import numpy as np
sys.path.insert(0, %r)
from abacusutils_amd import _lib, synth
from abacusutils_amd.hod.GRAND_HOD import gen_gal_cat
from abacusutils_amd.analysis.power_spectrum import calc_power
from oracle import oracle

_lib.lib()                                   # dlopen before the fork; no HIP call yet
hd, pd, params = synth.synth_hod_inputs(60000, 90000, seed=600, with_ranks=False)
tracers = {'LRG': dict(synth.LRG_PARAMS), 'ELG': dict(synth.ELG_PARAMS)}
want = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=2, enable_ranks=False, rsd=True)
pos = synth.synth_positions(50000, 500.0, seed=3, clustered=True)
kw = dict(kbins=8, mubins=2, paste='TSC', nmesh=64, poles=[0, 2])
pk = oracle.calc_power(pos.copy(), 500.0, nthread=2, accum64=True, **kw)

def work(tag):
    got = gen_gal_cat(hd, pd, tracers, params, enable_ranks=False, rsd=True)
    for tr in tracers:
        assert got[tr]['Ncent'] == want[tr]['Ncent'], tag
        for c in ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass', 'id'):
            assert np.array_equal(got[tr][c], want[tr][c]), (tag, tr, c)
    tab = calc_power(pos.copy(), 500.0, **kw)
    assert np.array_equal(np.asarray(tab['N_mode']), pk['N_mode']), tag
    ok = pk['N_mode'] > 0
    assert np.abs(np.asarray(tab['power'])[ok] / pk['power'][ok] - 1).max() < 1e-5, tag

def child(q, tag):
    try:
        work(tag)
        q.put((tag, 'ok'))
    except BaseException as e:
        q.put((tag, repr(e)))

ctx = mp.get_context('fork')
q = ctx.Queue()
procs = [ctx.Process(target=child, args=(q, t)) for t in ('w0', 'w1')]
for p in procs: p.start()
res = dict(q.get(timeout=240) for _ in procs)
for p in procs: p.join(60)
assert res == {'w0': 'ok', 'w1': 'ok'}, res
work('parent-after-fork')                    # the parent opens the GPU only now
print('FORK-OK')
'''


def test_fork_before_first_call():
    r = subprocess.run([sys.executable, '-c', SCRIPT % str(REPO)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'FORK-OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
