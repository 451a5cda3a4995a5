"""`AbacusHOD.staging()` (file plumbing, reference abacusnbody/hod/abacus_hod.py:263-702) against the staged arrays of
tests/golden/hod_mini.npz, which oracle/make_golden.py built from the same prepare_sim HDF5 fixtures with its own
HDF5 reader.  Needs h5py and the reference's Mini_N64_L32 fixtures, so it only runs in the build container (an
interpreter with h5py is looked up; the main one has none) and is skipped elsewhere."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REF_TESTS = Path('/root/reference/tests')

_SCRIPT = r'''
import sys, yaml, numpy as np, h5py
from pathlib import Path
sys.path.insert(0, sys.argv[1])
from abacusutils_amd.hod import abacus_hod
from abacusutils_amd.hod.abacus_hod import AbacusHOD
if sys.argv[4] == 'numpy-standins':
    # no GPU in the build container: the three device helpers of staging() (abacus_argsort_i64, abacus_searchsorted_i64,
    # abacus_fenv_rank) are replaced by the NumPy expressions of the reference (abacus_hod.py:566-588,1961-1970) so that the
    # file plumbing can be held to the golden arrays here; tests/test_staging_gpu.py runs the real helpers
    def _fenv(Menv, mbins, halosM):
        out = np.zeros(len(Menv))
        for ib in range(len(mbins) - 1):
            m = (halosM > mbins[ib]) & (halosM < mbins[ib + 1])
            if m.sum() > 1:
                out[m] = Menv[m].argsort(kind='stable').argsort(kind='stable') / (m.sum() - 1) - 0.5
        return out
    abacus_hod._searchsorted = lambda a, b: np.searchsorted(a, b).astype(np.int64)
    abacus_hod._argsort_ids = lambda a: np.argsort(a, kind='stable')
    abacus_hod.calc_fenv_opt = _fenv
ref, tmp = Path(sys.argv[2]), Path(sys.argv[3])
cfg = yaml.safe_load(open(ref / 'abacus_hod.yaml'))
cfg['sim_params'].update(sim_dir=str(ref) + '/', subsample_dir=str(ref / 'ref_hod') + '/', output_dir=str(tmp / 'out'))
g = np.load(Path(sys.argv[1]) / 'tests/golden/hod_mini.npz')

# (1) no assembly bias: nothing but the halo / particle subsample files is read
cfg['HOD_params']['want_AB'] = False
ball = AbacusHOD(cfg['sim_params'], cfg['HOD_params'], cfg['clustering_params'])
nchk = 0
for k in g.files:
    tab, _, name = k.partition('.')
    src = {'h': ball.halo_data, 'p': ball.particle_data, 'params': ball.params}.get(tab)
    if src is None or name not in src or src[name] is None:
        continue
    a, b = np.asarray(src[name]), g[k]
    assert a.shape == b.shape and np.array_equal(a, b), k
    nchk += 1
assert nchk >= 26, nchk
assert 'hdeltac' not in ball.halo_data and 'pfenv' not in ball.particle_data

# (2) assembly bias: missing env sidecars are an error, like the reference (abacus_hod.py:607-612)
cfg['HOD_params']['want_AB'] = True
try:
    AbacusHOD(cfg['sim_params'], cfg['HOD_params'], cfg['clustering_params'])
    raise SystemExit('expected FileNotFoundError')
except FileNotFoundError:
    pass

# (3) with sidecars (synthesised: id / mass / Menv per slab) the global fenv rank is mapped onto halos by id
sub = tmp / 'sub' / 'Mini_N64_L32' / 'z0.000'
sub.mkdir(parents=True)
src = ref / 'ref_hod' / 'Mini_N64_L32' / 'z0.000'
for f in src.glob('*.h5'):
    (sub / f.name).symlink_to(f)
rng = np.random.default_rng(5)
hid = g['h.hid']
parts = np.array_split(rng.permutation(len(hid)), 3)
for i, sel in enumerate(parts):
    with h5py.File(sub / f'env_xcom_{i}_abacushod_localenv_new.h5', 'w') as f:
        f['id'] = hid[sel]
        f['mass'] = g['h.hmass'][sel]
        f['Menv'] = g['h.hmass'][sel] * rng.uniform(0.5, 20, len(sel))
cfg['sim_params']['subsample_dir'] = str(tmp / 'sub') + '/'
ball = AbacusHOD(cfg['sim_params'], cfg['HOD_params'], cfg['clustering_params'])
hf, pf = ball.halo_data['hfenv'], ball.particle_data['pfenv']
assert np.array_equal(ball.halo_data['hdeltac'], g['h.hdeltac'])
assert np.array_equal(ball.particle_data['pdeltac'], g['p.pdeltac'])
assert hf.shape == hid.shape and np.all((hf >= -0.5) & (hf <= 0.5))
# independent restatement of the per-mass-bin rank (abacus_hod.py:1961-1970): rank of Menv among halos of the bin
Menv = np.empty(len(hid)); bins = np.logspace(11, 15.5, 101)
for i, sel in enumerate(parts):
    with h5py.File(sub / f'env_xcom_{i}_abacushod_localenv_new.h5', 'r') as f:
        Menv[sel] = f['Menv'][:]
mass = g['h.hmass']
ib = np.searchsorted(bins, mass, side='left') - 1          # bins[ib] < mass < bins[ib+1] (no mass sits on an edge)
assert not np.any(np.isin(mass, bins))
want = np.zeros(len(hid))
for b in np.unique(ib[(ib >= 0) & (ib < 100)]):
    m = np.nonzero(ib == b)[0]
    if len(m) > 1:
        order = np.argsort(Menv[m], kind='stable')
        rk = np.empty(len(m)); rk[order] = np.arange(len(m))
        want[m] = rk / (len(m) - 1) - 0.5
assert np.array_equal(hf, want), np.abs(hf - want).max()
assert hf.std() > 0.1
assert np.array_equal(pf, hf[ball.particle_data['pinds']])
print('STAGING-OK', nchk)
'''


def _python_with_h5py():
    for exe in (sys.executable, '/opt/conda/bin/python3.9', shutil.which('python3.9')):
        if exe and os.path.exists(exe):
            r = subprocess.run([exe, '-c', 'import h5py, yaml, numpy'], capture_output=True)
            if r.returncode == 0:
                return exe
    return None


def test_staging_matches_golden(tmp_path):
    if not (REF_TESTS / 'ref_hod' / 'Mini_N64_L32').exists():
        pytest.skip('reference HOD fixtures not present on this machine')
    exe = _python_with_h5py()
    if exe is None:
        pytest.skip('no interpreter with h5py')
    r = subprocess.run([exe, '-c', _SCRIPT, str(ROOT), str(REF_TESTS), str(tmp_path), 'numpy-standins'], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and 'STAGING-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
