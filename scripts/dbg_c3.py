import numpy as np, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from abacusutils_amd import _lib
from abacusutils_amd.analysis.power_spectrum import calc_power
from oracle import oracle
n, box, nmesh = 100_000_000, 2000.0, 1024
pos = np.random.default_rng(300).random((n, 3), dtype=np.float32) * np.float32(box)
kw = dict(kbins=512, mubins=4, k_max=np.pi * nmesh / box + 1e-6, paste='TSC', nmesh=nmesh, poles=[0, 2, 4], compensated=False, interlaced=False)
t = time.time(); b = oracle.calc_power(pos, box, nthread=oracle.max_threads(), accum64=True, **kw); print('oracle', time.time() - t, flush=True)
ok = b['N_mode'] > 0
for name, opts in (('default', {}), ('acc64', {'tsc_acc64': 1}), ('oldlists', {'tsc_oldlists': 1})):
    for k, v in opts.items(): _lib.set_option(k, v)
    a = calc_power(pos, box, **kw)
    for k in opts: _lib.set_option(k, 0)
    rel = np.abs(np.asarray(a['power'])[ok] / b['power'][ok] - 1)
    i = np.unravel_index(np.argmax(np.where(ok, np.abs(np.asarray(a['power']) / np.where(ok, b['power'], 1) - 1), 0)), ok.shape)
    pol = np.abs(np.asarray(a['poles']) - b['poles']).max() / np.abs(b['power']).max()
    print(name, 'max rel power', rel.max(), 'at bin', i, 'N_mode', b['N_mode'][i], 'n>1e-5:', int((rel > 1e-5).sum()), 'median', np.median(rel), 'poles/scale', pol, flush=True)
