import numpy as np, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from abacusutils_amd import _lib
from abacusutils_amd.analysis.tsc import tsc_parallel
n, box, ng = 100_000_000, 2000.0, 1024
pos = np.random.default_rng(300).random((n, 3), dtype=np.float32) * np.float32(box)
_lib.set_option('tsc_oldlists', 1)
b = tsc_parallel(pos, ng, box).astype('f8')
_lib.set_option('tsc_oldlists', 0)
for name, opt in (('new32', {}), ('new64', {'tsc_acc64': 1})):
    for k, v in opt.items(): _lib.set_option(k, v)
    a = tsc_parallel(pos, ng, box)
    for k in opt: _lib.set_option(k, 0)
    d = a.astype('f8') - b
    print(name, 'mass diff', d.sum(), 'max', np.abs(d).max(), flush=True)
    for ax in range(3):
        prof = d.sum(axis=tuple(i for i in range(3) if i != ax))
        print('  axis', ax, 'first128', prof[:128].sum(), 'rest', prof[128:].sum(), 'blocks of 128:', np.round(prof.reshape(8, 128).sum(axis=1), 3))
    lo = d[:128].sum() ; print('  x<128 slab', lo)
