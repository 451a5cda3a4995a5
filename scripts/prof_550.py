import sys, time
sys.path.insert(0, '.')
import numpy as np
from abacusutils_amd import _lib, synth
from abacusutils_amd.analysis.power_spectrum import calc_power
_lib.set_device(0)
box = 2000.0
pos = synth.synth_positions(200000, box, seed=5, clustered=True)
for nmesh in (550, 768):
    kw = dict(kbins=32, mubins=4, k_max=0.5, paste='TSC', nmesh=nmesh, compensated=True, interlaced=True, poles=[0, 2])
    for opt in (0, 1):
        _lib.set_option('pk_noxbin_inter', opt)
        calc_power(pos.copy(), box, **kw); calc_power(pos.copy(), box, **kw)
        _lib.profile_reset(); _lib.profile_enable(True)
        t = time.perf_counter()
        for _ in range(5): calc_power(pos.copy(), box, **kw)
        dt = (time.perf_counter() - t) / 5
        _lib.profile_enable(False)
        print(nmesh, 'unfused' if opt else 'fused', round(dt * 1e3, 2), {k: round(ms / 5, 3) for k, (ms, n) in _lib.profile_get().items() if ms / 5 > 0.02})
