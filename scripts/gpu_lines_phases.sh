#!/bin/bash
# phase breakdown of the list build's split rounds (option tsc_lines_clk): shader-clock ticks of thread 0 of every workgroup per
# phase, for the coarse and the fine pass of one 1024^3 (NMESH) deposit of 1e8 particles
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
import ctypes as C, os, numpy as np
from abacusutils_amd import _lib
from abacusutils_amd.analysis import power_spectrum as ps
nmesh = int(os.environ.get('NMESH', '1024')); n = 100_000_000; L = 2000.0
pos = np.random.default_rng(300).random((n, 3), dtype=np.float32); pos *= np.float32(L)
d = _lib.DeviceArray(pos)
lib = _lib.lib()
grid = _lib.DeviceArray(np.zeros(1, dtype=np.float32)) if False else None
import time
kb, mb = ps.get_k_mu_edges(L, np.pi * nmesh / L + 1e-6, 512, 4, False)
ke = np.ascontiguousarray(kb, dtype=np.float64); me = np.ascontiguousarray(mb, dtype=np.float64)
poles = np.array([0, 2, 4], dtype=np.int64); outs = ps._alloc_outputs(len(ke) - 1, len(me) - 1, 3)
def step():
    _lib.check(lib.abacus_power_from_particles_dev(d.ptr, C.c_int64(n), None, None, C.c_int64(0), None, C.c_double(L), nmesh, 0, None, 0,
               _lib.ptr(ke), len(ke) - 1, _lib.ptr(me), len(me) - 1, _lib.ptr(poles), 3, *[_lib.ptr(o) for o in outs]))
step(); _lib.sync()
_lib.set_option('tsc_lines_clk', 1)
step(); _lib.sync()
out = (C.c_uint64 * 32)()
_lib.check(lib.abacus_tsc_lines_clocks(out))
step(); _lib.sync()
_lib.check(lib.abacus_tsc_lines_clocks(out))
v = np.array(list(out), dtype=np.float64)
names = ['count', 'barrier1', 'owner1', 'barrier2', 'owner2+carry', 'barrier3', 'place', 'barrier4', 'writeout', 'mid: next geometry']
for tag, off in (('coarse', 0), ('fine', 16)):
    t = v[off:off + 10]; tot = t.sum()
    print(tag, 'total ticks', tot, {nm: round(100 * x / tot, 1) for nm, x in zip(names, t) if tot})
PY
