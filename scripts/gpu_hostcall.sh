#!/bin/bash
# the drop-in calc_power call with NumPy positions (PCIe included): 1e8 float32 particles, 2048^3 and 1024^3, TSC, non-interlaced
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/hostcall
timeout 900 python - > gpurun_out/hostcall/hostcall.txt 2> gpurun_out/hostcall/hostcall.err <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from abacusutils_amd import _lib
from abacusutils_amd.analysis.power_spectrum import calc_power
_lib.set_device(0)
n, L = 100_000_000, 2000.0
rng = np.random.default_rng(300)
pos = rng.random((n, 3), dtype=np.float32) * np.float32(L * 0.999999)
for what in ('pageable', 'pinned'):
    host = pos if what == 'pageable' else _lib.pinned_empty(pos.shape, np.float32)
    if what == 'pinned':
        host[:] = pos
    d = _lib.DeviceArray(nbytes=pos.nbytes, dtype=np.float32, shape=pos.shape)
    L_ = _lib.lib()
    import ctypes as C
    for rep in range(3):
        t = time.perf_counter(); _lib.check(L_.abacus_memcpy_h2d(d.ptr, _lib.ptr(host), C.c_uint64(pos.nbytes))); up = time.perf_counter() - t
        t = time.perf_counter(); _lib.check(L_.abacus_memcpy_d2h(_lib.ptr(host), d.ptr, C.c_uint64(pos.nbytes))); down = time.perf_counter() - t
    print(f'{what}: upload {pos.nbytes / up / 1e9:.1f} GB/s ({up * 1e3:.0f} ms), download {pos.nbytes / down / 1e9:.1f} GB/s ({down * 1e3:.0f} ms)', flush=True)
    del d
for nmesh in (1024, 2048):
    kw = dict(kbins=512, mubins=4, k_max=np.pi * nmesh / L + 1e-6, paste='TSC', nmesh=nmesh, compensated=False,
              interlaced=False, poles=[0, 2, 4])
    import ctypes as C
    last = _lib.lib().abacus_power_last_batches; last.restype = C.c_double
    for opt in (0, 1):      # batched upload behind the deposits (the default where it pays) / one copy in front (pk_nobatch)
        _lib.set_option('pk_nobatch', opt)
        for rep in range(3):
            t = time.perf_counter(); tab = calc_power(pos, L, **kw); dt = time.perf_counter() - t
            print(f'calc_power from NumPy, nmesh {nmesh}, {int(last())} upload batch(es): {dt * 1e3:.1f} ms (P mean {float(np.mean(tab["power"])):.3f})', flush=True)
    _lib.set_option('pk_nobatch', 0)
    moved = pos.copy(); moved[::1000, 0] += np.float32(L)
    t = time.perf_counter(); calc_power(moved, L, **kw); dt = time.perf_counter() - t
    print(f'  with positions to wrap (copied back): {dt * 1e3:.1f} ms', flush=True)
PY
cat gpurun_out/hostcall/hostcall.txt; tail -5 gpurun_out/hostcall/hostcall.err
