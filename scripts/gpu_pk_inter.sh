cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -20
import numpy as np, torch, time
from abacusutils_amd import _lib
from abacusutils_amd.analysis.power_spectrum import calc_power
import ctypes as C
n, box, nmesh = 100_000_000, 2000.0, 2048
rng = np.random.default_rng(300)
pos = rng.random((n, 3), dtype=np.float32) * np.float32(box)
kw = dict(kbins=512, mubins=4, k_max=np.pi*nmesh/box+1e-6, paste='TSC', nmesh=nmesh, poles=[0,2,4], compensated=True, interlaced=True)
calc_power(pos, box, **kw)
_lib.profile_reset(); _lib.profile_enable(True)
t0=time.perf_counter()
for _ in range(2): calc_power(pos, box, **kw)
dt=(time.perf_counter()-t0)/2
_lib.profile_enable(False)
k={a:(ms/c, c) for a,(ms,c) in _lib.profile_get().items() if c}
for a,(ms,c) in sorted(k.items(), key=lambda x:-x[1][0]*x[1][1]): print(f'{a:28s} {ms:8.3f} ms x {c/2:.0f} per call')
print('host-array call', dt*1e3, 'ms; kernel sum per call', sum(ms*c for ms,c in k.values())/2)
PY
