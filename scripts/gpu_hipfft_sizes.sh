cd $GRAFT_REPO_ROOT
timeout 1500 python - <<'PY' 2>&1 | grep -v "amdgpu.ids" | tail -30
import os, numpy as np, torch
from abacusutils_amd.analysis.power_spectrum import calc_power
n, box = 20_000_000, 2000.0
rng = np.random.default_rng(5)
pos = rng.random((n, 3), dtype=np.float32) * np.float32(box)
shot = box**3 / n
for nmesh, hip in ((1920, 0), (2000, 0), (2016, 0), (2100, 0), (2187, 0), (2304, 0), (2560, 0)):
    if hip: os.environ['ABACUS_FFT_HIPFFT'] = '1'
    else: os.environ.pop('ABACUS_FFT_HIPFFT', None)
    try:
        t = calc_power(pos, box, kbins=64, mubins=1, k_max=np.pi*nmesh/box, paste='TSC', nmesh=nmesh, compensated=True, interlaced=True, poles=[0])
        p0 = np.asarray(t['poles'])[:, 0]
        print(nmesh, 'hipfft' if hip else 'default', 'P0/shot mid', float(np.mean(p0[8:56]) / shot), 'min/max', float(p0[8:56].min()/shot), float(p0[8:56].max()/shot), flush=True)
    except Exception as e:
        print(nmesh, 'ERROR', repr(e)[:300], flush=True)
PY
