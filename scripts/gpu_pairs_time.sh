cd $GRAFT_REPO_ROOT
make -s -C oracle
timeout 900 python -m pytest tests/test_pairs_gpu.py tests/test_slab_pairs.py tests/test_abacus_hod_gpu.py -m gpu -x -q 2>&1 | tail -4
python - <<'PY'
import time, numpy as np
from abacusutils_amd.analysis.tpcf_corrfunc import _paircount
L=2000.0
for n in (1_000_000, 10_000_000):
    rng=np.random.default_rng(5)
    p=(rng.random((n,3),dtype=np.float32)*np.float32(L))
    for shift in (0.0, -L/2):
        q = p + np.float32(shift)
        x,y,z=[np.ascontiguousarray(q[:,i]) for i in range(3)]
        bins=np.geomspace(0.1,30.0,14).astype(np.float32)
        for mode,kw in ((0,{}),(1,dict(pimax=30.0,npibins=30)),(2,dict(mu_max=1.0,nmubins=20))):
            _paircount(mode,x,y,z,L,bins,**kw)
            t=time.perf_counter(); c=_paircount(mode,x,y,z,L,bins,**kw); dt=time.perf_counter()-t
            cand = n*(n/L**3)*27*30.0**3
            print(f'n={n:.0e} shift={shift} mode={mode} {dt*1e3:.1f} ms  pairs {int(c.sum()):.3e}  candidates {cand:.2e} -> {cand/dt:.2e}/s')
PY
bash scripts/gpu_pairs_prof.sh
