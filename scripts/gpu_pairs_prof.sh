cd $GRAFT_REPO_ROOT
python - <<'PY'
import time, numpy as np
from abacusutils_amd.analysis.tpcf_corrfunc import _paircount
from abacusutils_amd import _lib
L=2000.0
for n in (100_000, 1_000_000):
    rng=np.random.default_rng(5)
    p=(rng.random((n,3),dtype=np.float32)*np.float32(L))
    x,y,z=[np.ascontiguousarray(p[:,i]) for i in range(3)]
    bins=np.geomspace(0.1,30.0,14).astype(np.float32)
    for rep in range(3): _paircount(0,x,y,z,L,bins)
    _lib.profile_reset(); _lib.profile_enable(True)
    t=time.perf_counter()
    for rep in range(10): _paircount(0,x,y,z,L,bins)
    dt=(time.perf_counter()-t)/10
    _lib.profile_enable(False)
    print(n, f'{dt*1e3:.2f} ms per call', {k: round(ms/c,4) for k,(ms,c) in _lib.profile_get().items()})
PY
