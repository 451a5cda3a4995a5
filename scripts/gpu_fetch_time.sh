cd $GRAFT_REPO_ROOT
python - <<'PY'
import time, numpy as np, ctypes as C
from abacusutils_amd import synth, _lib
from abacusutils_amd.hod import GRAND_HOD as G
hd,pd,params=synth.synth_hod_inputs(10_000_000,10_000_000,seed=600)
p=G.marshal_params({'LRG':synth.LRG_PARAMS},params,False,True)
st=G.StagedCatalog(hd,pd)
st.populate(p); st.populate(p)
n=int(st.counts[0]+st.counts[3]); print('n',n)
def t(f,reps=20):
    f(); t0=time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter()-t0)/reps*1e3
print('fetch (2-D copy)      %.3f ms' % t(lambda: st.fetch('LRG')))
COLS=G.COLS
def old():
    cols={c:np.empty(n) for c in COLS}; ids=np.empty(n,np.int64)
    _lib.check(_lib.lib().abacus_hod_fetch(st._h,0,*[_lib.ptr(cols[c]) for c in COLS],_lib.ptr(ids)))
print('fetch (8 copies)      %.3f ms' % t(old))
blk=np.empty((8,n))
def reuse():
    _lib.check(_lib.lib().abacus_hod_fetch_block(st._h,0,_lib.ptr(blk),C.c_int64(n)))
print('2-D copy, reused host %.3f ms' % t(reuse))
print('np.empty+touch        %.3f ms' % t(lambda: np.empty((8,n)).fill(0)))
PY
python - <<'PY'
import time, numpy as np
from abacusutils_amd import synth, _lib
from abacusutils_amd.hod import GRAND_HOD as G
hd,pd,params=synth.synth_hod_inputs(10_000_000,10_000_000,seed=600)
p=G.marshal_params({'LRG':synth.LRG_PARAMS},params,False,True)
st=G.StagedCatalog(hd,pd)
for _ in range(3): st.populate(p)
for _ in range(20): st.populate_async(p)
st.wait_counts(); _lib.sync()
for _ in range(20): st.populate(p)
for k in range(6):
    t=time.perf_counter(); st.populate(p); t1=time.perf_counter(); st.fetch('LRG'); t2=time.perf_counter()
    print('populate %.3f ms fetch %.3f ms' % ((t1-t)*1e3,(t2-t1)*1e3))
PY
