cd $GRAFT_REPO_ROOT
python - <<'PY'
import ctypes as C, numpy as np, time
from abacusutils_amd import _lib, synth
from abacusutils_amd.hod import GRAND_HOD as G
hd, pd, params = synth.synth_hod_inputs(10_000_000, 10_000_000, seed=600)
st = G.StagedCatalog(hd, pd)
p = G.marshal_params({'LRG': synth.LRG_PARAMS}, params, False, True)
print(st.populate(p))
out = (C.c_uint * 2)()
_lib.check(_lib.lib().abacus_hod_debug_queue(st._h, out))
print('survivors cent, sat:', out[0], out[1])
PY
