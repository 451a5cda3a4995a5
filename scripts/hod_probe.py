"""A few populates of one HOD bench workload and nothing else: the process to put under `rocprofv3 --pmc ...`.
usage: python3 scripts/hod_probe.py [multi|c2] [populates]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from abacusutils_amd import _lib, synth  # noqa: E402
from abacusutils_amd.hod import GRAND_HOD as G  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'multi'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
multi = which == 'multi'
hd, pd, params = synth.synth_hod_inputs(10_000_000, 10_000_000, seed=600, with_ranks=multi)
tracers = synth.PRODUCTION_TRACERS if multi else {'LRG': synth.LRG_PARAMS}
p = G.marshal_params(tracers, params, multi, True)
st = G.StagedCatalog(hd, pd)
for _ in range(n):
    st.populate(p)
print(which, st.wait_counts(), st.candidates())
