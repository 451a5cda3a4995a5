import cProfile, pstats, sys, io, time
sys.path.insert(0, '.')
import numpy as np
from abacusutils_amd import _lib
from abacusutils_amd.hod import prepare_sim as prep
from abacusutils_amd.synth import synth_compaso_slabs
_lib.set_device(0)
slabs, header = synth_compaso_slabs(numslabs=1, n_halo=1_000_000, seed=900, lbox=2000.0, subsample_frac=0.006)
halos, parts = slabs[0]['halos'], slabs[0]['parts']
Mpart, h = header['ParticleMassHMsun'], header['H0'] / 100.0
kw = dict(MT=True, want_ranks=True, want_AB=True, Lbox=header['BoxSize'])
prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=7, **kw)
pr = cProfile.Profile(); pr.enable()
prep.prepare_slab_arrays(halos, parts, Mpart, h, rng=8, **kw)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35); print(s.getvalue()[:6000])
