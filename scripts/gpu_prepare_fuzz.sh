#!/bin/bash
# one-off fuzz of prepare_slab_arrays(rng=<seed>): the one-pass device path (abacus_prepare_slab) against the column-by-column path
# (option prep_columnwise) on seeded random slabs and switches - every column of both tables and the mask must be EQUAL
# usage: gpu_prepare_fuzz.sh [first_seed] [count]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fuzz
python3 - "${1:-0}" "${2:-60}" <<'PY' 2>&1 | tee gpurun_out/fuzz/prepare_fuzz.txt
import sys, warnings
warnings.simplefilter('ignore')
sys.path.insert(0, '.')
import numpy as np
from abacusutils_amd import _lib, synth
from abacusutils_amd.hod import prepare_sim as ps
_lib.set_device(0)
s0, cnt = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(770000 + seed)
    nh = int(rng.integers(2000, 120000))
    slabs, header = synth.synth_compaso_slabs(numslabs=1, n_halo=nh, seed=1000 + seed, lbox=float(rng.choice([300.0, 700.0, 2000.0])),
                                              subsample_frac=float(rng.choice([0.003, 0.01, 0.03])))
    halos, parts = slabs[0]['halos'], slabs[0]['parts']
    Mpart, h = header['ParticleMassHMsun'], header['H0'] / 100.0
    kw = dict(want_ranks=bool(rng.integers(2)), want_AB=bool(rng.integers(2)), Lbox=header['BoxSizeHMpc'], rng=int(rng.integers(1, 2**62)),
              part_index0=int(rng.integers(0, 2**40)), halo_index0=int(rng.integers(0, 2**36)))
    if rng.random() < 0.3:
        kw['shearmark'] = np.random.default_rng(seed).random((8, 8, 8))
    MT = bool(rng.integers(2))
    dev = bool(rng.integers(2))
    _lib.set_option('prep_columnwise', 1)
    H0, P0, m0 = ps.prepare_slab_arrays(halos, parts, Mpart, h, MT, **kw)
    _lib.set_option('prep_columnwise', 0)
    if dev:
        hh = {k: _lib.DeviceArray(v) for k, v in halos.items()}
        pp = {k: _lib.DeviceArray(v) for k, v in parts.items()}
    else:
        hh, pp = halos, parts
    H1, P1, m1 = ps.prepare_slab_arrays(hh, pp, Mpart, h, MT, **kw)
    ok = np.array_equal(m0, m1) and list(H0) == list(H1) and list(P0) == list(P1)
    for a, b in ((H0, H1), (P0, P1)):
        for k in a:
            ok = ok and a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True)
    if not ok:
        bad += 1
        print('seed', seed, 'MISMATCH', nh, MT, {k: v for k, v in kw.items() if k != 'shearmark'}, flush=True)
    if dev:
        for a in list(hh.values()) + list(pp.values()):
            a.free()
print('prepare fuzz:', cnt - bad, '/', cnt, 'equal (halo table, particle table, mask; host and device columns)')
PY
