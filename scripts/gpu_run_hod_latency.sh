#!/bin/bash
# host-side cost of the drop-in calls: AbacusHOD.run_hod() in an MCMC-style loop (C2: 1e7 halos + 1e7 particles, LRG),
# with a cProfile breakdown
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/latency
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/latency/run_hod.log | tail -60
import time, numpy as np
from abacusutils_amd import synth, _lib
from abacusutils_amd.hod.abacus_hod import AbacusHOD
HOD = dict(tracer_flags={'LRG': True, 'ELG': False, 'QSO': False}, want_ranks=False, want_AB=True, want_shear=False,
           want_rsd=True, LRG_params=synth.LRG_PARAMS, ELG_params=synth.ELG_PARAMS, QSO_params=synth.QSO_PARAMS)
hd, pd, params = synth.synth_hod_inputs(10_000_000, 10_000_000, seed=600)
ball = AbacusHOD.from_arrays(hd, pd, params, HOD)
m = ball.run_hod(ball.tracers, True, Nthread=16)
print('galaxies', len(m['LRG']['x']))
for lazy in (False, True):
    if hasattr(ball, 'lazy_columns'):
        ball.lazy_columns = lazy
    elif lazy:
        break
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(100):
            ball.tracers['LRG']['logM_cut'] = 13.3 + 0.001 * (i % 5)
            m = ball.run_hod(ball.tracers, True, Nthread=16)
        dt = (time.perf_counter() - t0) / 100
        print(f'run_hod() MCMC loop, lazy_columns={lazy}: {dt*1e3:.3f} ms per call')
if hasattr(ball, 'lazy_columns'):
    ball.lazy_columns = False
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(100): m = ball.run_hod(ball.tracers, True, Nthread=16)
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
PY
