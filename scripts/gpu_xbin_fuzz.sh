#!/bin/bash
# one-off fuzz of the fused last pass (cached geometry descriptor) on random bin edges: FUZZ_COUNT further seeds of
# tests/test_power_gpu.py::test_fused_last_pass_random_edges, counting which generation served them
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fuzz
timeout 1000 python - <<'PY' 2> gpurun_out/fuzz/xbin_fuzz.err | tee gpurun_out/fuzz/xbin_fuzz.log
import os, sys, time, warnings
warnings.simplefilter('ignore')
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_power_gpu as TP
from abacusutils_amd import _lib


class Opt:
    def set(self, k, v):
        _lib.set_option(k, v)


S0, NS = int(os.environ.get('FUZZ_START', '100')), int(os.environ.get('FUZZ_COUNT', '300'))
gens = {1: 0, 2: 0}
orig = _lib.lib().abacus_power_xbin_generation
bad = 0
t0 = time.time()
for seed in range(S0, S0 + NS):
    try:
        TP.test_fused_last_pass_random_edges(Opt(), seed)
        gens[orig()] = gens.get(orig(), 0) + 1
    except Exception as e:
        bad += 1
        print('seed', seed, 'FAILED', repr(e)[:300], flush=True)
    if (seed - S0 + 1) % 50 == 0:
        print('progress', seed - S0 + 1, 'failing', bad, 'last generation counts', gens, round(time.time() - t0, 1), 's', flush=True)
print('cases', NS, 'x 5 edge sets (auto, auto, interlaced pair, cross, interlaced cross), failing', bad, 'generation of the last set per seed', gens)
PY
