# one-off fuzz: more seeds of the calc_power / tsc_parallel / pair-count option sweeps than the test-suite runs
cd $GRAFT_REPO_ROOT
make -s -C oracle
timeout 3000 python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|Warning\|warnings.warn" | grep -v "tsc_parallel seed" | tail -15
import os, sys, time, traceback, warnings, torch
warnings.simplefilter('ignore')
sys.path.insert(0, 'tests')
import test_power_gpu as TP, test_tsc_gpu as TT, test_pairs_gpu as TC
S0, NS = int(os.environ.get('FUZZ_START', '100')), int(os.environ.get('FUZZ_COUNT', '150'))
for name, fn in (('calc_power', TP.test_random_option_sweep), ('tsc_parallel', TT.test_random_option_sweep),
                 ('pairs', TC.test_random_configuration_sweep)):
    bad = 0; t0 = time.time()
    for seed in range(S0, S0 + NS):
        try:
            fn(seed)
        except Exception as e:
            bad += 1
            print(name, 'seed', seed, 'FAILED', repr(e)[:300], flush=True)
    print(name, 'cases', NS, 'failing', bad, 'seconds', round(time.time() - t0, 1), flush=True)
PY
