# one-off fuzz: more seeds of the pair-count / tsc_parallel / calc_power option sweeps than the test-suite runs
# (FUZZ_START, FUZZ_COUNT seeds of each; FUZZ_SECONDS caps the whole run: the sweeps that did not finish say so)
cd $GRAFT_REPO_ROOT
make -s -C oracle
mkdir -p gpurun_out/fuzz
timeout 1100 python - <<'PY' 2> gpurun_out/fuzz/pk_fuzz.err
import os, sys, time, warnings
warnings.simplefilter('ignore')
sys.path.insert(0, 'tests')
import test_power_gpu as TP, test_tsc_gpu as TT, test_pairs_gpu as TC
S0, NS = int(os.environ.get('FUZZ_START', '100')), int(os.environ.get('FUZZ_COUNT', '150'))
T_END = time.time() + float(os.environ.get('FUZZ_SECONDS', '1000'))
for name, fn in (('pairs', TC.test_random_configuration_sweep), ('tsc_parallel', TT.test_random_option_sweep),
                 ('calc_power', TP.test_random_option_sweep)):
    bad = 0; done = 0; t0 = time.time()
    for seed in range(S0, S0 + NS):
        if time.time() > T_END:
            break
        try:
            fn(seed)
        except Exception as e:
            bad += 1
            print(name, 'seed', seed, 'FAILED', repr(e)[:300], flush=True)
        done += 1
        if done % 25 == 0:
            print(name, 'progress', done, 'failing', bad, round(time.time() - t0, 1), 's', flush=True)
    print(name, 'cases', done, 'of', NS, 'failing', bad, 'seconds', round(time.time() - t0, 1), flush=True)
PY
