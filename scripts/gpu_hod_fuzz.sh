# one-off fuzz: many more seeds of tests/sweep.py than the test-suite runs, HIP vs oracle, bit-exact
cd $GRAFT_REPO_ROOT
make -s -C oracle
mkdir -p gpurun_out/fuzz
timeout 1100 python - <<'PY' 2> gpurun_out/fuzz/hod_fuzz.err
import sys, time, numpy as np
sys.path.insert(0, 'tests')
from sweep import sweep_case
from abacusutils_amd.hod import GRAND_HOD as G
from oracle import oracle
bad = 0; t0 = time.time(); ngal = 0
import os
S0, NS = int(os.environ.get("FUZZ_START", "100")), int(os.environ.get("FUZZ_COUNT", "400"))
for seed in range(S0, S0 + NS):
    nh, npart = (60000, 90000) if seed % 10 else (700000, 1100000)
    hd, pd, params, tracers, ranks, rsd = sweep_case(seed, nh, npart)
    st = G.StagedCatalog(hd, pd)
    # FUZZ_REPEAT > 1: further populates on the same staged catalogue with shifted cuts - the lazy keep masks and the
    # mass-sorted key index of the sparse mixes only come into play from the second populate on
    for rep in range(int(os.environ.get("FUZZ_REPEAT", "1"))):
        if rep:
            tracers = {k: dict(v, logM_cut=v['logM_cut'] + 0.07 * (rep if rep % 2 else -rep)) for k, v in tracers.items()}
        st.populate(G.marshal_params(tracers, params, ranks, rsd))
        kc, ks = st.fetch_keep()
        mock = {tr: st.fetch(tr) for tr in tracers}
        want, wkc, wks = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=32, enable_ranks=ranks, rsd=rsd, return_keep=True)
        ok = np.array_equal(kc, wkc) and np.array_equal(ks, wks)
        for tr in tracers:
            for c in ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass', 'id'):
                ok = ok and np.array_equal(mock[tr][c], want[tr][c])
            ngal += len(want[tr]['x'])
        if not ok:
            bad += 1
            print('MISMATCH seed', seed, 'populate', rep, list(tracers), ranks, rsd, int((kc != wkc).sum()), int((ks != wks).sum()), flush=True)
    st.free()
    if (seed - S0) % 50 == 49:
        print('progress', seed - S0 + 1, 'of', NS, 'mismatching', bad, 'galaxies', ngal, round(time.time() - t0, 1), 's', flush=True)
print("cases", NS, 'mismatching', bad, 'galaxies compared', ngal, 'seconds', round(time.time() - t0, 1))
PY
