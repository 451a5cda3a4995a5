# one-off fuzz at sizes where set_superblocks rounds the superblock count to whole workgroups per CU (> 4.2e6 objects per kind):
# seeds of tests/sweep.py at 4.5e6 ... 9e6 objects, two populates each, HIP vs oracle, bit-exact
cd $GRAFT_REPO_ROOT
make -s -C oracle
mkdir -p gpurun_out/fuzz
timeout 1000 python - <<'PY' 2> gpurun_out/fuzz/hod_fuzz_large.err | tee gpurun_out/fuzz/hod_fuzz_large.txt
import sys, time, os, numpy as np
sys.path.insert(0, 'tests')
from sweep import sweep_case
from abacusutils_amd.hod import GRAND_HOD as G
from oracle import oracle
bad = 0; t0 = time.time(); ngal = 0
S0, NS = int(os.environ.get("FUZZ_START", "500")), int(os.environ.get("FUZZ_COUNT", "10"))
for seed in range(S0, S0 + NS):
    nh, npart = 4_500_000 + 450_000 * (seed % 7), 9_000_000 - 600_000 * (seed % 5)
    hd, pd, params, tracers, ranks, rsd = sweep_case(seed, nh, npart)
    st = G.StagedCatalog(hd, pd)
    for rep in range(2):
        if rep:
            tracers = {k: dict(v, logM_cut=v['logM_cut'] + 0.07) for k, v in tracers.items()}
        st.populate(G.marshal_params(tracers, params, ranks, rsd))
        kc, ks = st.fetch_keep()
        mock = {tr: st.fetch(tr) for tr in tracers}
        want, wkc, wks = oracle.gen_gal_cat(hd, pd, tracers, params, Nthread=oracle.max_threads(), enable_ranks=ranks, rsd=rsd, return_keep=True)
        ok = np.array_equal(kc, wkc) and np.array_equal(ks, wks)
        for tr in tracers:
            for c in ('x', 'y', 'z', 'vx', 'vy', 'vz', 'mass', 'id'):
                ok = ok and np.array_equal(mock[tr][c], want[tr][c])
            ngal += len(want[tr]['x'])
        if not ok:
            bad += 1
            print('MISMATCH seed', seed, 'rep', rep, sorted(tracers), flush=True)
    st.free()
    print(f'seed {seed}: {nh} + {npart} objects, tracers {sorted(tracers)}, {time.time() - t0:.0f} s', flush=True)
print(f'{NS} seeds x 2 populates, {ngal} galaxies compared, {bad} mismatches')
PY
