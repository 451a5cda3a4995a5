#!/bin/bash
# the `pk_slab` leg of bench.py as a rank process runs it, on the one GPU: without a transport and with a ONE-rank RCCL communicator
# (every collective of the leg executes, incl. the cross-power measurement and its provisional BENCH-LEG line)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/slab_leg; mkdir -p $O
timeout -k 10 400 python3 - > $O/leg.out 2> $O/leg.err <<'PY'
import argparse, json, sys
sys.path.insert(0, '.')
from abacusutils_amd import _lib
from abacusutils_amd.comm import Dist, RcclComm
import bench_pk
_lib.set_device(0)
args = argparse.Namespace(nmesh=1024, npk=40_000_000, steps=3, warmup=1, no_cpu=True, slab_presorted=False)
a = bench_pk.bench_pk_slab(args, Dist(None))
print('LOCAL', json.dumps({k: a[k] for k in ('value', 'cross', 'mean_P_over_shot_noise')}), flush=True)
d = Dist(RcclComm(0, 1, key='slableg'))
b = bench_pk.bench_pk_slab(args, d)
print('RCCL1', json.dumps({k: b.get(k) for k in ('value', 'cross', 'mean_P_over_shot_noise', 'all_to_all', 'routing')}), flush=True)
d.finish()
PY
cat $O/leg.out | cut -c1-700; tail -3 $O/leg.err
