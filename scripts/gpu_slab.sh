#!/bin/bash
# slab (multi-GPU) P(k) path on the one GPU of the box: parity tests, then ONE rank of the slab estimator at 2048^3 timed
# against the single-GPU path (the 8-GPU strong-scaling curve starts from this number)
# usage: gpu_slab.sh [tests|notests]
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/slab
mkdir -p "$O"
make -s -C oracle
if [ "${1:-tests}" = tests ]; then
  timeout 1500 python -m pytest tests/test_slab_power.py tests/test_comm_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee "$O/tests.log" || exit 1
fi
timeout 600 python - > "$O/slab_world1.json" 2> "$O/slab_world1.err" <<'PY' || { tail -5 "$O/slab_world1.err"; exit 1; }
import argparse, json, sys
sys.path.insert(0, '.')
from abacusutils_amd import _lib
from abacusutils_amd.comm import Dist
import bench_pk
_lib.set_device(0)
args = argparse.Namespace(nmesh=2048, npk=100_000_000, steps=4, warmup=1, no_cpu=True)
out = {}
for name, opt in (('fused', 0), ('plain', 1)):
    _lib.set_option('slab_nofuse', opt)
    _lib.profile_reset(); _lib.profile_enable(True)
    r = bench_pk.bench_pk_slab(args, Dist(None))
    _lib.profile_enable(False)
    r['kernels_ms'] = {k: round(ms / n, 3) for k, (ms, n) in _lib.profile_get().items() if n and ms / n > 0.05}
    out[name] = r
print(json.dumps(out))
PY
python - "$O/slab_world1.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, round(v['value'], 2), 'ms', v['kernels_ms'])
PY
