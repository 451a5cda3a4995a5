#!/bin/bash
# pair counting on the GPU box: parity tests, then the bench leg (10^7 points, DD(r) to 30 Mpc/h) per option set
# usage: gpu_pairs.sh [tests|notests] [name=value ...]   (each name=value is one extra timed run with that option)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/pairs
mkdir -p "$O"
make -s -C oracle
what=${1:-tests}; shift
if [ "$what" = tests ]; then
  timeout 1200 python -m pytest tests/test_pairs_gpu.py tests/test_slab_pairs.py -m gpu -x -q 2>&1 | tail -8 | tee "$O/tests.log" || exit 1
fi
for opt in base "$@"; do
  timeout 600 python - "$opt" > "$O/pairs_$opt.json" 2> "$O/pairs_$opt.err" <<'PY' || { tail -5 "$O/pairs_$opt.err"; exit 1; }
import argparse, json, sys
sys.path.insert(0, '.')
from abacusutils_amd import _lib
from abacusutils_amd.comm import Dist
import bench_pk
_lib.set_device(0)
opt = sys.argv[1]
if opt != 'base':
    k, _, v = opt.partition('=')
    _lib.set_option(k, int(v or 1))
_lib.profile_reset(); _lib.profile_enable(True)
r = bench_pk.bench_pairs(argparse.Namespace(no_cpu=True), Dist(None))
_lib.profile_enable(False)
print(json.dumps({k: r[k] for k in ('ms_per_call', 'kernels_ms', 'pairs_counted', 'candidates_evaluated', 'roofline') if k in r}))
PY
  echo "$opt: $(cat "$O/pairs_$opt.json")"
done
