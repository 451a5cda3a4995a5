#!/bin/bash
# catalogue-side kernels (rvint / PID / pack9 unpacking, Menv): parity tests, then the `catalog` leg of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/catalog
mkdir -p "$O"
make -s -C oracle
timeout 900 python -m pytest tests/test_catalog_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee "$O/tests.log" || exit 1
timeout 600 python - > "$O/catalog.json" 2> "$O/catalog.err" <<'PY' || { tail -5 "$O/catalog.err"; exit 1; }
import argparse, json, sys
sys.path.insert(0, '.')
from abacusutils_amd import _lib
from abacusutils_amd.comm import Dist
import bench_pk
_lib.set_device(0)
print(json.dumps(bench_pk.bench_catalog(argparse.Namespace(no_cpu=True), Dist(None))))
PY
cat "$O/catalog.json"
