#!/bin/bash
# drop-in call latencies + the tests of the mock containers
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/latency
make -s -C oracle
timeout 900 python -X faulthandler -m pytest tests/test_abacus_hod_gpu.py tests/test_catalog_gpu.py -m gpu -x -q > gpurun_out/latency/tests.log 2>&1
tail -4 gpurun_out/latency/tests.log
grep -n "Fatal\|test_.*py\", line\|Segmentation\|Error" gpurun_out/latency/tests.log | head -20
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/latency/calls.json | cut -c1-1800
import argparse, json, sys
sys.path.insert(0, '.')
import bench, bench_pk
from abacusutils_amd import _lib
from abacusutils_amd.comm import Dist
_lib.set_device(0)
args = argparse.Namespace(nhalo=10_000_000, npart=10_000_000, no_cpu=True)
out = bench.bench_calls(args, Dist(None))
out['catalog'] = bench_pk.bench_catalog(args, Dist(None))
print(json.dumps(out))
PY
