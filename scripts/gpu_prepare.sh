#!/bin/bash
# prepare_sim on the device: its parity tests and the `prepare` leg of the bench (one pass through HBM against the column-by-column path)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prepare
make -s -C oracle
timeout -k 10 500 python -u -m pytest tests/test_prepare_gpu.py tests/test_staging_gpu.py -m gpu -x -q -p no:cacheprovider > gpurun_out/prepare/tests.log 2>&1
tail -8 gpurun_out/prepare/tests.log
timeout -k 10 300 python3 - > gpurun_out/prepare/bench.json 2> gpurun_out/prepare/bench.err <<'PY'
import json, sys, types
sys.path.insert(0, '.')
import bench_pk
from abacusutils_amd import _lib
from abacusutils_amd.comm import Dist
_lib.set_device(0)
args = types.SimpleNamespace(no_cpu=True)
print(json.dumps(bench_pk.bench_prepare(args, Dist(None))))
PY
cat gpurun_out/prepare/bench.json | cut -c1-1500; tail -3 gpurun_out/prepare/bench.err
timeout 120 python3 scripts/prof_prepare_calls.py > gpurun_out/prepare/calls.txt 2>&1; cat gpurun_out/prepare/calls.txt
