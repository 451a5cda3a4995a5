cd "$GRAFT_REPO_ROOT" && make -s -C oracle && mkdir -p gpurun_out/prep && timeout 500 python - > gpurun_out/prep/prepare.json 2> gpurun_out/prep/prepare.err <<'PY'
import argparse, json, sys
sys.path.insert(0, '.')
from abacusutils_amd import _lib
from abacusutils_amd.comm import Dist
import bench_pk
_lib.set_device(0)
print(json.dumps(bench_pk.bench_prepare(argparse.Namespace(no_cpu=False), Dist(None))))
PY
cat gpurun_out/prep/prepare.json; tail -3 gpurun_out/prep/prepare.err
