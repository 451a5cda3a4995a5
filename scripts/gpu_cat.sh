cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python - <<'PY' 2>&1 | tail -60
import sys, json
sys.argv=['bench.py','--no-cpu']
import torch
import bench, bench_pk
args=bench.parse(); d=bench.Dist()
args.no_cpu=False
print(json.dumps(bench_pk.bench_catalog(args,d), indent=1))
PY
timeout 600 python -m pytest tests/test_catalog_gpu.py -m gpu -x -q 2>&1 | tail -5
