cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_tsc_gpu.py tests/test_power_gpu.py -m gpu -x -q 2>&1 | tail -3
for NM in 1024 2048; do
for v in 0 3; do
ABACUS_DBG_TSC=$v timeout 600 python bench.py --workload pk --nmesh $NM --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$NM DBG_TSC=$v tile_deposit', d['kernels_ms']['tsc_tile_deposit'], round(d['ms_per_step'],2))"
done
done
