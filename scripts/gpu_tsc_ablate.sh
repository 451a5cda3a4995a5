cd $GRAFT_REPO_ROOT
for NM in 1024 2048; do
for v in 0 8; do
ABACUS_DBG_TSC=$v timeout 600 python bench.py --workload pk --nmesh $NM --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$NM DBG_TSC=$v tile_deposit', d['kernels_ms']['tsc_tile_deposit'], round(d['ms_per_step'],2))"
done
done
