cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "0 0" "1 0" "2 0" "3 0" "0 1" "0 2" "0 3"; do
set -- $v
ABACUS_DBG=$1 ABACUS_DBG_TSC=$2 timeout 600 python bench.py --workload pk --nmesh 2048 --npk 100000000 --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('DBG=$1 TSC=$2', {k:d['kernels_ms'][k] for k in ('spectrum_bin','tsc_tile_deposit')})"
done
