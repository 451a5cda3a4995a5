cd $GRAFT_REPO_ROOT
for v in 0 4 8 1 2; do
ABACUS_DBG=$v timeout 600 python bench.py --workload pk --nmesh 2048 --npk 20000000 --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('DBG=$v spectrum_bin', d['kernels_ms']['spectrum_bin'])"
done
