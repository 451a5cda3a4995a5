cd $GRAFT_REPO_ROOT
for NM in 1024 2048; do
for v in 1024 512 256; do
ABACUS_TSC_COARSE=$v timeout 600 python bench.py --workload pk --nmesh $NM --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('$NM coarse<=$v', {a:round(b,2) for a,b in k.items() if 'ms_' in a}, 'sum', round(sum(b for a,b in k.items() if 'ms_' in a),2), d['mean_P_over_shot_noise'])"
done
done
