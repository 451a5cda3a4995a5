cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_tsc_gpu.py tests/test_power_gpu.py -m gpu -x -q 2>&1 | tail -3
ABACUS_TSC_ACC32=1 timeout 900 python -m pytest tests/test_tsc_gpu.py tests/test_power_gpu.py -m gpu -x -q 2>&1 | tail -3
for NM in 1024 2048; do
for v in "" "ABACUS_TSC_ACC32=1"; do
env $v timeout 600 python bench.py --workload pk --nmesh $NM --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$NM $v tile_deposit', d['kernels_ms']['tsc_tile_deposit'], d['ms_per_step'], d['mean_P_over_shot_noise'])"
done
done
