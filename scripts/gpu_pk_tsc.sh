cd $GRAFT_REPO_ROOT
make -s -C oracle
timeout 1500 python -m pytest tests/test_tsc_gpu.py tests/test_power_gpu.py tests/test_slab_power.py -m gpu -x -q 2>&1 | tail -4
for NM in 1024 2048; do
timeout 900 python bench.py --workload pk --nmesh $NM --steps 3 --warmup 1 --no-cpu 2>gpurun_out/pk$NM.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('nmesh $NM', round(d['ms_per_step'],2), {k:v for k,v in d['kernels_ms'].items() if not k.startswith('scan')}, d.get('interlaced_compensated'))"
done
