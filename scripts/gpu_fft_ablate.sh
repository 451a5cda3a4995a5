cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for DBG in 0 8; do
ABACUS_DBG_FFT=$DBG timeout 600 python bench.py --workload pk --nmesh 2048 --steps 2 --warmup 1 --no-cpu 2>gpurun_out/abl.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg $DBG', {k:round(v,2) for k,v in d['kernels_ms'].items() if k.startswith('fft')})"
done
