cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle
timeout 1200 python -m pytest tests/test_tsc_gpu.py tests/test_power_gpu.py -m gpu -x -q 2>&1 | tail -8
for nm in 1024 2048; do
timeout 900 python bench.py --workload pk --nmesh $nm --steps 4 --warmup 1 --no-cpu > gpurun_out/bench_pk$nm.json 2> gpurun_out/bench_pk$nm.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_pk$nm.json'))
print($nm, {k:d[k] for k in ('ms_per_step','kernels_ms','interlaced_compensated')}); print(d['roofline'])
PY
tail -2 gpurun_out/bench_pk$nm.err
done
