cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -s -C oracle
timeout 900 python -m pytest tests/test_hod_gpu.py tests/test_abacus_hod_gpu.py tests/test_hod_shard.py -m gpu -x -q 2>&1 | tail -15
for v in ""; do
  echo "== variant: $v"
  env $v timeout 600 python bench.py --no-pk --no-cpu --steps 50 --warmup 3 > gpurun_out/bench_hod.json 2> gpurun_out/bench_hod.err
  python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_hod.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_host_sync','kernels_ms')}); print(d.get('roofline'))
PY
  tail -2 gpurun_out/bench_hod.err
done
