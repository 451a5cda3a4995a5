cd $GRAFT_REPO_ROOT
make -s -C oracle
timeout 900 python -m pytest tests/test_hod_gpu.py tests/test_abacus_hod_gpu.py tests/test_hod_shard.py tests/test_nfw_gpu.py -m gpu -x -q 2>&1 | tail -2
for v in "" "ABACUS_HOD_NOREC=1"; do
  env $v timeout 600 python bench.py --no-pk --no-cpu --steps 50 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['ms_per_step_host_sync'], d['kernels_ms'])"
done
