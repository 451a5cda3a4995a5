cd $GRAFT_REPO_ROOT
for v in 0 1 2 3; do
ABACUS_DBG_FFT=$v timeout 600 python bench.py --workload pk --nmesh 2048 --npk 20000000 --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('DBG_FFT=$v', {k:d['kernels_ms'][k] for k in ('fft_z_r2c','fft_cols_y','fft_cols_x')})"
done
