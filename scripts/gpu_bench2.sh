cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 3 --warmup 1 --nhalo 1000000 --npart 1000000 --nmesh 256 --npk 2000000 --no-cpu --slab-timeout 90 > gpurun_out/bench2.json 2> gpurun_out/bench2.err
echo "rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step')}); print('pk', d['pk'].get('ms_per_step'), d['pk'].get('error')); print('slab', d.get('pk_slab'))
PY
tail -5 gpurun_out/bench2.err | cut -c1-300
