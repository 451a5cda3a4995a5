cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ablate
for o in 0 1 2 3; do
  timeout -k 10 200 python3 bench.py --workload pk --nmesh 2048 --steps 3 --warmup 1 --no-cpu --option dbg_tsc=$o > gpurun_out/ablate/tsc_$o.json 2>/dev/null
  python3 - $o <<'PY'
import json,sys
o=sys.argv[1]
d=json.load(open(f'gpurun_out/ablate/tsc_{o}.json'))
k=d['kernels_ms']
print('dbg_tsc',o,'step',round(d['ms_per_step'],2),'deposit',k.get('tsc_tile_deposit'),'fine',k.get('tsc_lines_fine'))
PY
done
