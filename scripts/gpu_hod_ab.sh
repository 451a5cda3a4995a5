#!/bin/bash
# HOD A/B on the GPU box: (tests of the HOD path, then) the bench legs once per mode.
# usage: gpu_hod_ab.sh [notest] [mode ...]    mode = name or name:option=value[,option=value...] (abacus_set_option names)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/hod_ab
mkdir -p "$O"
make -s -C oracle
if [ "$1" == "notest" ]; then
  shift
else
  timeout 1500 python -m pytest tests -m gpu -x -q -k "hod" 2>&1 | tail -8 | tee "$O/tests.log"
  grep -qE "[0-9]+ (failed|error)" "$O/tests.log" && exit 1
fi
[ $# -eq 0 ] && set -- base nokeys:hod_nokeys=1
for spec in "$@"; do
  mode=${spec%%:*}
  opt_=()
  if [ "$spec" != "$mode" ]; then
    IFS=, read -ra kv <<< "${spec#*:}"
    for o in "${kv[@]}"; do opt_+=(--option "$o"); done
  fi
  timeout 600 python bench.py --no-cpu --no-pk --steps 20 --warmup 3 "${opt_[@]}" > "$O/bench_$mode.json" 2> "$O/bench_$mode.err" || { tail -5 "$O/bench_$mode.err"; exit 1; }
  python - "$O/bench_$mode.json" "$mode" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ('', 'hod_multi', 'hod_large'):
    e = d[k] if k else d
    if 'error' in e:
        print(sys.argv[2], k, e['error']); continue
    n = e.get('launches_per_step', {})
    print(sys.argv[2], k or 'C2', 'ms/step %.4f' % e['ms_per_step'], 'halos/s %.3e' % e['value'],
          {a: (round(b * 1e3, 1), n.get(a)) for a, b in e['kernels_ms'].items() if a in ('hod_filter', 'hod_exact', 'hod_emit')},
          'cand', e.get('filter_candidates'), 'gal', e.get('galaxies'))
PY
done
