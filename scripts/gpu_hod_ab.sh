#!/bin/bash
# HOD tests, then the HOD bench legs (C2, hod_multi, hod_large) once per option set: gpu_hod_ab.sh [tests|notests] [name:opt=val,...] ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/hod_ab
mkdir -p "$O"
make -s -C oracle
what=${1:-tests}; shift
if [ "$what" = tests ]; then
  timeout 1000 python -m pytest tests/test_hod_gpu.py tests/test_abacus_hod_gpu.py tests/test_nfw_gpu.py tests/test_reseed_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee "$O/tests.log" || exit 1
fi
for spec in ${@:-base}; do
  mode=${spec%%:*}
  opt_=()
  if [ "$spec" != "$mode" ]; then
    IFS=, read -ra kv <<< "${spec#*:}"
    for o in "${kv[@]}"; do opt_+=(--option "$o"); done
  fi
  timeout 400 python bench.py --no-pk --no-cpu --steps 30 --warmup 3 "${opt_[@]}" > "$O/hod_$mode.json" 2> "$O/hod_$mode.err" || { tail -3 "$O/hod_$mode.err"; exit 1; }
  python - "$O/hod_$mode.json" "$mode" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ('', 'hod_multi', 'hod_large'):
    e = d if not k else d.get(k, {})
    if 'ms_per_step' in e:
        print(sys.argv[2], k or 'c2', round(e['ms_per_step'], 4), 'ms', '%.3g halos/s' % e['value'], {a: round(b * 1e3, 1) for a, b in e['kernels_ms'].items()})
PY
done
