#!/bin/bash
# P(k) bench once per mode: gpu_pk_quick.sh mode ...   mode = name or name:option=value[,option=value...]; NMESH (1024)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/pk_quick
mkdir -p "$O"
for spec in "$@"; do
  mode=${spec%%:*}
  opt_=()
  if [ "$spec" != "$mode" ]; then
    IFS=, read -ra kv <<< "${spec#*:}"
    for o in "${kv[@]}"; do opt_+=(--option "$o"); done
  fi
  timeout -k 10 300 python bench.py --workload pk --nmesh ${NMESH:-1024} --steps 4 --warmup 1 --no-cpu "${opt_[@]}" > "$O/pk_$mode.json" 2> "$O/pk_$mode.err" || { tail -3 "$O/pk_$mode.err"; exit 1; }
  python - "$O/pk_$mode.json" "$mode" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"], 2), {k.replace('tsc_lines_', 'L.'): round(v, 3) for k, v in d["kernels_ms"].items() if v > 0.02 and 'fft' not in k}, 'il', round(d.get('interlaced_compensated', {}).get('ms_per_step', 0), 2))
PY
done
