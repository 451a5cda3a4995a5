#!/bin/bash
# second-generation TSC lists: parity tests, then the 1024^3 / 2048^3 P(k) legs with per-kernel times (A/B against the
# first generation with --option tsc_oldlists=1)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/lines
mkdir -p "$O"
make -s -C oracle
timeout -k 10 900 python -m pytest tests/test_tsc_gpu.py -m gpu -x -q -k "${TSC_K:-line_lists}" 2>&1 | tail -15 | tee "$O/tests.log"
grep -q "passed" "$O/tests.log" || exit 1
grep -q "failed\|error" "$O/tests.log" && exit 1
for NM in ${MESHES:-1024}; do
  for spec in new old:tsc_oldlists=1; do
    mode=${spec%%:*}; opt_=()
    [ "$spec" != "$mode" ] && opt_=(--option "${spec#*:}")
    timeout -k 10 300 python bench.py --workload pk --nmesh $NM --steps 4 --warmup 1 --no-cpu "${opt_[@]}" > "$O/pk${NM}_$mode.json" 2> "$O/pk${NM}_$mode.err" || { tail -5 "$O/pk${NM}_$mode.err"; exit 1; }
    python - "$O/pk${NM}_$mode.json" "$NM $mode" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"], 2), {k: round(v, 3) for k, v in d["kernels_ms"].items() if v > 0.02}, 'interlaced', round(d.get('interlaced_compensated', {}).get('ms_per_step', 0), 2), 'P/shot', round(d['mean_P_over_shot_noise'], 5))
PY
  done
done
