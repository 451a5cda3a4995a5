#!/bin/bash
# 2048^3 P(k) bench once per value of the `dbg` option (ablation bits of fft_x_bin / spectrum_bin): gpu_pk_ablate.sh 0 4 8 ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/pk_ablate
mkdir -p "$O"
for o in "$@"; do
  timeout 300 python bench.py --workload pk --nmesh 2048 --steps 4 --warmup 1 --no-cpu --option dbg=$o > "$O/pk_dbg$o.json" 2> "$O/pk_dbg$o.err" || { tail -3 "$O/pk_dbg$o.err"; exit 1; }
  python - "$O/pk_dbg$o.json" "$o" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("dbg", sys.argv[2], round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["kernels_ms"].items() if v > 0.1})
PY
done
