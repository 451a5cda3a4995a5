#!/bin/bash
# mixed-radix native transform (csrc/gfft.hip) against hipFFT, per mesh size: FFT kernel times of one P(k) step
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/gfft
mkdir -p "$O"
for NM in ${MESHES:-96 384 550 768 1536}; do
  for spec in gen hip:fft_hipfft=1; do
    mode=${spec%%:*}; opt_=()
    [ "$spec" != "$mode" ] && opt_=(--option "${spec#*:}")
    NP=$(( NM > 600 ? 100000000 : 20000000 ))
    timeout -k 10 300 python bench.py --workload pk --nmesh $NM --npk $NP --steps 6 --warmup 2 --no-cpu "${opt_[@]}" > "$O/pk${NM}_$mode.json" 2> "$O/pk${NM}_$mode.err" || { tail -3 "$O/pk${NM}_$mode.err"; exit 1; }
    python - "$O/pk${NM}_$mode.json" "$NM $mode" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
f = {k: round(v, 3) for k, v in d["kernels_ms"].items() if 'fft' in k}
print(sys.argv[2], 'step', round(d["ms_per_step"], 3), 'fft total', round(sum(f.values()), 3), f)
PY
  done
done
