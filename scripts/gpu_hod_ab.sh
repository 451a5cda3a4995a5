#!/bin/bash
# HOD A/B on the GPU box: tests of the HOD path, then the bench legs with the interval classifier of hod_exact on and off
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/hod_ab
mkdir -p "$O"
make -s -C oracle
if [ "$1" != "notest" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "hod" 2>&1 | tail -8 | tee "$O/tests.log"
  grep -qE "[0-9]+ (failed|error)" "$O/tests.log" && exit 1
fi
for mode in keys nokeys; do
  case $mode in
    keys) opt_=() ;;
    nokeys) opt_=(--option hod_nokeys=1) ;;
  esac
  timeout 600 python bench.py --no-cpu --no-pk --steps 20 --warmup 3 "${opt_[@]}" > "$O/bench_$mode.json" 2> "$O/bench_$mode.err" || { tail -5 "$O/bench_$mode.err"; exit 1; }
  python - "$O/bench_$mode.json" "$mode" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ('', 'hod_multi', 'hod_large'):
    e = d[k] if k else d
    if 'error' in e:
        print(sys.argv[2], k, e['error']); continue
    print(sys.argv[2], k or 'C2', 'ms/step %.4f' % e['ms_per_step'], 'halos/s %.3e' % e['value'], {a: round(b * 1e3, 1) for a, b in e['kernels_ms'].items() if a.startswith('hod_') and a not in ('hod_build_recs', 'hod_shadow', 'hod_minmax')})
PY
done
