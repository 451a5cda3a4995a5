#!/bin/bash
# kernel + copy timeline of one P(k) step (gaps between device activities): gpu_timeline.sh [nmesh] [npk]
cd "$GRAFT_REPO_ROOT" || exit 1
NM=${1:-1024}; NP=${2:-100000000}
O=$GRAFT_REPO_ROOT/gpurun_out/ktrace
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace -d "$O" -o t --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload pk --nmesh $NM --npk $NP --steps 4 --warmup 2 --no-cpu > "$O/bench.log" 2>&1 || { tail -5 "$O/bench.log"; exit 1; }
python3 - "$O" <<'PY'
import csv, sys
O = sys.argv[1]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:48])
      for r in csv.DictReader(open(O + '/t_kernel_trace.csv'))]
try:
    for r in csv.DictReader(open(O + '/t_memory_copy_trace.csv')):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
except OSError:
    pass
ev.sort()
ev = ev[-60:]
t0, prev = ev[0][0], ev[0][0]
for s, e, n in ev:
    print(f"{(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {n}")
    prev = max(prev, e)
PY
rm -f "$O"/t_kernel_trace.csv
