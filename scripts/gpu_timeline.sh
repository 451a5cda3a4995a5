#!/bin/bash
# kernel + copy timeline of one P(k) step (gaps between device activities): gpu_timeline.sh [nmesh] [npk]
cd "$GRAFT_REPO_ROOT" || exit 1
NM=${1:-1024}; NP=${2:-100000000}
O=$GRAFT_REPO_ROOT/gpurun_out/ktrace
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace -d "$O" -o t --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload pk --nmesh $NM --npk $NP --steps 4 --warmup 2 --no-cpu $PK_OPTS > "$O/bench.log" 2>&1 || { tail -5 "$O/bench.log"; exit 1; }
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
# spans between consecutive counting passes = one deposit + transform (+ binning) each
starts = [i for i, e in enumerate(ev) if 'lines3_count' in e[2] or 'lines_count' in e[2]]
for a, b in zip(starts, starts[1:]):
    busy = sum(e[1] - e[0] for e in ev[a:b])
    print(f"step of {b - a:3d} activities: span {(ev[b][0] - ev[a][0]) / 1e3:9.1f} us, busy {busy / 1e3:9.1f} us, last activity ends {(max(e[1] for e in ev[a:b]) - ev[a][0]) / 1e3:9.1f}")
ev = ev[-60:]
t0, prev = ev[0][0], ev[0][0]
for s, e, n in ev:
    print(f"{(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {n}")
    prev = max(prev, e)
PY
rm -f "$O"/t_kernel_trace.csv
