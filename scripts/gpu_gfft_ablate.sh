#!/bin/bash
# where the time of the mixed-radix passes goes: no stages (gfft_dbg=1), no mesh traffic (2), and the counters of the full form
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/gfft
mkdir -p "$O"
NM=${NM:-1536}
for d in 0 1 2; do
  timeout -k 10 300 python bench.py --workload pk --nmesh $NM --npk 100000000 --steps 3 --warmup 1 --no-cpu --option gfft_dbg=$d > "$O/ab$d.json" 2> "$O/ab$d.err" || { tail -3 "$O/ab$d.err"; exit 1; }
  python - "$O/ab$d.json" "dbg=$d" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], {k: round(v, 3) for k, v in d["kernels_ms"].items() if 'fft' in k})
PY
done
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-include-regex "gfft" -d "$O/pmc_$tag" -o pmc --output-format csv -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload pk --nmesh $NM --npk 100000000 --steps 1 --warmup 0 --no-cpu > "$O/pmc_$tag.log" 2>&1 || { tail -5 "$O/pmc_$tag.log"; exit 1; }
done
python - "$O" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    print(k, {c: f'{x:.4g}' for c, x in sorted(v.items())})
PY
