#!/bin/bash
# SQ / TCC counter passes over the kernels of the P(k) bench (separate passes, kernel-trace only): gpu_pk_pmc.sh [nmesh]
# PK_OPTS="--option name=value ..." is handed to bench.py; PK_KEEP=substring prints only kernels whose name holds it
cd "$GRAFT_REPO_ROOT" || exit 1
NM=${1:-1024}
O=$GRAFT_REPO_ROOT/gpurun_out/pk_pmc
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
         "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum WRITE_SIZE"; do
  i=$((i + 1))
  timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/p$i" -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload pk --nmesh $NM --steps 2 --warmup 1 --no-cpu $PK_OPTS > "$O/p$i.log" 2>&1 || tail -5 "$O/p$i.log"
done
python3 - "$O" <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
for d in sorted(glob.glob(O + '/p*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        last = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').replace('abacus::', '').split('(')[0][:44]
            if k.startswith('__amd') or 'scan_' in k or os.environ.get('PK_KEEP', '') not in k:
                continue
            last.setdefault(k, {})[r['Counter_Name']] = (int(r['Dispatch_Id']), float(r['Counter_Value']))   # keeps the last dispatch
        for k, v in last.items():
            print(k, {c: '%.4g' % x[1] for c, x in v.items()})
PY
find "$O" \( -name "*kernel_trace.csv" -o -name "*counter_collection.csv" -o -name "*.db" \) -delete
